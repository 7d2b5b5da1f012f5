#!/usr/bin/env python3
"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  Builds the checker's native pieces:

  oracle/_ref/libviya_time.so   the reference's OWN src/util/time.cc + time.h, compiled where
                                they lie with plain g++ (they build standalone).
                                Only when /root/reference is present; the output directory is
                                git-ignored and travels to the GPU box with the snapshot.

Everything else of the reference's path needs Boost / glog / nlohmann-json / fmt / CRoaring /
cityhash (all absent: third_party/ submodules are empty) plus the runtime g++ JIT over generated
headers, so the rest of the reference is UNBUILDABLE here and is restated in viya_oracle.py
and cpu_twin.py instead (pinned by tests/golden/reference_cases.json).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def build_ref_time(force=False):
    src = os.path.join(REF, "src", "util", "time.cc")
    if not os.path.exists(src):
        return None
    out_dir = os.path.join(HERE, "_ref")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "libviya_time.so")
    wrap = os.path.join(HERE, "ref_time_wrap.cc")
    if not force and os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(src), os.path.getmtime(wrap)):
        return out
    subprocess.run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-I", os.path.join(REF, "src"), src, wrap,
                    "-o", out], check=True)
    return out


if __name__ == "__main__":
    print(build_ref_time(force="--force" in sys.argv))

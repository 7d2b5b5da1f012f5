#!/usr/bin/env python3
"""Build libviya_hip.so a second time with extra hipcc flags into viyadb_amd/build/variants/<name>/ (measurement only).
Run a process against it with VIYA_HIP_LIB=<path> (viyadb_amd/capi.py honours it).
usage: python tools/build_variant.py <name> <flag> [<flag> ...]"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viyadb_amd import build as b  # noqa: E402


def main():
    name, flags = sys.argv[1], sys.argv[2:]
    out = os.path.join(b.HERE, "build", "variants", name)
    os.makedirs(out, exist_ok=True)
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    units = [s for s in b.hip_sources() if s.endswith(".hip")]
    base = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-unused-result", "-Wno-pass-failed"]
    procs, objs = [], []
    for u in units:
        o = os.path.join(out, os.path.basename(u)[:-4] + ".o")
        objs.append(o)
        procs.append(subprocess.Popen([hipcc] + base + flags + ["-c", u, "-o", o]))
    if any(p.wait() for p in procs):
        raise SystemExit("compile failed")
    lib = os.path.join(out, "libviya_hip.so")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib], check=True)
    print(lib)


if __name__ == "__main__":
    main()

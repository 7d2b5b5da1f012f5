"""In-tree native builds. `hipcc --offload-arch=gfx950` cross-compiles without a GPU."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libviya_hip.so")


def _newer(target: str, sources) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def hip_sources():
    incl = os.path.join(os.path.dirname(HERE), "include", "viya_hip.h")
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))] + [incl]


def build_hip(force: bool = False, verbose: bool = False) -> str:
    """libviya_hip.so: kernels + C-ABI for gfx950. One object per .hip (compiled in parallel), then linked."""
    srcs = hip_sources()
    if not force and _newer(LIB, srcs):
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [s for s in srcs if s.endswith(".h")]
    units = [s for s in srcs if s.endswith(".hip")]
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-unused-result"]
    flags += os.environ.get("VH_EXTRA_HIPCC_FLAGS", "").split()   # experiments (e.g. -DVH_SUBSTEPS=2); force=True to apply
    procs, objs = [], []
    for u in units:
        o = os.path.join(objdir, os.path.basename(u)[:-4] + ".o")
        objs.append(o)
        if not force and _newer(o, [u] + hdrs):
            continue
        cmd = [hipcc] + flags + ["-c", u, "-o", o]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


def build_all(force: bool = False, verbose: bool = False):
    out = [build_hip(force, verbose)]
    host = os.path.join(HERE, "host", "build_host.py")
    if os.path.exists(host):
        from .host import build_host
        out.append(build_host.build(force, verbose))
    return out


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))

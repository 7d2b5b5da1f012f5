"""In-tree build of libviya_host.so (C++17 host shim) against libviya_hip.so."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
LIB = os.path.join(PKG, "libviya_host.so")
SRCS = ["viya_db.cc", "viya_query.cc", "gpu_aggregate.cc", "partial_state.cc", "shim_session.cc", "shim_codegen.cc", "viya_host_c.cc"]


def build(force: bool = False, verbose: bool = False) -> str:
    srcs = [os.path.join(HERE, s) for s in SRCS]
    deps = srcs + [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")] + \
        [os.path.join(os.path.dirname(PKG), "include", h) for h in ("viya_hip.h", "viya_host.h", "viya_shim.h")]
    hip = os.path.join(PKG, "libviya_hip.so")
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps + [hip]):
        return LIB
    cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"] + srcs + \
          ["-L" + PKG, "-lviya_hip", "-Wl,-rpath,$ORIGIN", "-lpthread", "-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))

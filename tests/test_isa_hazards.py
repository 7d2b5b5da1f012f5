"""Static checks on the gfx950 machine code of the built kernels (no GPU needed).

Round 1 hit a code-generation hazard: for a kernel-argument struct that mixes byte-sized and 64-bit
members and is indexed dynamically, hipcc (ROCm 7.2) emitted ``s_load_dwordx2 sN, s[base], 0x5e`` where
``base`` pointed at a byte member (2 mod 4).  Scalar memory instructions ignore the two low bits of the
base, so the load silently returned the wrong 8 bytes and every dense group key decoded as 0.  The compiler
only produces an immediate offset that is not a multiple of 4 when it believes the base is misaligned in
the complementary way, so that pattern is what this test looks for in every code object of the library.
"""
import glob
import os
import re
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def _disassemble(lib, tmp):
    """llvm-objdump --offloading drops one code object per TU next to its input, so work on a copy."""
    copy = os.path.join(tmp, os.path.basename(lib))
    shutil.copy(lib, copy)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", copy], check=True, capture_output=True)
    for co in sorted(glob.glob(copy + ".*gfx950")):
        yield os.path.basename(co), subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], check=True,
                                                   capture_output=True, text=True).stdout


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-objdump")), reason="no llvm-objdump in this image")
def test_no_misaligned_scalar_loads(tmp_path):
    from viyadb_amd import build
    build.build_hip()
    pat = re.compile(r"\bs_(?:buffer_)?load_dword\w*\s+\S+,\s*s\[\d+:\d+\],\s*(0x[0-9a-f]+|\d+)\b")
    total, bad = 0, []
    for name, text in _disassemble(build.LIB, str(tmp_path)):
        for line in text.splitlines():
            m = pat.search(line)
            if not m:
                continue
            total += 1
            if int(m.group(1), 0) % 4:
                bad.append(f"{name}: {line.strip()[:120]}")
    assert total > 100, "disassembly did not contain the kernels"
    assert not bad, "scalar loads with a non-dword immediate offset (misaligned base):\n" + "\n".join(bad[:10])

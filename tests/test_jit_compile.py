"""Per-query compiled scan kernels, CPU side: the text viyadb_amd/csrc/vh_jit.hip generates for canonical plan shapes
compiles for gfx950 with hipRTC (which cross-compiles without a GPU), and the code objects look the way DESIGN.md says they
do — packed predicate columns compared in place (SDWA selectors), no scratch, few enough registers for 7 waves per SIMD on
the C3 shape. The reference's counterpart: every generated query function must pass g++ (src/codegen/compiler.cc:97-144)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from viyadb_amd import capi

LLVM = "/opt/rocm/lib/llvm/bin"
SHAPES = {0: "C3: narrow predicate copies, record gathers, tuples for DENSE_PART",
          1: "C3 from the arenas into the dense HBM table",
          2: "int32 / float range, LDS table, SUM + MAX(double)",
          3: "time rollup keys, hash + LDS front table, IN on u8, != on i64",
          4: "wide hash key (double, i16, u64), row-id MIN, NOT IN on u16",
          5: "no filter, no group columns",
          6: "C5: hashed partitioning, 32-byte tuples that carry the ids of a bitset metric",
          7: "C3 with the payload in a compressed 8-byte record (i64 in 4 bytes, u32s in 2 and 1)",
          8: "C5 with packed 16-byte tuples (payload, two ids and their count in one word)",
          9: "C3 with one-word tuples for DENSE_PART (gid, SUM value and COUNT value in 29 bits)",
          10: "C3 with the payload in a 4-byte bit-field record",
          11: "C2 in the no-compaction form (a lane keeps its own rows, payload columns with vector loads)",
          12: "C3 with the predicate columns as bit fields of a predicate projection's byte planes (3 bytes per row)",
          13: "C3 with the payload records streamed beside the predicate planes (a survivor's record queued in its row's place, no gathers)",
          14: "C3 with bit-sliced predicate columns (22 planes of one bit per row, comparisons bit-serial on 32 rows per lane)",
          15: "C5 (32-byte tuples) whose scan writes the level-A pool itself: 1024-thread blocks, waiting lines per digit in LDS",
          16: "C5 (packed 16-byte tuples) whose scan writes the level-A pool itself",
          18: "C3 with one-word tuples leaving through the block's ring writer (16 partitions' waiting lines per block, extents by position)",
          19: "C3 with two-word tuples leaving through the block's ring writer"}


def _compile(which, tmp_path):
    import __graft_entry__ as g
    g.build()
    lib = capi.load()
    buf = C.create_string_buffer(1 << 20)
    out = str(tmp_path / f"shape{which}.hsaco")
    rc = lib.vh_jit_selftest(which, out.encode(), buf, len(buf))
    return rc, buf.value.decode(), out


def _meta(path):
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", path], capture_output=True, text=True, check=True).stdout
    return {k: int(re.search(rf"\.{k}:\s+(\d+)", notes).group(1)) for k in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "vgpr_spill_count")}


@pytest.mark.parametrize("which", sorted(SHAPES))
def test_generated_text_compiles_for_gfx950(which, tmp_path):
    rc, text, out = _compile(which, tmp_path)
    assert rc == 0, f"{SHAPES[which]}:\n{text[:4000]}"
    assert "vj_scan<VJ>" in text and os.path.getsize(out) > 4096
    m = _meta(out)
    assert m["private_segment_fixed_size"] == 0 and m["vgpr_spill_count"] == 0, (SHAPES[which], m)


def test_c3_shape_compares_packed_columns_in_place(tmp_path):
    rc, text, out = _compile(0, tmp_path)
    assert rc == 0, text[:4000]
    isa = subprocess.run([f"{LLVM}/llvm-objdump", "-d", out], capture_output=True, text=True, check=True).stdout
    # three predicates x 16 row slots, twice (full steps / the step that reaches a segment's end): every one a single SDWA compare
    # writing an SGPR pair, none rebuilt from a per-lane mask
    assert len(re.findall(r"v_cmp_eq_u32_sdwa s\[", isa)) == 32 and len(re.findall(r"v_cmp_lt_u32_sdwa s\[", isa)) == 32
    assert len(re.findall(r"v_cmp_ge_u32_sdwa s\[", isa)) == 32
    assert "src0_sel:BYTE_3" in isa and "src0_sel:WORD_1" in isa
    m = _meta(out)
    assert m["vgpr_count"] <= 72, m      # 7 waves per SIMD (the queues and waiting lines in LDS allow six blocks of four waves per CU)


def test_unknown_shape_is_refused():
    assert capi.load().vh_jit_selftest(99, None, None, 0) == -1


def test_compiles_with_the_hiprtc_a_torch_process_carries():
    """bench.py and the GPU tests import torch first, and the wheel bundles its own (older) libhiprtc / comgr: the generated
    text must not lean on builtins only the newer compiler knows (round 3 found `__builtin_amdgcn_inverse_ballot_w64` that way)."""
    import sys
    code = ("import torch, ctypes as C\n"
            "from viyadb_amd import capi\n"
            "lib = capi.load(); buf = C.create_string_buffer(1 << 20)\n"
            "rcs = [lib.vh_jit_selftest(w, None, buf, len(buf)) for w in range(12)]\n"
            "assert rcs == [0] * 12, (rcs, buf.value.decode()[:2000])\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]

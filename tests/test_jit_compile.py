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
          18: "C3 with one-word tuples (= shape 9: every DENSE_PART tuple of one or two words leaves through the block's ring writer)",
          19: "C3 with two-word tuples (= shape 0)",
          20: "C3 with FOUR-byte tuples (gid 17 + SUM value 10 + COUNT value 2 bits: thirty-two to a 128-byte line)"}


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


RING_SHAPES = (0, 7, 9, 10, 12, 13, 14, 15, 16, 20)     # every selftest shape whose tuples leave through the block's ring writer (vh_ring_add_tb): C5's scan-written level A
                                    # with 32- and 16-byte tuples, C3's phase 1 with one- and two-word tuples


@pytest.mark.parametrize("which", RING_SHAPES)
def test_ring_writer_instruction_order(which, tmp_path):
    """the ring writer's correctness rests on the LDS (DS) instructions of one wave executing in the order they are issued —
    a hardware rule (see the header comment of vh_ring_add_tb), not a HIP memory-model guarantee — with the COMPILER held by signal fences.
    What the fences must produce is asserted here on the disassembly of every ring shape, site by site:
      gen read (ds_read_b32) -> the tuple into the waiting line (ds_write_b64 / b128, b128 x 2 for 32-byte tuples) -> IMMEDIATELY the
      `done` count (ds_add_rtn_u32) -> the owner's list entry (ds_write_b64) -> the copy-out's list read (ds_read_b64) and line read
      (ds_read_b128) -> the line's way out (global_store_dwordx4) -> `done` reset, then `gen` (two ds_write_b32 back to back, the first at the
      address the count went to) — and no `s_waitcnt vmcnt` between the line read and the `gen` store (the next step's predicate loads
      stay in flight). A compiler that reorders any of it fails this test instead of corrupting tuples silently."""
    rc, text, out = _compile(which, tmp_path)
    assert rc == 0, text[:3000]
    isa = subprocess.run([f"{LLVM}/llvm-objdump", "-d", out], capture_output=True, text=True, check=True).stdout
    body = isa[isa.index("<viya_jit_scan_selftest>:"):]
    nxt = re.search(r"\n[0-9a-f]+ <", body[10:])
    L = [l.split("//")[0].strip() for l in (body[:nxt.start() + 10] if nxt else body).splitlines()]
    store = r"ds_write_b(64|128)\b" if which != 20 else r"ds_write_b32\b"       # (four-byte tuples: the store is a ds_write_b32 — one, not the back-to-back pair that resets `done` and bumps `gen`)
    sites = [i for i, l in enumerate(L) if l.startswith("ds_add_rtn_u32") and re.match(store, L[i - 1]) and not (which == 20 and re.match(r"ds_write_b32\b", L[i - 2]))]
    # a ring site per drain form the shape compiles (full steps / the step that reaches a segment's end; C5 drains two survivors per lane)
    assert len(sites) >= 1, "no ring-writer site found"

    def first(pat, start, stop):
        for k in range(start, stop):
            if re.match(pat, L[k]):
                return k
        return None
    for n, i in enumerate(sites):
        stop = sites[n + 1] if n + 1 < len(sites) else min(len(L), i + 400)
        # (1) the place's generation is read before the tuple is written
        k = i - 1
        while re.match(r"ds_write_b128\b", L[k - 1]):
            k -= 1                                       # (32-byte tuples: two stores, back to back)
        gen_read = max([j for j in range(max(0, k - 40), k) if re.match(r"ds_read_b32\b", L[j])], default=None)
        assert gen_read is not None, (which, n, L[k - 10:k + 1])
        # (2) tuple store(s) and the count are adjacent: nothing the compiler could have slipped between them
        assert all(re.match(store, L[j]) for j in range(k, i)), (which, n, L[k:i + 1])
        # (3) the owner's part, in program order
        p_list = first(r"ds_write_b64\b", i + 1, stop)
        p_rl = first(r"ds_read_b64\b", (p_list or stop) + 1, stop)
        p_line = first(r"ds_read_b128\b", (p_rl or stop) + 1, stop)
        p_store = first(r"global_store_dwordx4\b", (p_line or stop) + 1, stop)
        p_done0 = first(r"ds_write_b32\b", (p_store or stop) + 1, stop)
        p_gen = first(r"ds_write_b32\b", (p_done0 or stop) + 1, stop)
        assert None not in (p_list, p_rl, p_line, p_store, p_done0, p_gen), (which, n, p_list, p_rl, p_line, p_store, p_done0, p_gen)
        # done[rl] = 0 (the address the count above went to), THEN gen[rl] = want + 1 (another address)
        cnt = re.match(r"ds_add_rtn_u32 v\d+, (v\d+), v\d+(?: offset:(\d+))?", L[i])
        a, b = re.match(r"ds_write_b32 (v\d+), v\d+(?: offset:(\d+))?", L[p_done0]), re.match(r"ds_write_b32 (v\d+), v\d+(?: offset:(\d+))?", L[p_gen])
        assert cnt and a and b and a.groups() == cnt.groups() and b.groups() != cnt.groups(), (which, n, L[i], L[p_done0], L[p_gen])
        assert p_gen == p_done0 + 1, (which, n, L[p_done0:p_gen + 1])
        assert not any("vmcnt" in L[j] for j in range(p_line, p_gen + 1)), (which, n, [L[j] for j in range(p_line, p_gen + 1) if "vmcnt" in L[j]])

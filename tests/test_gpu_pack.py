"""Payload projections (vh_table_pack): a second, row-major mirror of a few columns from which the compacting scan
kernels gather a survivor's group / metric values. Results must be those of the column arenas — i.e. the oracle's —
for every table organisation and element type, and the projection must follow vh_segment_sync*."""
import numpy as np
import pytest

from oracle import viya_oracle as vo
from tests.parity import check_workload, compare
from tests.planner import mirror_table, plan_from_query
from tests.test_gpu_typed import F, NOW, TYPES, _rand, run, typed_table
from viyadb_amd import capi

pytestmark = pytest.mark.gpu
PACK = capi.PLAN_FORCE_PACK


@pytest.fixture(scope="module", autouse=True)
def _init():
    from viyadb_amd import executor
    executor.init(0)


@pytest.fixture(scope="module")
def typed():
    tab = typed_table()
    dt = mirror_table(tab)
    yield tab, dt
    dt.close()


@pytest.mark.parametrize("flags,path", [(0, "dense_global"), (64, "dense_part"), (1, "hash"), (8, "dense_global"), (2, "dense_global"),
                                        (16 | 32, "dense_global"), (1 | 2048, "hash"), (9, "hash"), (8 | 64, "dense_global")])
def test_c3_through_a_projection(flags, path):
    from viyadb_amd import synth
    w = synth.c3(segment_rows=250_000)
    res, _ = check_workload(w, nseg=4, rows_per_seg=249_991, flags=flags | PACK, expect_path=path)
    assert res.packed


@pytest.mark.parametrize("flags,path", [(128, "dense_lds"), (128 | 2, "dense_global"), (128 | 1, "hash"), (8, "dense_lds")])
def test_c2_through_a_projection(flags, path):
    from viyadb_amd import synth
    w = synth.c2(segment_rows=250_000)
    res, _ = check_workload(w, nseg=3, flags=flags | PACK, expect_path=path)
    assert res.packed
    # the lanes kernels read column ranges, never records
    res, _ = check_workload(w, nseg=3, flags=256 | PACK)
    assert res.lanes and not res.packed


@pytest.mark.parametrize("t", TYPES)
@pytest.mark.parametrize("flags", [0, 1, 8, 64])
def test_every_type_as_packed_metric(typed, t, flags):
    tab, dt = typed
    res, _ = run(tab, dt, {"dimensions": ["s8", "flag"], "metrics": ["count"] + [f"{t}_{a}" for a in ("sum", "min", "max", "avg")],
                           "filter": F("lt", "d_uint", "20")}, flags=flags | PACK)
    assert res.packed


@pytest.mark.parametrize("dims", [["d_byte", "d_float", "d_double"], ["d_ulong", "d_long"], ["s8", "s16", "s32", "flag", "d_short"],
                                  ["d_ubyte", "d_ushort", "id"], ["uts", "ts"]])
@pytest.mark.parametrize("flags", [0, 1])
def test_every_type_as_packed_key(typed, dims, flags):
    tab, dt = typed
    res, _ = run(tab, dt, {"dimensions": dims, "metrics": ["count", "int_sum", "double_max"], "filter": F("ge", "d_int", "-30")}, flags=flags | PACK)
    assert res.packed


def test_time_truncation_reads_the_projection(typed):
    tab, dt = typed
    res, _ = run(tab, dt, {"select": [{"column": "ts", "granularity": "day"}, {"column": "s8"}, {"column": "count"}],
                           "filter": F("lt", "d_uint", "30")}, flags=PACK)
    assert res.packed and res.path == "hash"


def test_wide_payloads_fall_back_to_the_arenas(typed):
    tab, dt = typed
    # 10 gather columns > VH_PACK_MAX_COLS: no projection can cover them; the query still answers from the arenas
    res, _ = run(tab, dt, {"dimensions": ["s8"], "metrics": ["count", "long_sum", "long_min", "long_max", "double_sum", "ulong_max", "int_sum",
                                                             "double_min", "ulong_min"], "filter": F("lt", "d_uint", "20")}, flags=PACK)
    assert not res.packed


@pytest.mark.parametrize("t", ["int", "double", "ubyte"])
def test_hidden_count_travels_in_the_projection(t):
    rng = np.random.default_rng(5)
    tab = vo.Table({"name": "t", "segment_size": 20000, "dimensions": [{"name": "k", "type": "ushort"}, {"name": "f", "type": "uint"}],
                    "metrics": [{"name": "a", "type": t + "_avg"}]})
    for _ in range(2):
        n = 15000
        tab.add_segment_arrays([rng.integers(0, 300, n).astype(np.uint16), rng.integers(0, 100, n).astype(np.uint32)],
                               [_rand(rng, vo.NUMERIC_TYPES[t][0], n, small=True)], rng.integers(1, 5, n).astype(np.uint64), n)
    dt = mirror_table(tab)
    try:
        res, _ = run(tab, dt, {"dimensions": ["k"], "metrics": ["a"], "filter": F("lt", "f", "10")}, flags=PACK)
        assert res.hidden_count is not None and res.packed
    finally:
        dt.close()


def test_projection_follows_segment_syncs():
    """vh_segment_sync / _sync_range after the projection was built: the changed segment is re-packed before the next query
    that gathers from it (upsert appends rows and updates metrics of existing rows in place, src/codegen/db/upsert.cc:384-411)."""
    rng = np.random.default_rng(11)
    desc = {"name": "t", "segment_size": 30000, "dimensions": [{"name": "a", "type": "uint"}, {"name": "b", "type": "ushort"}, {"name": "f", "type": "uint"}],
            "metrics": [{"name": "v", "type": "long_sum"}, {"name": "count", "type": "count"}]}

    def seg(n):
        return ([rng.integers(0, 50, n).astype(np.uint32), rng.integers(0, 40, n).astype(np.uint16), rng.integers(0, 100, n).astype(np.uint32)],
                [rng.integers(-1000, 1000, n).astype(np.int64), rng.integers(1, 4, n).astype(np.uint32)])

    tab = vo.Table(desc)
    for n in (30000, 20000):
        d, m = seg(n)
        tab.add_segment_arrays(d, m, None, n)
    dt = mirror_table(tab, reserve=2)
    q = {"dimensions": ["a", "b"], "metrics": ["v", "count"], "filter": F("lt", "f", "8")}
    try:
        dt.pack([0, 1, 3, 4])
        res, _ = run(tab, dt, q)                       # 8 % pass, a projection covers a, b, v, count: taken without being forced
        assert res.packed
        # in-place metric update of rows [100, 5000) of segment 0 + 5000 appended rows in segment 1
        s0, s1 = tab.segments[0], tab.segments[1]
        s0["m"][0][100:5000] += 7
        d, m = seg(5000)
        for i in range(3):
            s1["d"][i] = np.concatenate([s1["d"][i][:20000], d[i]])
        for j in range(2):
            s1["m"][j] = np.concatenate([s1["m"][j][:20000], m[j]])
        s1["size"] = 25000
        import ctypes as C
        cols0 = [s0["d"][0], s0["d"][1], s0["d"][2], s0["m"][0], s0["m"][1]]
        ptrs = (C.c_void_p * 5)(*[np.ascontiguousarray(c).ctypes.data for c in cols0])
        capi.check(dt.lib.vh_segment_sync_range(dt.handle, 0, 100, 4900, 30000, ptrs))
        dt.sync_segment(1, [s1["d"][0], s1["d"][1], s1["d"][2], s1["m"][0], s1["m"][1]], 25000)
        res, _ = run(tab, dt, q)
        assert res.packed
        # a third segment appears (the table grows past its reserve: arenas and the projection move)
        d, m = seg(30000)
        tab.add_segment_arrays(d, m, None, 30000)
        dt.sync_segment(2, d + m, 30000)
        res, _ = run(tab, dt, q)
        assert res.packed and res.scanned_segments == 3
        dt.unpack()
        assert not run(tab, dt, q, flags=0)[0].packed
    finally:
        dt.close()


def test_derived_layouts_moved_to_other_memory():
    """vh_table_relocate (what vh_table_prepare does per candidate place): projections and predicate planes copied to fresh allocations, pointers
    swapped — the same groups before and after, for each kind and both, a sync after the move still lands in the moved layouts, and a prepare
    (whose placement moves them again where the scan is long enough) changes nothing either."""
    from viyadb_amd import synth
    from viyadb_amd.executor import AggPlan
    from tests.parity import build_oracle_table
    w = synth.c3(segment_rows=250_000)
    dt = synth.create_device_table(w, 4, 249_991)
    try:
        want = vo.scan_aggregate(vo.parse_query(build_oracle_table(w, 4, 249_991), w.query), now=getattr(w, "now", NOW))
        plan = AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, flags=64 | capi.PLAN_FORCE_JIT | PACK, groups_hint=w.plan.groups_hint)
        dt.pack(dt.gather_columns(plan)); dt.predpack(dt.filter_columns(plan))
        r = dt.query_agg(plan)
        compare(r, want, "before the move")
        assert r.packed and r.predpack
        for which in (1, 2, 0, 2, 1):
            dt.relocate(which)
            r = dt.query_agg(plan)
            compare(r, want, f"after relocate({which})")
            assert r.packed and r.predpack
        dt.generate(1, 1, 249_991, 249_991, [c.gen for c in w.columns], 42)      # (segment 1 written again with the same rows: the journalled ranges are re-derived into the MOVED layouts)
        compare(dt.query_agg(plan), want, "after a sync behind the move")
        dt.warm(plan)
        compare(dt.query_agg(plan), want, "after vh_table_prepare")
    finally:
        dt.close()


def test_library_builds_a_projection_for_a_repeated_selective_query():
    """VH_AUTO_PACK (default 3): the third selective query over the same payload columns gets a projection built for it."""
    from viyadb_amd import synth
    from viyadb_amd.executor import AggPlan
    from tests.parity import build_oracle_table
    w = synth.c3(segment_rows=100_000)
    dt = synth.create_device_table(w, 3, 100_000)
    try:
        st = vo.scan_aggregate(vo.parse_query(build_oracle_table(w, 3, 100_000), w.query))
        seen = []
        for _ in range(4):
            res = dt.query_agg(AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics))
            compare(res, st, "auto pack")
            seen.append(res.packed)
        assert seen == [False, False, True, True]
        # an unselective query over the same columns keeps reading the arenas
        res = dt.query_agg(AggPlan(filter=[("rel", 3, capi.OP_LT, 900)], groups=w.plan.groups, metrics=w.plan.metrics))
        assert not res.packed
        res = dt.query_agg(AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, flags=capi.PLAN_NO_PACK))
        assert not res.packed
        compare(res, st, "no pack")
    finally:
        dt.close()


def test_prepare_pays_the_first_use_costs_up_front():
    """vh_table_prepare (executor.warm): the FIRST query of a prepared plan shape already runs on what the third one would otherwise get — a
    payload projection, narrow predicate copies — and vh_result_info.reserved says so; an unprepared twin of the table starts on the arenas.
    (The compiled kernel joins from VH_JIT_MIN_ROWS up, and the measured pool placement from 128 MB of tuples: tests/test_gpu_fullsize.py.)"""
    from viyadb_amd import synth
    from viyadb_amd.executor import AggPlan
    from tests.parity import build_oracle_table
    w = synth.c3(segment_rows=100_000)
    st = vo.scan_aggregate(vo.parse_query(build_oracle_table(w, 3, 100_000), w.query))
    plan = AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics)
    cold, warm = synth.create_device_table(w, 3, 100_000), synth.create_device_table(w, 3, 100_000)
    try:
        flags = warm.warm(plan)
        assert flags & 8 and flags & 16, flags                    # bit 3: projection, bit 4: narrow predicate copies
        first = warm.query_agg(plan)
        compare(first, st, "prepared")
        assert first.packed and first.narrow
        res = cold.query_agg(plan)
        compare(res, st, "unprepared")
        assert not res.packed and not res.narrow
    finally:
        cold.close(); warm.close()


# ---- compressed records (round 3): integers at the width their values need, read by the per-query compiled kernels only
JITPACK = capi.PLAN_FORCE_JIT | PACK
from tests.conftest import JIT_OFF  # noqa: E402
needs_jit = pytest.mark.skipif(JIT_OFF, reason="VH_JIT=off: compressed records are read by the per-query compiled kernels only")


@pytest.fixture(scope="module")
def typedc():      # its own mirror: a plain projection that covers the same columns would be taken instead of building a compressed one
    tab = typed_table()
    dt = mirror_table(tab)
    yield tab, dt
    dt.close()


@needs_jit
@pytest.mark.parametrize("flags,path", [(0, "dense_global"), (64, "dense_part"), (1, "hash"), (2, "dense_global"), (16 | 32, "dense_global")])
def test_c3_through_compressed_records(flags, path):
    from viyadb_amd import synth
    w = synth.c3(segment_rows=250_000)
    res, _ = check_workload(w, nseg=4, rows_per_seg=249_991, flags=flags | JITPACK, expect_path=path)
    assert res.packed and res.jit and res.packed_compressed


@needs_jit
@pytest.mark.parametrize("t", TYPES)
@pytest.mark.parametrize("flags", [0, 1, 64])
def test_every_type_as_compressed_metric(typedc, t, flags):
    tab, dt = typedc
    res, _ = run(tab, dt, {"dimensions": ["s8", "flag"], "metrics": ["count"] + [f"{t}_{a}" for a in ("sum", "min", "max")],
                           "filter": F("lt", "d_uint", "20")}, flags=flags | JITPACK | capi.PLAN_NO_LANES)
    assert res.packed and res.jit and res.packed_compressed


@pytest.mark.parametrize("dims", [["d_byte", "d_float", "d_double"], ["d_ulong", "d_long"], ["s8", "s16", "s32", "flag", "d_short"],
                                  ["d_ubyte", "d_ushort", "id"], ["uts", "ts"], ["d_int", "d_short"]])
@needs_jit
@pytest.mark.parametrize("flags", [0, 1])
def test_every_type_as_compressed_key(typedc, dims, flags):
    tab, dt = typedc
    res, _ = run(tab, dt, {"dimensions": dims, "metrics": ["count", "int_sum", "long_min"], "filter": F("ge", "d_int", "-30")},
                 flags=flags | JITPACK | capi.PLAN_NO_LANES)
    assert res.packed and res.jit and res.packed_compressed


@needs_jit
def test_prebuilt_kernels_never_read_compressed_records(typedc):
    """The same columns, asked for by a plan the pre-built kernels run: a plain projection is built next to the compressed one."""
    tab, dt = typedc
    q = {"dimensions": ["s16", "d_short"], "metrics": ["count", "long_sum"], "filter": F("lt", "d_uint", "20")}
    a, _ = run(tab, dt, q, flags=JITPACK | capi.PLAN_NO_LANES)
    b, _ = run(tab, dt, q, flags=capi.PLAN_NO_JIT | PACK | capi.PLAN_NO_LANES)
    assert a.packed_compressed and b.packed and not b.packed_compressed and not b.jit


@needs_jit
def test_compressed_records_follow_syncs_and_outgrown_widths():
    """Values that fit one byte / two bytes when the projection is built; a later vh_segment_sync brings values that need more: the
    projection is void (pack_kernel notices while re-packing the segment) and is rebuilt wider before the query reads it."""
    rng = np.random.default_rng(23)
    desc = {"name": "t", "segment_size": 30000, "dimensions": [{"name": "a", "type": "uint"}, {"name": "b", "type": "int"}, {"name": "f", "type": "uint"}],
            "metrics": [{"name": "v", "type": "long_sum"}, {"name": "count", "type": "count"}, {"name": "w", "type": "int_min"}]}

    def seg(n, amax=50, vmag=100):
        return ([rng.integers(0, amax, n).astype(np.uint32), rng.integers(-40, 40, n).astype(np.int32), rng.integers(0, 100, n).astype(np.uint32)],
                [rng.integers(-vmag, vmag, n).astype(np.int64), rng.integers(1, 4, n).astype(np.uint32), rng.integers(-vmag, vmag, n).astype(np.int32)])

    tab = vo.Table(desc)
    for n in (30000, 20000):
        d, m = seg(n)
        tab.add_segment_arrays(d, m, None, n)
    dt = mirror_table(tab, reserve=2)
    q = {"dimensions": ["a", "b"], "metrics": ["v", "count", "w"], "filter": F("lt", "f", "8")}
    fl = capi.PLAN_FORCE_JIT
    try:
        dt.pack([0, 1, 3, 4, 5], compressed=True)
        res, _ = run(tab, dt, q, flags=fl)
        assert res.packed and res.packed_compressed and res.jit
        before = dt.info()[2]
        # segment 1 is replaced by rows whose values need 4 bytes (a), 8 bytes (v): every stored width is outgrown
        d, m = seg(25000, amax=3_000_000, vmag=2 ** 40)
        s1 = tab.segments[1]
        for i in range(3):
            s1["d"][i] = d[i]
        for j in range(3):
            s1["m"][j] = m[j]
        s1["size"] = 25000
        dt.sync_segment(1, d + m, 25000)
        res, _ = run(tab, dt, q, flags=fl | capi.PLAN_FORCE_HASH)
        assert res.packed and res.packed_compressed
        assert dt.info()[2] > before            # wider records
        # and a third segment within the new widths: re-packed in place
        d, m = seg(30000, amax=2_000_000, vmag=2 ** 39)
        tab.add_segment_arrays(d, m, None, 30000)
        dt.sync_segment(2, d + m, 30000)
        res, _ = run(tab, dt, q, flags=fl | capi.PLAN_FORCE_HASH)
        assert res.packed and res.packed_compressed and res.scanned_segments == 3
    finally:
        dt.close()


@needs_jit
def test_bit_field_records_follow_syncs_outgrown_bits_and_negative_values():
    """A compressed projection whose columns are all non-negative integers keeps every column at the BITS its values need, in one 4- or 8-byte
    word (C3: 29 bits -> 4-byte records). Here: a (6 bits) + b (6) + v (7) + count (2) + w (7) = 28 bits -> 4 bytes per row; a later sync
    brings values that need more bits (the word grows to 8 bytes), then NEGATIVE values (fields are unsigned: back to byte widths). The rows
    are the oracle's every time."""
    rng = np.random.default_rng(29)
    desc = {"name": "t", "segment_size": 30000, "dimensions": [{"name": "a", "type": "uint"}, {"name": "b", "type": "int"}, {"name": "f", "type": "uint"}],
            "metrics": [{"name": "v", "type": "long_sum"}, {"name": "count", "type": "count"}, {"name": "w", "type": "int_min"}]}

    def seg(n, amax=50, vlo=0, vhi=100):
        return ([rng.integers(0, amax, n).astype(np.uint32), rng.integers(0, 40, n).astype(np.int32), rng.integers(0, 100, n).astype(np.uint32)],
                [rng.integers(vlo, vhi, n).astype(np.int64), rng.integers(1, 4, n).astype(np.uint32), rng.integers(max(vlo, -2 ** 31), min(vhi, 2 ** 31 - 1), n).astype(np.int32)])

    tab = vo.Table(desc)
    for n in (30000, 20000):
        d, m = seg(n)
        tab.add_segment_arrays(d, m, None, n)
    dt = mirror_table(tab, reserve=3)
    q = {"dimensions": ["a", "b"], "metrics": ["v", "count", "w"], "filter": F("lt", "f", "8")}
    fl = capi.PLAN_FORCE_JIT
    rows_cap = 4 * 30208          # the mirror reserves four segments (table_grow), rows padded to 256

    def replace(segno, d, m, n):
        sg = tab.segments[segno]
        for i in range(3):
            sg["d"][i] = d[i]
        for j in range(3):
            sg["m"][j] = m[j]
        sg["size"] = n
        dt.sync_segment(segno, d + m, n)
    try:
        base = dt.info()[2]
        dt.pack([0, 1, 3, 4, 5], compressed=True)
        res, _ = run(tab, dt, q, flags=fl)
        assert res.packed and res.packed_compressed and res.jit
        assert dt.info()[2] - base < rows_cap * 5                      # 4-byte records (8-byte ones would take 8 x rows)
        d, m = seg(25000, amax=3_000_000, vhi=2 ** 30)                  # a needs 22 bits, v and w 30: 22 + 6 + 30 + 2 + 30 > 64 -> byte widths; first within 64:
        d2, m2 = seg(25000, amax=3000, vhi=2 ** 14)                     # 12 + 6 + 14 + 2 + 14 = 48 bits -> 8-byte word
        replace(1, d2, m2, 25000)
        res, _ = run(tab, dt, q, flags=fl | capi.PLAN_FORCE_HASH)
        assert res.packed and res.packed_compressed
        grown = dt.info()[2] - base
        assert rows_cap * 8 <= grown < rows_cap * 9
        replace(1, d, m, 25000)
        res, _ = run(tab, dt, q, flags=fl | capi.PLAN_FORCE_HASH)
        assert res.packed and res.packed_compressed and dt.info()[2] - base > grown
        d3, m3 = seg(30000, vlo=-50, vhi=50)                            # negative values: unsigned bit fields cannot hold them
        replace(0, d3, m3, 30000)
        res, _ = run(tab, dt, q, flags=fl)
        assert res.packed and res.packed_compressed
    finally:
        dt.close()


@needs_jit
@pytest.mark.parametrize("flags,path", [(0, "dense_global"), (64, "dense_part"), (1, "hash"), (2, "dense_global"), (16 | 32, "dense_global"), (1 | 2048, "hash")])
def test_c3_with_streamed_payload_records(flags, path):
    """The compiled compacting scan STREAMS a bit-field projection's 4-byte records beside the predicate columns and queues a survivor's record
    in its row's place (VhJitShape::qpay): no gathers. Same answers under every table organisation, with and without the predicate
    projection; off below the selectivity where gathers are cheaper, and on request (VH_PLAN_NO_QPAY)."""
    from viyadb_amd import synth
    from viyadb_amd.executor import AggPlan
    from tests.parity import build_oracle_table
    w = synth.c3(segment_rows=250_000)
    nseg, rows = 4, 249_991
    dt = synth.create_device_table(w, nseg, rows)
    try:
        st = vo.scan_aggregate(vo.parse_query(build_oracle_table(w, nseg, rows), w.query))
        mk = lambda f: AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, flags=f, groups_hint=w.plan.groups_hint)
        dt.pack(dt.gather_columns(mk(0)), compressed=True)
        base = flags | capi.PLAN_FORCE_JIT | PACK
        base |= capi.PLAN_FORCE_QPAY                        # (5 % pass: below the selectivity from which the library streams unasked)
        res = dt.query_agg(mk(base))
        compare(res, st, f"streamed payload flags={flags}")
        assert res.path == path and res.jit and res.packed and res.packed_compressed and res.streamed_payload, (res.flags, res.kernel)
        res = dt.query_agg(mk((base & ~capi.PLAN_FORCE_QPAY) | capi.PLAN_NO_QPAY))
        compare(res, st, f"gathered payload flags={flags}")
        assert res.packed and not res.streamed_payload
        dt.predpack(dt.filter_columns(mk(0)), sliced=False)      # (queued records go with rows: the byte-plane form)
        res = dt.query_agg(mk(base))
        compare(res, st, f"streamed payload + predicate projection flags={flags}")
        assert res.streamed_payload and res.predpack and not res.sliced
    finally:
        dt.close()


@needs_jit
def test_streamed_payload_follows_the_selectivity():
    from viyadb_amd import synth
    from viyadb_amd.executor import AggPlan
    from tests.parity import build_oracle_table
    w = synth.c3(segment_rows=250_000)
    nseg, rows = 4, 250_000
    dt = synth.create_device_table(w, nseg, rows)
    try:
        ot = build_oracle_table(w, nseg, rows)
        dt.pack(dt.gather_columns(AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics)), compressed=True)
        for lit, dlit, want in ((30, 553, False), (447, 553, False), (1000, 0, True)):      # ~0.3 %, ~5 % (gathers are cheaper: measured), 25 % of the rows pass
            flt = [("rel", 2, capi.OP_EQ, 1), ("rel", 3, capi.OP_LT, lit), ("rel", 4, capi.OP_GE, dlit), ("and", 3)]
            q = dict(w.query, filter={"op": "and", "filters": [{"op": "eq", "column": "d2", "value": "1"}, {"op": "lt", "column": "d3", "value": str(lit)},
                                                              {"op": "ge", "column": "d4", "value": str(dlit)}]})
            res = dt.query_agg(AggPlan(filter=flt, groups=w.plan.groups, metrics=w.plan.metrics, flags=capi.PLAN_FORCE_JIT | PACK, groups_hint=w.plan.groups_hint))
            compare(res, vo.scan_aggregate(vo.parse_query(ot, q)), f"d3 < {lit}")
            assert res.streamed_payload == want, (lit, res.flags)
        # every row passes, GROUP BY through the records: still the compacting kernel, 4 bytes per row instead of 20 from the arenas
        q = dict(w.query, filter={"op": "ge", "column": "d3", "value": "0"})
        res = dt.query_agg(AggPlan(filter=[("rel", 3, capi.OP_GE, 0)], groups=w.plan.groups, metrics=w.plan.metrics, flags=capi.PLAN_FORCE_JIT | PACK | 64, groups_hint=w.plan.groups_hint))
        compare(res, vo.scan_aggregate(vo.parse_query(ot, q)), "all rows")
        assert res.streamed_payload and res.path == "dense_part"
    finally:
        dt.close()

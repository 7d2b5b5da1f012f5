"""Narrow copies of predicate columns (vh_table_narrow): 8- / 16-bit copies of unsigned 32-bit columns whose values fit, streamed
by the register-resident kernels instead of the arenas. Results must be those of the arenas — i.e. the oracle's — under every
table organisation; the copies must follow vh_segment_sync*, be dropped when a value stops fitting, and be built unasked for a
column selective queries keep filtering on."""
import numpy as np
import pytest

from oracle import viya_oracle as vo
from tests.parity import build_oracle_table, compare
from tests.planner import mirror_table
from tests.test_gpu_typed import F, run
from viyadb_amd import capi

pytestmark = pytest.mark.gpu
from tests.conftest import JIT_OFF  # noqa: E402
needs_jit = pytest.mark.skipif(JIT_OFF, reason="VH_JIT=off: this layout / form is read by the per-query compiled kernels only")


@pytest.fixture(scope="module", autouse=True)
def _init():
    from viyadb_amd import executor
    executor.init(0)


@pytest.mark.parametrize("flags", [0, 64, 64 | 128, 1, 8, 16, 2, 256, 64 | 32768, 8192, 64 | 8192])
@pytest.mark.parametrize("name", ["C3", "C2"])
def test_workloads_through_narrow_copies(name, flags):
    """C3's predicate columns have 4 / 1000 / 1000 distinct values (one 8-bit and two 16-bit copies); C2's has a million (no copy:
    the request is skipped silently)."""
    from viyadb_amd import synth
    from viyadb_amd.executor import AggPlan
    w = synth.WORKLOADS[name](segment_rows=200_000)
    dt = synth.create_device_table(w, 3, 199_993)
    try:
        plan = AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, flags=flags, groups_hint=w.plan.groups_hint)
        dt.narrow(dt.filter_columns(plan))
        res = dt.query_agg(plan)
        st = vo.scan_aggregate(vo.parse_query(build_oracle_table(w, 3, 199_993), w.query))
        compare(res, st, f"{name} narrow flags={flags}")
        generic = bool(flags & 8)
        assert res.narrow == (name == "C3" and not generic), (res.narrow, res.kernel)
        res = dt.query_agg(AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, flags=flags | capi.PLAN_NO_NARROW,
                                   groups_hint=w.plan.groups_hint))
        compare(res, st, f"{name} arenas flags={flags}")
        assert not res.narrow
    finally:
        dt.close()


def _table(rng, n, hi_a=200, hi_b=60000):
    tab = vo.Table({"name": "t", "segment_size": n, "dimensions": [{"name": "a", "type": "uint"}, {"name": "b", "type": "uint"}, {"name": "g", "type": "uint"}],
                    "metrics": [{"name": "v", "type": "long_sum"}, {"name": "count", "type": "count"}]})
    for _ in range(3):
        tab.add_segment_arrays([rng.integers(0, hi_a, n).astype(np.uint32), rng.integers(0, hi_b, n).astype(np.uint32), rng.integers(0, 300, n).astype(np.uint32)],
                               [rng.integers(-1000, 1000, n).astype(np.int64), np.ones(n, dtype=np.uint32)], None, n)
    return tab


def test_copies_follow_syncs_and_are_dropped_when_values_outgrow_them():
    rng = np.random.default_rng(3)
    n = 50_000
    tab = _table(rng, n)
    dt = mirror_table(tab, reserve=5)
    q = {"dimensions": ["g"], "metrics": ["v", "count"], "filter": {"op": "and", "filters": [F("lt", "a", "40"), F("ge", "b", "1000"), F("ne", "a", "7")]}}   # ~19 % pass: the compacting kernels (the lanes kernels read the 4-byte arenas)
    try:
        dt.narrow([0, 1])
        res, _ = run(tab, dt, q)
        assert res.narrow and not res.lanes
        res, _ = run(tab, dt, dict(q, filter=F("lt", "a", "190")))          # most rows pass: lanes kernel, arenas
        assert res.lanes and not res.narrow
        # IN lists, OR, NOT: every leaf kind over the widened values
        run(tab, dt, dict(q, filter={"op": "or", "filters": [{"op": "in", "column": "a", "values": ["3", "199", "250", "70000"]},
                                                            {"op": "not", "filter": F("le", "b", "59990")}]}))
        # literals beyond the copies' range compare like they do against the 4-byte column
        for flt in (F("lt", "a", "100000"), F("gt", "b", "4000000000"), F("eq", "a", "256"), F("ne", "b", "65536")):
            run(tab, dt, dict(q, filter=flt))
        # a segment is re-synced with different (still fitting) values: the copy follows
        seg = tab.segments[1]
        seg["d"][0][:] = rng.integers(0, 256, n).astype(np.uint32)
        seg["d"][1][:] = rng.integers(0, 65536, n).astype(np.uint32)
        dt.sync_segment(1, [seg["d"][0], seg["d"][1], seg["d"][2], seg["m"][0], seg["m"][1]], n)
        res, _ = run(tab, dt, q)
        assert res.narrow
        # a new segment whose values need more bits: `a` moves to 16 bits or is dropped, `b` is dropped; answers stay right
        tab.add_segment_arrays([rng.integers(0, 5000, n).astype(np.uint32), rng.integers(0, 3_000_000, n).astype(np.uint32), rng.integers(0, 300, n).astype(np.uint32)],
                               [rng.integers(-1000, 1000, n).astype(np.int64), np.ones(n, dtype=np.uint32)], None, n)
        seg = tab.segments[3]
        dt.sync_segment(3, [seg["d"][0], seg["d"][1], seg["d"][2], seg["m"][0], seg["m"][1]], n)
        for _ in range(4):
            run(tab, dt, q)
        run(tab, dt, q, flags=64 | 128)
    finally:
        dt.close()


def test_copies_are_built_unasked_for_columns_queries_keep_filtering_on():
    rng = np.random.default_rng(4)
    n = 50_000
    tab = _table(rng, n)
    dt = mirror_table(tab)
    q = {"dimensions": ["g"], "metrics": ["v", "count"], "filter": F("lt", "a", "30")}
    try:
        seen = [run(tab, dt, q)[0].narrow for _ in range(5)]
        assert seen[0] is False and seen[-1] is True, seen
        dt.unpack()                                     # drops projections and narrow copies
        assert run(tab, dt, q)[0].narrow is False
    finally:
        dt.close()


JIT = capi.PLAN_FORCE_JIT


@needs_jit
@pytest.mark.parametrize("sliced", [True, False])
@pytest.mark.parametrize("flags", [0, 64, 1, 16, 2, 64 | 8192, 8192])
def test_c3_through_a_predicate_projection(flags, sliced):
    """vh_table_predpack: C3's predicate columns as bit fields of one word per row (2 + 10 + 10 bits) — as a 2-byte and a 1-byte plane, or
    bit-sliced into 22 planes of one bit per row whose comparisons run bit-serially on 32 rows per lane — streamed by the compiled scan under
    every table organisation; the pre-built kernels (no compiled kernel asked for) never read it."""
    from viyadb_amd import synth
    from viyadb_amd.executor import AggPlan
    w = synth.c3(segment_rows=200_000)
    dt = synth.create_device_table(w, 3, 199_993)
    try:
        mk = lambda f: AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, flags=f, groups_hint=w.plan.groups_hint)
        dt.predpack(dt.filter_columns(mk(0)), sliced=sliced)
        st = vo.scan_aggregate(vo.parse_query(build_oracle_table(w, 3, 199_993), w.query))
        res = dt.query_agg(mk(flags | JIT))
        compare(res, st, f"C3 predicate projection flags={flags} sliced={sliced}")
        assert res.jit and res.predpack and res.narrow and res.sliced == sliced, (res.flags, res.kernel)
        if sliced:
            res = dt.query_agg(mk(flags | JIT | capi.PLAN_NO_SLICED))
            compare(res, st, f"C3 sliced projection not used flags={flags}")
            assert res.jit and not res.sliced
        res = dt.query_agg(mk(flags | JIT | capi.PLAN_NO_PREDPACK))
        compare(res, st, f"C3 arenas flags={flags}")
        assert res.jit and not res.predpack
        res = dt.query_agg(mk(flags | capi.PLAN_NO_JIT))
        compare(res, st, f"C3 pre-built flags={flags}")
        assert not res.predpack
    finally:
        dt.close()


@needs_jit
def test_predicate_projection_leaves_of_every_kind_and_types():
    """IN lists, OR, NOT, literals beyond a field's range, columns of 1 / 2 / 4 / 8 bytes, a superset projection (a query that filters on
    two of its three columns), and a set of columns that does not qualify (negative values)."""
    rng = np.random.default_rng(13)
    n = 60_000
    tab = vo.Table({"name": "t", "segment_size": n, "dimensions": [{"name": "a", "type": "uint"}, {"name": "b", "type": "ushort"}, {"name": "c", "type": "ubyte"},
                                                                  {"name": "e", "type": "ulong"}, {"name": "s", "type": "int"}, {"name": "g", "type": "uint"}],
                    "metrics": [{"name": "v", "type": "long_sum"}, {"name": "count", "type": "count"}]})
    for _ in range(3):
        tab.add_segment_arrays([rng.integers(0, 900, n).astype(np.uint32), rng.integers(0, 3000, n).astype(np.uint16), rng.integers(0, 5, n).astype(np.uint8),
                                rng.integers(0, 70, n).astype(np.uint64), rng.integers(-50, 50, n).astype(np.int32), rng.integers(0, 300, n).astype(np.uint32)],
                               [rng.integers(-1000, 1000, n).astype(np.int64), np.ones(n, dtype=np.uint32)], None, n)
    dt = mirror_table(tab)
    base = {"dimensions": ["g"], "metrics": ["v", "count"]}
    try:
        dt.predpack([0, 1, 2, 3], sliced=False)    # 10 + 12 + 3 + 7 = 32 bits: two 2-byte planes (what the no-compaction kernels read) ...
        dt.predpack([0, 1, 2, 3], sliced=True)     # ... and 32 planes of one bit per row (what the compacting ones read)
        q = dict(base, filter={"op": "and", "filters": [F("lt", "a", "300"), F("ge", "b", "1000"), F("ne", "c", "2"), F("le", "e", "40")]})
        res, _ = run(tab, dt, q, flags=JIT)
        assert res.predpack and res.jit
        res, _ = run(tab, dt, dict(base, filter={"op": "and", "filters": [F("lt", "a", "100"), F("eq", "c", "1")]}), flags=JIT)     # two of the four columns
        assert res.predpack
        run(tab, dt, dict(base, filter={"op": "or", "filters": [{"op": "in", "column": "a", "values": ["3", "899", "1024", "70000"]},
                                                               {"op": "not", "filter": F("le", "b", "2990")}, {"op": "in", "column": "e", "values": ["0", "69"]}]}), flags=JIT)
        for flt in (F("lt", "a", "100000"), F("gt", "b", "4000"), F("eq", "a", "1024"), F("ne", "c", "200"), F("ge", "e", "5000000000")):
            res, _ = run(tab, dt, dict(base, filter=flt), flags=JIT)
            assert res.predpack
        # a signed column with negative values has no bit field: the query reads its arena
        res, _ = run(tab, dt, dict(base, filter={"op": "and", "filters": [F("lt", "a", "300"), F("gt", "s", "-10")]}), flags=JIT)
        assert res.jit and not res.predpack
    finally:
        dt.close()


@needs_jit
def test_predicate_projection_follows_syncs_and_is_dropped_when_outgrown():
    rng = np.random.default_rng(5)
    n = 50_000
    tab = _table(rng, n, hi_a=16, hi_b=3000)       # 4 + 12 bits: ONE 2-byte plane instead of a 1-byte and a 2-byte narrow copy
    dt = mirror_table(tab, reserve=5)
    q = {"dimensions": ["g"], "metrics": ["v", "count"], "filter": {"op": "and", "filters": [F("lt", "a", "3"), F("ge", "b", "1000"), F("ne", "a", "1")]}}   # ~8 % pass: a compacting kernel
    try:
        dt.predpack([0, 1])
        res, _ = run(tab, dt, q, flags=JIT)
        assert res.predpack and res.sliced
        # rows of a segment change in place (a batch) and a segment is re-synced whole: the planes follow by row range
        seg = tab.segments[1]
        seg["d"][0][1000:1900] = rng.integers(0, 16, 900).astype(np.uint32)
        seg["d"][1][1000:1900] = rng.integers(0, 4096, 900).astype(np.uint32)
        dt.sync_batch([(1, 1000, 900, n, [seg["d"][0], seg["d"][1], seg["d"][2], seg["m"][0], seg["m"][1]], 0)])
        res, _ = run(tab, dt, q, flags=JIT)
        assert res.predpack
        seg = tab.segments[2]
        seg["d"][0][:] = rng.integers(0, 16, n).astype(np.uint32)
        dt.sync_segment(2, [seg["d"][0], seg["d"][1], seg["d"][2], seg["m"][0], seg["m"][1]], n)
        res, _ = run(tab, dt, q, flags=JIT)
        assert res.predpack
        # a new segment whose values need more bits than the fields have: the projection goes, answers stay right
        tab.add_segment_arrays([rng.integers(0, 5000, n).astype(np.uint32), rng.integers(0, 3_000_000, n).astype(np.uint32), rng.integers(0, 300, n).astype(np.uint32)],
                               [rng.integers(-1000, 1000, n).astype(np.int64), np.ones(n, dtype=np.uint32)], None, n)
        seg = tab.segments[3]
        dt.sync_segment(3, [seg["d"][0], seg["d"][1], seg["d"][2], seg["m"][0], seg["m"][1]], n)
        for _ in range(5):                         # (13 + 22 bits no longer fit one 32-bit word: none is built again either)
            res, _ = run(tab, dt, q, flags=JIT)
            assert not res.predpack
    finally:
        dt.close()


@needs_jit
def test_predicate_projection_is_built_unasked_where_the_compiled_kernel_runs():
    rng = np.random.default_rng(6)
    n = 50_000
    tab = _table(rng, n, hi_a=16, hi_b=3000)
    dt = mirror_table(tab)
    q = {"dimensions": ["g"], "metrics": ["v", "count"], "filter": {"op": "and", "filters": [F("lt", "a", "9"), F("ge", "b", "1000")]}}
    try:
        seen = [run(tab, dt, q, flags=JIT)[0].predpack for _ in range(5)]
        assert seen[0] is False and seen[-1] is True, seen
        assert not run(tab, dt, q)[0].predpack         # a table this small gets no compiled kernel unasked: arenas / narrow copies
        dt.unpack()
        assert run(tab, dt, q, flags=JIT)[0].predpack is False
    finally:
        dt.close()


@needs_jit
def test_bit_sliced_predicates_on_ragged_snapshots():
    """size() snapshots that end inside a lane's 32 rows, inside a wave's step, at 0 — the tail mask of the bit-sliced scan — and every relation."""
    rng = np.random.default_rng(17)
    n = 70_000
    tab = _table(rng, n, hi_a=16, hi_b=3000)
    dt = mirror_table(tab)
    try:
        dt.predpack([0, 1], sliced=True)
        for snap in ([n, n, n], [1, 31, 33], [2047, 2048, 2049], [0, n - 1, 4097], [65_537, 0, 8191]):
            for op_a, op_b in (("lt", "ge"), ("le", "gt"), ("eq", "ne"), ("ne", "lt"), ("ge", "le"), ("gt", "eq")):
                q = {"dimensions": ["g"], "metrics": ["v", "count"], "filter": {"op": "and", "filters": [F(op_a, "a", "9"), F(op_b, "b", "1000")]}}
                res, _ = run(tab, dt, q, flags=JIT | capi.PLAN_NO_LANES, seg_rows=snap)      # (the compacting form whatever passes: the no-compaction one reads rows)
                assert res.sliced, (snap, op_a, op_b, res.flags)
    finally:
        dt.close()

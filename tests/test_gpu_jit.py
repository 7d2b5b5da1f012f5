"""Scan kernels compiled per plan shape (viyadb_amd/csrc/vh_jit.hip + vh_jit_body.h — the GPU analogue of the reference's
generated query function, src/codegen/query/agg_query.cc:26-75 / filter.cc:206-261 / compiler.cc:97-144) against the oracle:
every table organisation, every predicate type and operator, nested filters, narrow copies and projections, ragged segments.
Small tables never get a compiled kernel unasked (the compile would cost more than the query): VH_PLAN_FORCE_JIT asks."""
import numpy as np
import pytest

from oracle import viya_oracle as vo
from tests.parity import check_workload, compare
from tests.planner import mirror_table, plan_from_query
from tests.test_gpu_typed import F, NOW, TYPES, run, typed_table
from viyadb_amd import capi

pytestmark = pytest.mark.gpu
J = capi.PLAN_FORCE_JIT


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


@pytest.mark.parametrize("flags,path", [(0, "dense_global"), (64, "dense_part"), (16, "dense_global"), (20, "dense_global"), (48, "dense_global"),
                                        (1, "hash"), (2, "dense_global"), (64 | 4 | 32, "dense_part"), (64 | 8192, "dense_part"), (16 | 8192, "dense_global"),
                                        (1 | 2048, "hash"), (1 | 2048 | 512, "hash"), (64 | 65536, "dense_part"), (1 | 8192, "hash")])
def test_c3_table_organisations_compiled(flags, path):
    from viyadb_amd import synth
    res, _ = check_workload(synth.c3(segment_rows=250_000), nseg=4, flags=flags | J, expect_path=path)
    assert res.jit and res.kernel.startswith("viya_jit_scan_"), res.kernel


@pytest.mark.parametrize("table_kb,flags", [(32, 64), (16, 64), (32, 64 | 8192), (8, 64), (8, 64 | 32)])
def test_c3_with_17_to_64_partitions_compiled(table_kb, flags):
    """Group-id spaces of 17-64 LDS-sized ranges (VH_PART_TABLE_KB shrinks the ranges: 38 / 75 -> two levels / 150 partitions' worth):
    the compiled phase 1 keeps waiting lines per partition for up to 64 of them (the block's ring writer, vj_part_ring_add with J::PART_RING = 64), one level or two."""
    import os
    from viyadb_amd import synth
    os.environ["VH_PART_TABLE_KB"] = str(table_kb)
    try:
        res, _ = check_workload(synth.c3(segment_rows=250_000), nseg=4, flags=flags | J, expect_path="dense_part")
    finally:
        del os.environ["VH_PART_TABLE_KB"]
    assert res.jit and res.kernel.startswith("viya_jit_scan_"), res.kernel


@pytest.mark.parametrize("flags,path", [(128, "dense_lds"), (128 | 2, "dense_global"), (128 | 1, "hash"), (128 | 4, "dense_lds")])
def test_c2_table_organisations_compiled(flags, path):
    """C2's 50 % selectivity would take the no-compaction kernels (pre-built only): VH_PLAN_NO_LANES keeps the compacting form."""
    from viyadb_amd import synth
    res, _ = check_workload(synth.c2(segment_rows=250_000), nseg=4, flags=flags | J, expect_path=path)
    assert res.jit, res.kernel


def test_c2_no_compaction_form_compiled():
    """C2 as the planner runs it at full size: 50 % of the rows pass -> the no-compaction form, compiled (J::LANES)."""
    from viyadb_amd import synth
    res, _ = check_workload(synth.c2(segment_rows=250_000), nseg=4, flags=J, expect_path="dense_lds")
    assert res.jit and res.lanes, res.kernel
    res, _ = check_workload(synth.c2(segment_rows=250_000), nseg=4, flags=J | 256 | 4, expect_path="dense_lds")      # forced, no XCD-private copies
    assert res.jit and res.lanes, res.kernel


@pytest.mark.parametrize("dims,metrics", [
    (["d_int"], ["long_sum", "int_min"]), (["d_long"], ["float_sum", "double_max"]), (["d_uint"], ["uint_max", "ulong_min"]),
    (["d_ulong"], ["count", "int_avg"]), (["d_int"], ["double_sum"]),
    (["d_int"], ["short_sum", "long_sum"]),        # a 2-byte metric column: only the compiled form loads it
    (["d_int", "flag"], ["long_sum"]),             # a 1-byte group column
    (["s8", "flag"], ["ubyte_max", "short_min"]),  # two 1-byte group columns (a dictionary code, a boolean), narrow metrics with and without sign
    (["d_short"], ["byte_sum", "ushort_max"]),
    (["d_int", "flag"], ["long_sum", "int_min", "count", "float_max"]),     # more columns than the pre-built kernel takes (2 + 2): 28 bytes of payload per row
    (["s8"], ["count", "int_sum", "uint_max", "short_min", "double_sum"])])
def test_no_compaction_form_compiled_on_every_width(typed, dims, metrics):
    """The compiled no-compaction kernel: group and metric columns of 1, 2, 4 and 8 bytes (one aligned load per column and sub-step), segment
    tails (40 000 rows per segment: the last step of each is partial), with a filter most rows pass, one few pass, and none."""
    tab, dt = typed
    for flt in (F("ge", "d_int", "-30"), F("lt", "d_uint", "3"), None):
        q = {"dimensions": dims, "metrics": metrics}
        if flt:
            q["filter"] = flt
        res, _ = run(tab, dt, q, flags=J | 256)
        assert res.path == "dense_lds" and res.jit and res.lanes, (dims, metrics, res.path, res.kernel)


@pytest.mark.parametrize("name", ["C1", "C2", "C3"])
@pytest.mark.parametrize("flags", [0, 128, 64])
def test_ragged_segments_compiled(name, flags):
    """Segment size not a multiple of anything; the last rows of every segment lie beyond size(): the step that reaches the end
    of a segment takes the kernel's masked path."""
    from viyadb_amd import synth
    check_workload(synth.WORKLOADS[name](segment_rows=100_003), nseg=4, rows_per_seg=99_991, flags=flags | J)
    for w in (synth.c2(segment_rows=1000), synth.c3(segment_rows=1000), synth.c1(segment_rows=1000)):
        check_workload(w, nseg=1, rows_per_seg=1, flags=flags | J)
        check_workload(w, nseg=3, rows_per_seg=63, flags=flags | J)


def test_c5_time_rollup_hash_path_compiled():
    from viyadb_amd import synth
    for flags in (0, 2048, 512):
        res, _ = check_workload(synth.c5t(segment_rows=150_000), nseg=3, flags=flags | J, expect_path="hash")
        assert res.jit and res.ngroups > 100_000


def test_bitset_metrics_keep_the_prebuilt_kernels():
    from viyadb_amd import synth
    res, _ = check_workload(synth.c5(segment_rows=60_000), nseg=2, flags=J, expect_path="hash")
    assert not res.jit


@pytest.mark.parametrize("t", [x for x in TYPES if x not in ("byte", "short")])
@pytest.mark.parametrize("op", ["eq", "ne", "lt", "le", "gt", "ge"])
def test_predicate_on_every_type_compiled(typed, t, op):
    """1-, 2-, 4- and 8-byte predicate columns, signed, unsigned and floating: packed in registers, compared in the column's own type."""
    tab, dt = typed
    val = "3.5" if t in ("float", "double") else "7"
    res, _ = run(tab, dt, {"dimensions": ["s8"], "metrics": ["count", "long_sum"], "filter": F(op, "d_" + t, val)}, flags=J)
    assert res.jit and res.fast


def test_nested_filters_in_lists_and_metric_predicates_compiled(typed):
    tab, dt = typed
    qs = [{"op": "or", "filters": [{"op": "and", "filters": [F("lt", "d_uint", "40"), F("gt", "d_int", "-40"), F("ne", "d_float", "1.5")]},
                                    {"op": "in", "column": "d_ushort", "values": ["3", "4", "5", "17"]},
                                    {"op": "not", "filter": {"op": "in", "column": "s8", "values": ["v1", "v2", "v150", "nope"]}}]},
          {"op": "and", "filters": [F("ge", "count", "2"), F("lt", "double_max", "100.5"), {"op": "or", "filters": [F("eq", "flag", "true"), F("gt", "ts", str(NOW - 86400 * 300))]}]},
          {"op": "not", "filter": {"op": "or", "filters": [F("eq", "d_long", "5"), F("le", "d_ulong", "3"), F("gt", "d_double", "2.25")]}},
          {"op": "in", "column": "d_int", "values": [str(v) for v in range(-12, 12)]}]
    for f in qs:
        for dims, flags in ((["s8", "flag"], 0), (["s16", "d_int"], 1), (["s32"], 64), ([], 0)):
            res, _ = run(tab, dt, {"dimensions": dims, "metrics": ["count", "long_sum", "double_max", "int_min"], "filter": f}, flags=flags | J)
            assert res.jit, (f, dims, flags, res.kernel)


@pytest.mark.parametrize("t", TYPES)
def test_all_aggregations_per_type_compiled(typed, t):
    tab, dt = typed
    for flags in (0, 1, 16, 64):
        res, _ = run(tab, dt, {"dimensions": ["s8", "flag"], "metrics": [f"{t}_sum", f"{t}_min", f"{t}_max", f"{t}_avg", "count"], "filter": F("lt", "d_int", "20")}, flags=flags | J)
        assert res.jit


@pytest.mark.parametrize("dims", [["d_byte", "d_float", "d_double"], ["d_ulong", "d_long"], ["s8", "s16", "s32", "flag", "d_short"],
                                  ["id"], ["d_short", "d_int"], ["d_ubyte", "d_ushort", "d_uint", "d_byte"]])
def test_group_key_shapes_compiled(typed, dims):
    """Wide and narrow keys, signed digits in 32-bit wrap-around arithmetic, float keys (-0.0 == 0.0), multi-word hash keys."""
    tab, dt = typed
    for flags in (0, 1):
        res, _ = run(tab, dt, {"dimensions": dims, "metrics": ["count", "double_sum"], "filter": F("gt", "d_int", "-30")}, flags=flags | J | capi.PLAN_NO_LANES)
        assert res.jit


@pytest.mark.parametrize("gran", ["year", "month", "day", "hour", "minute", "second"])
@pytest.mark.parametrize("col", ["ts", "uts"])
def test_query_granularity_compiled(typed, gran, col):
    tab, dt = typed
    q = {"type": "aggregate", "table": "t", "select": [{"column": col, "granularity": gran}, {"column": "s8"}, {"column": "count"}, {"column": "int_max"}],
         "filter": F("ge", "d_uint", "3")}
    res, _ = run(tab, dt, q, flags=J)
    assert res.jit


def test_narrow_copies_and_projection_compiled():
    """The C3 layout of the bench: 8- / 16-bit copies of the predicate columns (compared where they sit, SDWA selectors) and a survivor's
    payload out of one 32-byte record."""
    from viyadb_amd import synth
    from viyadb_amd.executor import AggPlan
    from tests.parity import build_oracle_table
    w = synth.c3(segment_rows=300_000)
    dt = synth.create_device_table(w, 3, 299_977, 0, 42)
    try:
        ot = build_oracle_table(w, 3, 299_977, 0, 42)
        st = vo.scan_aggregate(vo.parse_query(ot, w.query))
        for prep in ("none", "narrow", "pack", "both"):
            if prep in ("narrow", "both"):
                dt.narrow(dt.filter_columns(AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics)))
            if prep in ("pack", "both"):
                dt.pack(dt.gather_columns(AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics)))
            for flags in (0, 64, 16, 1):
                res = dt.query_agg(AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, flags=flags | J))
                compare(res, st, f"{prep} flags={flags}")
                assert res.jit
                # (the library builds both layouts by itself for a shape it keeps seeing: only what was asked for is asserted)
                assert (res.narrow or prep in ("none", "pack")) and (res.packed or prep in ("none", "narrow")), (prep, flags, res.narrow, res.packed)
    finally:
        dt.close()


def test_compiled_and_prebuilt_kernels_agree_bit_for_bit(typed):
    tab, dt = typed
    q = {"type": "aggregate", "table": "t", "select": [{"column": "s16"}, {"column": "d_int"}, {"column": "count"}, {"column": "long_sum"}, {"column": "uint_max"}],
         "filter": {"op": "and", "filters": [F("lt", "d_uint", "50"), F("ne", "d_int", "3")]}}
    aq = vo.parse_query(tab, q)
    a = dt.query_agg(plan_from_query(tab, aq, now=NOW, flags=J))
    b = dt.query_agg(plan_from_query(tab, aq, now=NOW, flags=capi.PLAN_NO_JIT))
    assert a.jit and not b.jit and a.ngroups == b.ngroups and a.passed_recs == b.passed_recs

    def canon(r):
        o = np.lexsort([k for k in reversed(r.keys)])
        return [k[o] for k in r.keys] + [x[o] for x in r.states]
    for x, y in zip(canon(a), canon(b)):
        assert np.array_equal(x, y)


def test_random_plans_take_the_compiled_kernel_when_eligible(typed):
    """the specialised drain for (nearly) every plan, not for two shapes. Random queries as in
    test_random_plans_against_oracle; every plan the pre-built register-resident kernels could run — and the ones only a compiled
    kernel can hold in registers (1-, 2-, 8-byte predicate columns) — must report a compiled kernel, results equal to the oracle's."""
    import random
    rnd = random.Random(4242)
    tab, dt = typed
    dims_pool = ["s8", "s16", "s32", "flag", "d_ubyte", "d_short", "d_ushort", "d_int", "d_uint", "d_long", "d_ulong", "d_float", "d_double", "id"]
    metric_pool = ["count"] + [f"{t}_{a}" for t in TYPES for a in ("sum", "min", "max", "avg")]
    filt_cols = [("d_int", lambda: rnd.randrange(-60, 61)), ("d_uint", lambda: rnd.randrange(0, 61)), ("d_long", lambda: rnd.randrange(-60, 61)),
                 ("d_ulong", lambda: rnd.randrange(0, 61)), ("d_float", lambda: rnd.randrange(-200, 200) / 8.0), ("d_double", lambda: rnd.randrange(-200, 200) / 8.0),
                 ("d_ubyte", lambda: rnd.randrange(0, 61)), ("d_ushort", lambda: rnd.randrange(0, 61)), ("s8", lambda: "v%d" % rnd.randrange(0, 160)),
                 ("flag", lambda: rnd.choice(["true", "false"])), ("ts", lambda: NOW - rnd.randrange(0, 2 * 365 * 86400)),
                 ("count", lambda: rnd.randrange(1, 4)), ("int_sum", lambda: rnd.randrange(-2 ** 19, 2 ** 19))]

    def leaf():
        col, gen = rnd.choice(filt_cols)
        if rnd.random() < 0.2:
            return {"op": "in", "column": col, "values": [str(gen()) for _ in range(rnd.randrange(1, 5))]}
        op = rnd.choice(["eq", "ne", "lt", "le", "gt", "ge"]) if col not in ("s8", "flag") else rnd.choice(["eq", "ne"])
        return F(op, col, gen())

    def tree(depth):
        if depth == 0 or rnd.random() < 0.35:
            f = leaf()
        else:
            f = {"op": rnd.choice(["and", "or"]), "filters": [tree(depth - 1) for _ in range(rnd.randrange(2, 4))]}
        return {"op": "not", "filter": f} if rnd.random() < 0.15 else f

    width = {"d_int": 4, "d_uint": 4, "d_long": 8, "d_ulong": 8, "d_float": 4, "d_double": 8, "d_ubyte": 1, "d_ushort": 2, "s8": 1, "flag": 1, "ts": 4, "count": 4, "int_sum": 4}

    def filter_columns(f, acc):
        if "filters" in f:
            for x in f["filters"]:
                filter_columns(x, acc)
        elif "filter" in f:
            filter_columns(f["filter"], acc)
        else:
            acc.add(f["column"])
        return acc

    ran = eligible = compiled = 0
    missed = []
    for _ in range(36):
        sel = [{"column": d} for d in rnd.sample(dims_pool, rnd.randrange(0, 4))]
        if rnd.random() < 0.3:
            sel.append({"column": "ts", "granularity": rnd.choice(["year", "month", "day", "hour", "minute"])})
        ms = rnd.sample(metric_pool, rnd.randrange(1, 5))
        if any(m.endswith("_avg") for m in ms) and "count" not in ms:
            ms.append("count")
        sel += [{"column": m} for m in ms]
        rnd.shuffle(sel)
        q = {"select": sel, "filter": tree(2)}
        flags = rnd.choice([0, 0, 1, 16, 64, 8192, 8192 | 64, 1 | 2048, 65536]) | capi.PLAN_NO_LANES
        try:
            res, _ = run(tab, dt, q, flags=flags | J)
        except vo.Unsupported:
            continue
        ran += 1
        # what a compiled kernel holds in registers: 16 rows of every distinct predicate column per lane, at most 96 registers (VJ_MAX_NV)
        fits = sum(4 * width[c] for c in filter_columns(q["filter"], set())) <= 96
        eligible += fits
        compiled += bool(res.jit) and fits
        if fits and not res.jit:
            missed.append((flags, res.path, res.kernel, q))
        assert fits or not res.jit
    assert ran >= 25 and eligible >= 15 and compiled >= 0.9 * eligible, (ran, eligible, compiled, missed)

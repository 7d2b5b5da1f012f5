"""Skewed shapes of the bench's queries (tools/skew_probe.py measures them at full size): a Zipf-like group key, a table
loaded in the order of a predicate column (the reference's own scenario: test/index.cc:44-75), a hot composite key under the hashed
partitioning. Each against the oracle on identical generated rows (the numpy twin of the generator modes is compared column by column
first), through the planner's own choice and through the partitioning organisations forced, compiled and pre-built kernels."""
import pytest

from tests.conftest import JIT_OFF
from tests.parity import check_workload
from viyadb_amd import capi, synth

pytestmark = pytest.mark.gpu

FORCE_PART = 64


@pytest.fixture(scope="module", autouse=True)
def _init():
    from viyadb_amd import executor
    executor.init(0)


@pytest.mark.parametrize("flags", [0, FORCE_PART, FORCE_PART | capi.PLAN_FORCE_JIT, FORCE_PART | capi.PLAN_NO_JIT, 1])
def test_zipf_group_key(flags):
    w = synth.c3z(segment_rows=200_000)
    res, st = check_workload(w, nseg=6, flags=flags)
    hot = res.keys[0] == 0                      # d0 = 0 holds a tenth of the rows: its 100 groups a tenth of the survivors
    assert 0.07 < res.states[1][hot].sum() / res.states[1].sum() < 0.13


@pytest.mark.parametrize("flags", [0, FORCE_PART | capi.PLAN_FORCE_JIT, FORCE_PART | capi.PLAN_NO_JIT])
def test_table_loaded_in_predicate_order(flags):
    w = synth.c3s(segment_rows=100_000, total_segments=20)        # 2 000 rows per value of d3: 50 values per segment, `d3 < 447` = the first 9 segments
    res, st = check_workload(w, nseg=20, flags=flags)
    assert res.scanned_segments == 9 and res.scanned_recs == 20 * 100_000        # 11 segments skipped by min / max, all counted as scanned rows (scan.cc:44-51)


HP = 1 | (1 << 18) | (1 << 20)      # hash organisation, compiled scan, hashed partitioning


@pytest.mark.parametrize("flags", [0, HP])
def test_hot_composite_key_with_distinct_counts(flags):
    w = synth.c5h(segment_rows=60_000)
    res, st = check_workload(w, nseg=4, flags=flags | capi.PLAN_CARD32)
    top = int(res.states[1].argmax())
    assert res.states[1][top] > 0.15 * res.states[1].sum()        # the hot (t, u): a tenth of the rows, a fifth of the survivors (u < 500 000 keeps half of the rest)


def test_heavy_ranges_of_the_hashed_partitioning_take_a_second_pass(monkeypatch):
    """C5h through the hashed partitioning: the hot (t, u) group's range holds a hundred times a range's share of the tuples and more ids than a
    block's LDS set takes. The ranges' kernel marks it heavy instead of voiding the attempt, the host runs the plain hash organisation over the rows
    of exactly that range (the generic scan drops every other survivor behind its key) and appends its groups: one query, two passes, the oracle's
    rows — the hot group's COUNT and COUNT DISTINCT included. Without the second pass (VH_NO_HEAVY_PASS) the whole query ends on the plain table."""
    if JIT_OFF:
        pytest.skip("the ring writer and the hashed partitioning live in the compiled kernels (VH_JIT=off: the pre-built ones answer)")
    w = synth.c5h(segment_rows=80_000)
    res, st = check_workload(w, nseg=4, flags=HP | capi.PLAN_CARD32)
    assert res.hpart and "_hpagg" in res.kernel and res.retries == 1, (res.hpart, res.retries, res.kernel)
    top = int(res.states[1].argmax())
    assert int(res.states[1][top]) > 30_000 and int(res.states[0][top]) > 30_000          # rows and distinct users of the hot group
    # twice more on one table (the bitmap of heavy ranges is cleared per query), and the uniform twin takes no second pass
    from tests.parity import build_oracle_table, compare
    from oracle import viya_oracle as vo
    from viyadb_amd.executor import AggPlan
    dt = synth.create_device_table(w, 4, 80_000)
    try:
        want = vo.scan_aggregate(vo.parse_query(build_oracle_table(w, 4, 80_000), w.query), now=w.now)
        for _ in range(3):
            r = dt.query_agg(AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, flags=HP | capi.PLAN_CARD32))
            compare(r, want, "heavy ranges, repeated")
            assert r.hpart and r.retries == 1
    finally:
        dt.close()
    res, st = check_workload(synth.c5(segment_rows=80_000), nseg=4, flags=HP | capi.PLAN_CARD32)
    assert res.hpart and res.retries == 0
    monkeypatch.setenv("VH_NO_HEAVY_PASS", "1")
    res, st = check_workload(w, nseg=4, flags=HP | capi.PLAN_CARD32)
    assert not res.hpart and res.retries >= 1 and res.path == "hash"


def _c5h_variant(kind, segment_rows):
    """C5h as it is (packed 16-byte tuples), with three ids per row (a row's further ids travel in "ids only" tuples), or without the bitset metric
    (plain 16-byte tuples: C5t with the hot key)."""
    from viyadb_amd.synth import SynthColumn
    w = synth.c5h(segment_rows=segment_rows)
    if kind == "ids3":
        u = w.columns[3]
        w.columns[3] = SynthColumn(u.name, u.kind, u.elem, (u.gen[0], u.gen[1], 3, u.gen[3]), u.json_type)
    elif kind == "count_only":
        h = w
        w = synth.c5t(segment_rows=segment_rows)
        w.columns[0], w.columns[1] = h.columns[0], h.columns[1]
    return w


@pytest.mark.parametrize("kind,flags", [("packed", 0), ("unpacked", capi.PLAN_NO_HP_PACK), ("ids3", 0), ("ids3", capi.PLAN_NO_HP_PACK), ("count_only", 0)])
def test_heavy_partitions_second_pass_reads_the_first_passs_tuples(kind, flags, monkeypatch):
    """A level-A partition that hp_plan_kernel leaves out whole (the hot key's) is aggregated from its tuples in pool a — every tuple form: packed
    16-byte, 32-byte with words for the ids, "ids only" tuples of rows with more than two ids, plain (mixed key, payload) — into the plain hash
    organisation's table (hp_heavy_tuples_kernel, named in the result's kernel string); the same rows with the table scanned again (VH_HEAVY_RESCAN)."""
    if JIT_OFF:
        pytest.skip("the ring writer and the hashed partitioning live in the compiled kernels (VH_JIT=off: the pre-built ones answer)")
    w = _c5h_variant(kind, 80_000)
    res, st = check_workload(w, nseg=4, flags=HP | capi.PLAN_CARD32 | flags)
    assert res.hpart and res.retries == 1 and "hp_heavy_tuples_kernel" in res.kernel, (res.hpart, res.retries, res.kernel)
    monkeypatch.setenv("VH_HEAVY_RESCAN", "1")
    res, st = check_workload(w, nseg=4, flags=HP | capi.PLAN_CARD32 | flags)
    assert res.hpart and res.retries == 1 and "hp_heavy_tuples_kernel" not in res.kernel and "scan_agg_kernel" in res.kernel, (res.hpart, res.retries, res.kernel)


def test_heavy_second_pass_under_a_big_results_copy(monkeypatch):
    """A result too big for one-shot delivery (1.2 M groups): its rows are packed into the staging buffer over PCIe WHILE the second pass scans —
    the buffer laid out for the rows the pass can add at most —, and the pass's rows are appended behind them. The same answer with the pass waited
    for first (VH_NO_HEAVY_OVERLAP), twice on one table."""
    if JIT_OFF:
        pytest.skip("the ring writer and the hashed partitioning live in the compiled kernels (VH_JIT=off: the pre-built ones answer)")
    from tests.parity import build_oracle_table, compare
    from oracle import viya_oracle as vo
    from viyadb_amd.executor import AggPlan
    w = synth.c5h(segment_rows=600_000)
    dt = synth.create_device_table(w, 4, 600_000)
    try:
        want = vo.scan_aggregate(vo.parse_query(build_oracle_table(w, 4, 600_000), w.query), now=w.now)
        plan = AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, flags=HP | capi.PLAN_CARD32)
        for overlap in (True, True, False):
            if not overlap:
                monkeypatch.setenv("VH_NO_HEAVY_OVERLAP", "1")
            r = dt.query_agg(plan)
            compare(r, want, f"heavy ranges under a big result (overlap {overlap})")
            assert r.hpart and r.retries == 1 and r.ngroups > 1_000_000, (r.hpart, r.retries, r.ngroups)
    finally:
        dt.close()


def test_heavy_second_pass_from_several_threads():
    """The second pass takes an execution context of its own while the query holds one: four threads on one table, each with its own result alive,
    every answer the oracle's."""
    if JIT_OFF:
        pytest.skip("the ring writer and the hashed partitioning live in the compiled kernels (VH_JIT=off: the pre-built ones answer)")
    import threading
    from tests.parity import build_oracle_table, compare
    from oracle import viya_oracle as vo
    from viyadb_amd.executor import AggPlan
    w = synth.c5h(segment_rows=80_000)
    dt = synth.create_device_table(w, 4, 80_000)
    errs = []
    try:
        want = vo.scan_aggregate(vo.parse_query(build_oracle_table(w, 4, 80_000), w.query), now=w.now)
        plan = AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, flags=HP | capi.PLAN_CARD32)
        dt.query_agg(plan)

        def loop(i):
            try:
                for _ in range(4):
                    r = dt.query_agg(plan)
                    compare(r, want, f"heavy ranges, thread {i}")
                    assert r.hpart and r.retries == 1
            except Exception as e:   # noqa: BLE001
                errs.append(repr(e)[:500])
        ths = [threading.Thread(target=loop, args=(i,)) for i in range(4)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
    finally:
        dt.close()
    assert not errs, errs


def test_heavy_second_pass_without_a_spare_context():
    """VH_MAX_EXEC=1 (a fresh process: the knob is read once): the query holds the table's only execution context, the second pass does not wait
    for another — the whole query is planned again for the plain hash table and still answers with the oracle's rows."""
    if JIT_OFF:
        pytest.skip("the ring writer and the hashed partitioning live in the compiled kernels (VH_JIT=off: the pre-built ones answer)")
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from viyadb_amd import capi, executor, synth\n"
        "executor.init(0)\n"
        "from tests.parity import check_workload\n"
        "res, st = check_workload(synth.c5h(segment_rows=80_000), nseg=4, flags=1 | capi.PLAN_FORCE_HPART | capi.PLAN_FORCE_JIT | capi.PLAN_CARD32)\n"
        "assert res.path == 'hash' and not res.hpart and res.retries >= 1, (res.path, res.hpart, res.retries, res.kernel)\n"
        "print('ok')\n" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, VH_MAX_EXEC="1", VH_TEST_HOOKS="1"), cwd=root)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("levels", ["0", "1"])
def test_every_tuple_through_the_overflow_region(levels, monkeypatch):
    """VH_TEST_POS_LEVELS: the ring writer's streams get no (or one) positional extent, so phase 1 of DENSE_PART takes (nearly) all its extents
    from the pool's shared overflow region through the block's LDS table — what a hot partition does, at a size a test affords. One- and
    two-word tuples, 13 and 1 partitions' worth of skew; C5's scan-written level A and its level B the same way."""
    if JIT_OFF:
        pytest.skip("the ring writer and the hashed partitioning live in the compiled kernels (VH_JIT=off: the pre-built ones answer)")
    monkeypatch.setenv("VH_TEST_POS_LEVELS", levels)
    for wl, flags in (("C3z", FORCE_PART | capi.PLAN_FORCE_JIT), ("C3", FORCE_PART | capi.PLAN_FORCE_JIT | capi.PLAN_NO_NARROW_TUPLES), ("C3s", FORCE_PART | capi.PLAN_FORCE_JIT)):
        w = synth.WORKLOADS[wl](segment_rows=150_000) if wl != "C3s" else synth.c3s(150_000, 8)
        res, st = check_workload(w, nseg=8, flags=flags)
        assert res.path == "dense_part" and res.retries == 0, (wl, res.path, res.retries, res.kernel)
    w = synth.c5(segment_rows=60_000)      # (C5h's hot group holds more ids than a range's LDS set: it ends on the plain hash table — test_hot_composite_key_with_distinct_counts)
    res, st = check_workload(w, nseg=4, flags=HP | capi.PLAN_CARD32)
    assert res.retries == 0 and res.hpart, (res.retries, res.kernel)


def test_ring_writer_stress_two_partitions_every_row_passing(monkeypatch):
    """two LDS-sized ranges and no filter — every drain of 64 survivors puts ~32 tuples into each of two partitions, four
    times what a waiting line holds, so every call of the ring writer goes through several rounds of its wait loop with owners flushing
    while later lanes of the same call still wait — looped 1 000 times per tuple size (8- and 16-byte tuples) on scratch memory that is
    poisoned before use (VH_POISON): every answer must be the first one's, bit for bit, and the first one the oracle's."""
    if JIT_OFF:
        pytest.skip("the ring writer and the hashed partitioning live in the compiled kernels (VH_JIT=off: the pre-built ones answer)")
    import numpy as np
    from oracle import viya_oracle as vo
    from tests.planner import mirror_table, plan_from_query
    from tests.parity import compare
    monkeypatch.setenv("VH_POISON", "1")
    rng = np.random.default_rng(99)
    n = 250_000
    tab = vo.Table({"name": "t", "segment_size": n, "dimensions": [{"name": "a", "type": "uint"}, {"name": "b", "type": "uint"}],
                    "metrics": [{"name": "v", "type": "long_sum"}, {"name": "count", "type": "count"}]})
    for _ in range(4):
        tab.add_segment_arrays([rng.integers(0, 150, n).astype(np.uint32), rng.integers(0, 100, n).astype(np.uint32)],
                               [rng.integers(0, 1000, n).astype(np.int64), np.ones(n, dtype=np.uint32)], None, n)
    dt = mirror_table(tab)
    aq = vo.parse_query(tab, {"type": "aggregate", "table": "t", "dimensions": ["a", "b"], "metrics": ["v", "count"]})
    st = vo.scan_aggregate(aq, now=1496570140)
    try:
        for flags in (FORCE_PART | capi.PLAN_FORCE_JIT | capi.PLAN_NO_LANES, FORCE_PART | capi.PLAN_FORCE_JIT | capi.PLAN_NO_LANES | capi.PLAN_NO_NARROW_TUPLES):
            plan = plan_from_query(tab, aq, now=1496570140, flags=flags)
            first = dt.query_agg(plan)
            compare(first, st, "stress, flags %d" % flags)
            assert first.path == "dense_part" and first.jit and "viya_jit_scan" in first.kernel and first.retries == 0, (first.path, first.kernel)
            o = np.lexsort([first.keys[1], first.keys[0]])
            want = [x[o] for x in first.keys + first.states]
            for it in range(1000):
                r = dt.query_agg(plan)
                assert r.retries == 0 and r.ngroups == first.ngroups, (it, r.retries, r.ngroups)
                o = np.lexsort([r.keys[1], r.keys[0]])
                for a, b in zip(want, [x[o] for x in r.keys + r.states]):
                    assert np.array_equal(a, b), (flags, it)
    finally:
        dt.close()


def test_phase_2_blocks_follow_the_partitions_tuple_counts(monkeypatch):
    """C3z at a size where DENSE_PART's private copies are merged by the merge kernel (100 K groups): 62 % of the tuples land in one of 13 partitions,
    and phase 2 shares its blocks — and the private table copies — out by the counts phase 1 took (vh_part_shares). Same answer as the oracle with the
    shares on, with them off (VH_NO_PART_BALANCE: every partition the same number of blocks), with MIN / MAX states whose copies are never pre-filled,
    and three times over on one table (copies beyond a partition's share hold an earlier query's states: the merge must not read them)."""
    from oracle import viya_oracle as vo
    from tests.parity import build_oracle_table, compare
    from viyadb_amd.executor import AggPlan
    w = synth.c3z(segment_rows=200_000)
    dt = synth.create_device_table(w, 6, 200_000)
    try:
        ot = build_oracle_table(w, 6, 200_000)
        for metrics, mnames in (([7, 9], ["m0", "count"]), ([8, 11, 9], ["m1", "m4", "count"])):       # SUM + COUNT; MAX(long) + MIN(uint) + COUNT
            q = dict(w.query, metrics=mnames)
            st = vo.scan_aggregate(vo.parse_query(ot, q))
            for filt in (w.plan.filter, [("rel", 2, capi.OP_EQ, 1)], w.plan.filter):      # different selectivities leave different shares behind
                qq = dict(q) if filt is w.plan.filter else dict(q, filter={"op": "eq", "column": "d2", "value": "1"})
                want = st if filt is w.plan.filter else vo.scan_aggregate(vo.parse_query(ot, qq))
                for env in (None, "1"):
                    if env:
                        monkeypatch.setenv("VH_NO_PART_BALANCE", env)
                    else:
                        monkeypatch.delenv("VH_NO_PART_BALANCE", raising=False)
                    res = dt.query_agg(AggPlan(filter=filt, groups=w.plan.groups, metrics=metrics, flags=FORCE_PART | capi.PLAN_FORCE_JIT, groups_hint=100_000))
                    compare(res, want, "balanced phase 2, metrics %s, env %s" % (mnames, env))
                    assert res.path == "dense_part" and res.retries == 0
    finally:
        monkeypatch.delenv("VH_NO_PART_BALANCE", raising=False)
        dt.close()


def test_four_byte_tuples_through_two_levels(monkeypatch):
    """More than 64 LDS-sized ranges (VH_PART_TABLE_KB shrinks them: C3's 100 K groups in ~200 ranges, four level-1 partitions): the four-byte tuple
    carries the gid RELATIVE to its level-1 partition, the second split moves four-byte words through the ring writer (part_split_ring_kernel<256, 4>)
    and the compiled phase 2 adds the range's place inside the partition. Same groups as with the 8-byte tuple (VH_NO_TUPLE4_TWO) and the oracle's,
    uniform and Zipf-like keys, and with every tuple of both levels through the overflow regions."""
    if JIT_OFF:
        pytest.skip("packed tuples need the compiled kernels")
    monkeypatch.setenv("VH_PART_TABLE_KB", "8")
    for wl in ("C3", "C3z"):
        w = synth.WORKLOADS[wl](segment_rows=150_000)
        res, st = check_workload(w, nseg=8, flags=FORCE_PART | capi.PLAN_FORCE_JIT)
        assert res.path == "dense_part" and res.retries == 0 and "part_split_ring_kernel<256, 4>" in res.kernel, (res.path, res.retries, res.kernel)
        monkeypatch.setenv("VH_NO_TUPLE4_TWO", "1")
        res8, _ = check_workload(w, nseg=8, flags=FORCE_PART | capi.PLAN_FORCE_JIT)
        monkeypatch.delenv("VH_NO_TUPLE4_TWO")
        assert "part_split_ring_kernel<256, 8>" in res8.kernel, res8.kernel
        monkeypatch.setenv("VH_TEST_POS_LEVELS", "0")
        res, st = check_workload(w, nseg=8, flags=FORCE_PART | capi.PLAN_FORCE_JIT)
        monkeypatch.delenv("VH_TEST_POS_LEVELS")
        assert "part_split_ring_kernel<256, 4>" in res.kernel


def test_four_byte_tuples(monkeypatch):
    """C3's gid (17 bits) and values (10 + 2 bits) fit 32 bits: DENSE_PART's tuple is four bytes — thirty-two to a 128-byte line through the ring
    writer, read back as 32-bit words by the compiled phase 2. Same groups as with the 8-byte tuple (VH_NO_TUPLE4) and as the oracle's, uniform and
    Zipf-like keys, with every tuple through the overflow region too; a value that outgrows its recorded bits re-plans (VH_ERR_HP_WIDE) as with any
    packed tuple."""
    if JIT_OFF:
        pytest.skip("packed tuples need the compiled kernels")
    for wl in ("C3", "C3z"):
        w = synth.WORKLOADS[wl](segment_rows=150_000)
        res, st = check_workload(w, nseg=8, flags=FORCE_PART | capi.PLAN_FORCE_JIT)
        assert res.path == "dense_part" and res.retries == 0 and (res.flags & 1024), (res.path, res.retries, res.flags)
        monkeypatch.setenv("VH_NO_TUPLE4", "1")
        res8, _ = check_workload(w, nseg=8, flags=FORCE_PART | capi.PLAN_FORCE_JIT)
        monkeypatch.delenv("VH_NO_TUPLE4")
        assert res8.kernel != res.kernel                      # another compiled shape
        monkeypatch.setenv("VH_TEST_POS_LEVELS", "0")
        check_workload(w, nseg=8, flags=FORCE_PART | capi.PLAN_FORCE_JIT)
        monkeypatch.delenv("VH_TEST_POS_LEVELS")

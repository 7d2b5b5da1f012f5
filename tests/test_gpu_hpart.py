"""Hashed partitioning of the hash path (viyadb_amd/csrc/vh_hpart.h) against the oracle on tables small enough for it: the path
is only taken unasked from 8 M survivors and 2 M groups on, so VH_PLAN_FORCE_HPART asks. What it stands in for is
`agg_map[agg_tuple.d].Update(agg_tuple.m)` (src/codegen/query/scan.cc:174-177,242) and, for a bitset metric,
`_j |= metrics._j` + cardinality() (src/codegen/db/store.cc:153-155, src/util/bitset.h:26-67) — so every case is the same
query through the oracle. Covered: rows with no, one, two and more than two ids (the tuple carries two; the rest follow in
ids-only tuples), plans without a bitset metric (16-byte tuples), time-truncated keys, SUM / MIN / MAX payloads, ragged
snapshots, LDS tables that overflow (re-plan with more passes, then without partitioning), pools that run out of extents,
skew (one group gets everything), and the full-size C5 shape in tests/test_gpu_fullsize.py."""
import os

import numpy as np
import pytest

from oracle import viya_oracle as vo
from tests.planner import mirror_table
from tests.test_gpu_typed import F, run
from viyadb_amd import capi

pytestmark = pytest.mark.gpu
HP = capi.PLAN_FORCE_HASH | capi.PLAN_FORCE_HPART | capi.PLAN_FORCE_JIT


@pytest.fixture(scope="module", autouse=True)
def _init():
    from viyadb_amd import executor
    executor.init(0)
    from tests.conftest import JIT_OFF
    if JIT_OFF:
        pytest.skip("VH_JIT=off: the hashed partitioning needs the scan kernel compiled for the plan")


def sets_table(max_ids, n=30_000, nseg=3, seed=11, id_space=5000, ncode=40, nx=100):
    rng = np.random.default_rng(seed)
    tab = vo.Table({"name": "t", "segment_size": n,
                    "dimensions": [{"name": "c", "type": "ushort"}, {"name": "x", "type": "uint"}, {"name": "ts", "type": "time", "format": "posix"}],
                    "metrics": [{"name": "users", "type": "bitset", "max": 2 ** 31}, {"name": "count", "type": "count"},
                                {"name": "v", "type": "int_sum"}, {"name": "lo", "type": "int_min"}]})
    for _ in range(nseg):
        sizes = rng.integers(0, max_ids + 1, n)
        sets = [set(int(v) for v in rng.integers(0, id_space, k)) for k in sizes]
        tab.add_segment_arrays([rng.integers(0, ncode, n).astype(np.uint16), rng.integers(0, nx, n).astype(np.uint32),
                                (1496000000 + rng.integers(0, 40 * 86400, n)).astype(np.uint32)],
                               [sets, np.ones(n, dtype=np.uint32), rng.integers(-1000, 1000, n).astype(np.int32), rng.integers(-10 ** 6, 10 ** 6, n).astype(np.int32)],
                               None, n)
    return tab


@pytest.fixture(scope="module")
def sets5():
    tab = sets_table(5)
    dt = mirror_table(tab)
    yield tab, dt
    dt.close()


def took_hpart(res):
    assert res.path == "hash" and res.hpart and res.jit and "scatter_kernel" in res.kernel and "_hpagg" in res.kernel, (res.path, res.kernel)


def scan_wrote_level_a(res):
    """The scan kernel partitioned 256 ways by itself (vj_fan_add): ONE scatter launch behind it instead of two, the barrier-free one."""
    return res.kernel.count("scatter_kernel") == 1 and "hp_ring_scatter_kernel" in res.kernel


@pytest.mark.parametrize("pack", [True, False])
@pytest.mark.parametrize("max_ids", [0, 1, 2, 3, 7])
def test_rows_with_any_number_of_ids(max_ids, pack):
    """0..max_ids ids per row: a tuple carries two of them and says how many count; rows with more send ids-only tuples. Both tuple
    forms: packed (16 bytes: payload, two ids and their count in the second word — COUNT is 1 and the ids are below 5000 here) and
    with words of their own for the ids (32 bytes, VH_PLAN_NO_HP_PACK)."""
    tab = sets_table(max_ids, n=20_000, nseg=2, seed=20 + max_ids)
    dt = mirror_table(tab)
    try:
        res, st = run(tab, dt, {"dimensions": ["c", "x"], "metrics": ["users", "count"], "filter": F("lt", "x", "70")}, flags=HP | (0 if pack else capi.PLAN_NO_HP_PACK))
        took_hpart(res)
        assert res.hp_packed == pack and ("scatter_kernel<1024, 1>" in res.kernel) == pack and ("scatter_kernel<1024, 2>" in res.kernel) == (not pack), res.kernel
        assert res.retries == 0 and res.ngroups == st.ngroups > 2000 and scan_wrote_level_a(res), res.kernel
        res, _ = run(tab, dt, {"dimensions": ["c"], "metrics": ["users"]}, flags=HP)          # few groups, many ids each: sets fill up -> more passes
        assert res.path == "hash"
    finally:
        dt.close()


@pytest.mark.parametrize("id_space,packed", [(2 ** 30, True), (2 ** 31, False)])
def test_packed_tuples_at_the_widest_ids_that_fit(id_space, packed):
    """Two ids of 30 bits + a one-bit COUNT fill the second word exactly (61 bits + the count of ids); 31-bit ids do not fit and keep words
    of their own. A non-negative SUM payload packs next to them at the width its maximum needs."""
    rng = np.random.default_rng(77)
    n = 20_000
    tab = vo.Table({"name": "t", "segment_size": n, "dimensions": [{"name": "c", "type": "ushort"}, {"name": "x", "type": "uint"}],
                    "metrics": [{"name": "users", "type": "bitset", "max": 2 ** 31}, {"name": "count", "type": "count"}, {"name": "v", "type": "uint_sum"}]})
    for _ in range(2):
        sets = [set(int(v) for v in rng.integers(id_space - 5000, id_space, k)) for k in rng.integers(0, 4, n)]
        tab.add_segment_arrays([rng.integers(0, 40, n).astype(np.uint16), rng.integers(0, 100, n).astype(np.uint32)],
                               [sets, np.ones(n, dtype=np.uint32), rng.integers(0, 1000, n).astype(np.uint32)], None, n)
    dt = mirror_table(tab)
    try:
        res, st = run(tab, dt, {"dimensions": ["c", "x"], "metrics": ["users", "count"]}, flags=HP)
        took_hpart(res)
        assert res.hp_packed == packed, res.kernel
        assert res.retries == 0 and res.ngroups == st.ngroups == 4000 and int(res.states[0].max()) >= 5
        if id_space == 2 ** 30:          # 10 bits of SUM payload on top: 1 + 10 + 60 > 61 -> words of their own again
            res, _ = run(tab, dt, {"dimensions": ["c", "x"], "metrics": ["users", "count", "v"]}, flags=HP)
            took_hpart(res)
            assert not res.hp_packed and "scatter_kernel<1024, 2>" in res.kernel, res.kernel
    finally:
        dt.close()


def test_packed_payload_next_to_small_ids(sets5):
    """COUNT + a non-negative SUM in the packed word; a SUM with negative values (or a MIN of them) keeps the 32-byte form."""
    tab, dt = sets5
    res, _ = run(tab, dt, {"dimensions": ["c", "x"], "metrics": ["users", "count"], "filter": F("ge", "x", "10")}, flags=HP)
    took_hpart(res)
    assert res.hp_packed
    res, _ = run(tab, dt, {"dimensions": ["c", "x"], "metrics": ["users", "count", "v"], "filter": F("ge", "x", "10")}, flags=HP)
    took_hpart(res)
    assert not res.hp_packed


def test_ids_beyond_the_recorded_range_void_the_packed_attempt(sets5):
    """The packed word is sized from the segments' recorded largest id. Should data and record ever disagree (forced here: the planner is
    told 4 bits), the scan flags VH_ERR_HP_WIDE instead of cutting ids short, and the query answers from the plain hash table."""
    tab, dt = sets5
    os.environ["VH_TEST_HP_IDBITS"] = "4"
    try:
        res, st = run(tab, dt, {"dimensions": ["c", "x"], "metrics": ["users", "count"], "filter": F("lt", "x", "60")}, flags=HP)
    finally:
        del os.environ["VH_TEST_HP_IDBITS"]
    assert res.path == "hash" and not res.hpart and res.retries >= 1 and res.ngroups == st.ngroups


def test_cardinalities_as_uint32_when_asked(sets5):
    """VH_PLAN_CARD32: a count distinct over 32-bit ids arrives as a uint32 column (vh_result_state_elem says so) with the same values —
    through the hashed partitioning, the plain hash table and its device-wide (group, id) set alike."""
    tab, dt = sets5
    q = {"dimensions": ["c", "x"], "metrics": ["users", "count"], "filter": F("lt", "x", "60")}
    for flags in (HP, capi.PLAN_FORCE_HASH | capi.PLAN_NO_HPART, 0):
        wide, _ = run(tab, dt, q, flags=flags)
        narrow, _ = run(tab, dt, q, flags=flags | capi.PLAN_CARD32)
        assert wide.states[0].dtype == np.uint64 and narrow.states[0].dtype == np.uint32
        assert np.array_equal(np.sort(wide.states[0]), np.sort(narrow.states[0].astype(np.uint64)))


def test_payloads_and_key_shapes(sets5):
    tab, dt = sets5
    for q in ({"dimensions": ["c", "x"], "metrics": ["users", "count", "v"], "filter": F("ge", "x", "10")},      # two 32-bit states + the set
              {"dimensions": ["x", "c"], "metrics": ["lo", "users"]},                                              # MIN in the payload word, no filter
              {"dimensions": ["c", "x"], "metrics": ["count", "v"], "filter": F("lt", "x", "50")},                 # no bitset metric: 16-byte tuples
              {"dimensions": ["c", "x"], "metrics": ["lo"], "filter": F("ne", "c", "3")},
              {"select": [{"column": "ts", "granularity": "hour"}, {"column": "c"}, {"column": "users"}, {"column": "count"}]},
              {"select": [{"column": "ts", "granularity": "day"}, {"column": "x"}, {"column": "count"}], "filter": F("lt", "x", "30")}):
        res, st = run(tab, dt, q, flags=HP)
        took_hpart(res)
        assert res.ngroups == st.ngroups
    # the same answers as the plain hash table and as the device-wide (group, id) set, bit for bit (compare() checked both against the oracle)
    run(tab, dt, {"dimensions": ["c", "x"], "metrics": ["users", "count", "v"], "filter": F("ge", "x", "10")}, flags=capi.PLAN_FORCE_HASH | capi.PLAN_NO_HPART)


def test_ragged_snapshots(sets5):
    tab, dt = sets5
    q = {"dimensions": ["c", "x"], "metrics": ["users", "count"], "filter": F("lt", "x", "90")}
    for snap in ([30_000, 1, 29_999], [0, 12_345, 0], [63, 64, 65]):
        res, _ = run(tab, dt, q, flags=HP, seg_rows=snap)
        took_hpart(res)


def test_lds_tables_that_overflow_are_replanned(sets5):
    """A range whose (group slot, id) set cannot hold its ids flags VH_ERR_HPART_FULL: the query runs again with more passes per range
    and, when that does not help either (40 groups with ~3 000 distinct ids each: sub-ranges never split a group), without partitioning
    — the rows are the oracle's either way."""
    tab, dt = sets5
    res, st = run(tab, dt, {"dimensions": ["c"], "metrics": ["users", "count"]}, flags=HP)
    assert res.path == "hash" and res.retries >= 1 and res.ngroups == st.ngroups == 40
    assert int(res.states[0].max()) > 2000           # (what did not fit: the set is sized for a 65 536th of the table's ids)
    q = {"dimensions": ["c", "x"], "metrics": ["users", "count"]}
    os.environ["VH_TEST_HPART_PASSES"] = "4"          # sub-ranges by the next bits of the mixed key
    try:
        res, _ = run(tab, dt, q, flags=HP)
    finally:
        del os.environ["VH_TEST_HPART_PASSES"]
    took_hpart(res)
    assert res.retries == 0


def test_pools_that_run_out_of_extents_are_resized(sets5):
    tab, dt = sets5
    q = {"dimensions": ["c", "x"], "metrics": ["users", "count"], "filter": F("lt", "x", "95")}
    for knob in ("VH_TEST_PART_EXTENTS", "VH_TEST_PART_EXTENTS2"):
        os.environ[knob] = "3"
        try:
            res, _ = run(tab, dt, q, flags=HP)
        finally:
            del os.environ[knob]
        assert res.path == "hash" and res.retries >= 1, knob


def _splitmix64(x):
    M = (1 << 64) - 1
    x = (x + 0x9E3779B97F4A7C15) & M
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M
    return x ^ (x >> 31)


def test_a_digits_second_extent_by_position(monkeypatch):
    """Groups picked so that their mixed keys (vh_splitmix64 of c | x << 16) start with one of THREE bytes: every scan block then appends
    ~5 000 tuples to each of three digits — more than one 4 096-tuple extent, so the (block, digit)'s second extent, a whole level further
    into the pool, is written and read (what the full-size tables never do: C5 puts ~950 tuples into a (block, digit)). Level B behind it
    spreads them over 256 ranges as usual. (Three partitions that hold everything are HEAVY partitions to the planner on the device — left to a
    second pass through the plain hash organisation —, which is not what this case is about: VH_NO_HEAVY_PASS; with the second pass the answer is
    checked once more at the end.)"""
    monkeypatch.setenv("VH_NO_HEAVY_PASS", "1")
    rng = np.random.default_rng(9)
    pairs = np.array([(c, x) for c in range(40) for x in range(5000) if _splitmix64(c | (x << 16)) >> 56 < 3], dtype=np.int64)
    assert 1500 < len(pairs) < 3500
    n = 200_000
    tab = vo.Table({"name": "t", "segment_size": n, "dimensions": [{"name": "c", "type": "ushort"}, {"name": "x", "type": "uint"}],
                    "metrics": [{"name": "count", "type": "count"}, {"name": "v", "type": "int_sum"}]})
    for _ in range(3):
        pick = pairs[rng.integers(0, len(pairs), n)]
        tab.add_segment_arrays([pick[:, 0].astype(np.uint16), pick[:, 1].astype(np.uint32)],
                               [np.ones(n, dtype=np.uint32), rng.integers(-1000, 1000, n).astype(np.int32)], None, n)
    dt = mirror_table(tab)
    try:
        res, st = run(tab, dt, {"dimensions": ["c", "x"], "metrics": ["count", "v"]}, flags=HP)
        took_hpart(res)
        assert res.retries == 0 and scan_wrote_level_a(res) and res.ngroups == st.ngroups == len(pairs), (res.retries, res.kernel, res.ngroups)
        monkeypatch.delenv("VH_NO_HEAVY_PASS")
        res, st = run(tab, dt, {"dimensions": ["c", "x"], "metrics": ["count", "v"]}, flags=HP)
        took_hpart(res)
        assert res.retries == 1 and res.ngroups == st.ngroups == len(pairs), (res.retries, res.kernel, res.ngroups)      # (every tuple through the second pass)
    finally:
        dt.close()


def test_a_hot_key_takes_its_extents_from_the_overflow_regions():
    """The scan that writes level A itself gives every (block, digit) stream its share of the tuples by POSITION; four rows of five in ONE
    group put far more than that into one digit of level A and into one range of level B: those streams go on in their pool's shared overflow
    region (one global atomic per extent) — no void attempt, no other writer. Same rows as the oracle's; and again with NO positional levels at
    all, where every tuple of both levels goes through the overflow regions."""
    rng = np.random.default_rng(8)
    n = 200_000
    tab = vo.Table({"name": "t", "segment_size": n, "dimensions": [{"name": "c", "type": "ushort"}, {"name": "x", "type": "uint"}],
                    "metrics": [{"name": "count", "type": "count"}, {"name": "v", "type": "int_sum"}]})
    for _ in range(3):
        hot = rng.random(n) < 0.8
        tab.add_segment_arrays([np.where(hot, 7, rng.integers(0, 40, n)).astype(np.uint16), np.where(hot, 3, rng.integers(0, 5000, n)).astype(np.uint32)],
                               [np.ones(n, dtype=np.uint32), rng.integers(-1000, 1000, n).astype(np.int32)], None, n)
    dt = mirror_table(tab)
    try:
        res, st = run(tab, dt, {"dimensions": ["c", "x"], "metrics": ["count", "v"]}, flags=HP)
        took_hpart(res)
        # (the hot group's range is HEAVY for the ranges' kernel: its rows take the second pass through the plain hash organisation — one more pass, not a void attempt)
        assert res.retries <= 1 and scan_wrote_level_a(res) and res.ngroups == st.ngroups > 50_000, (res.retries, res.kernel)
        for levels in ("0", "1"):
            os.environ["VH_TEST_POS_LEVELS"] = levels
            try:
                res, st = run(tab, dt, {"dimensions": ["c", "x"], "metrics": ["count", "v"]}, flags=HP)
            finally:
                del os.environ["VH_TEST_POS_LEVELS"]
            took_hpart(res)
            assert res.retries <= 1 and scan_wrote_level_a(res), (levels, res.retries, res.kernel)
    finally:
        dt.close()


def test_skew_one_group_gets_everything():
    """Every row in ONE group with ids from a large space: the range's set cannot hold them in any number of passes, the block's list of
    extents overflows — the query ends up on the plain hash table and still answers like the oracle."""
    rng = np.random.default_rng(5)
    n = 60_000
    tab = vo.Table({"name": "t", "segment_size": n, "dimensions": [{"name": "c", "type": "ushort"}, {"name": "x", "type": "uint"}],
                    "metrics": [{"name": "users", "type": "bitset", "max": 2 ** 31}, {"name": "count", "type": "count"}]})
    for _ in range(3):
        sets = [set(int(v) for v in rng.integers(0, 10 ** 6, 3)) for _ in range(n)]
        tab.add_segment_arrays([np.full(n, 7, dtype=np.uint16), np.full(n, 3, dtype=np.uint32)], [sets, np.ones(n, dtype=np.uint32)], None, n)
    dt = mirror_table(tab)
    try:
        res, st = run(tab, dt, {"dimensions": ["c", "x"], "metrics": ["users", "count"]}, flags=HP)
        assert res.path == "hash" and res.ngroups == 1 and int(res.states[1][0]) == 3 * n
        res, st = run(tab, dt, {"dimensions": ["c", "x"], "metrics": ["count"]}, flags=HP)      # without the set: one slot takes every tuple
        assert res.ngroups == 1
    finally:
        dt.close()


def test_pairs_for_the_exchange_come_out_of_the_tuple_pool(sets5):
    """Sharded queries and the cluster merge need a result's (group, id) pairs regrouped by owner (vh_result_partition_pairs). A result
    of the hashed partitioning kept no device-wide set: the pairs are read out of its last tuple pool — every (group, id) the rows held,
    duplicates included. As SETS per owner they must be exactly the oracle's, and every pair must sit with the owner of its group."""
    import ctypes as C
    tab, dt = sets5
    q = {"type": "aggregate", "table": "t", "dimensions": ["c", "x"], "metrics": ["users", "count"], "filter": F("lt", "x", "80")}
    from tests.planner import plan_from_query
    from tests.test_gpu_typed import NOW
    aq = vo.parse_query(tab, q)
    plan = plan_from_query(tab, aq, now=NOW, flags=HP)
    h = dt.query_agg_keep(plan)
    try:
        res = dt.collect(h, plan)
        took_hpart(res)
        nparts = 3
        goffs, gbufs = dt.partition(h, nparts)
        offs, bufs = dt.partition_pairs(h, 0, nparts)
        assert len(bufs) == 3 and int(offs[0]) == 0 and all(b[1] == int(offs[-1]) for b in bufs)
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

        def read(ptr, n, dtype):
            out = np.empty(int(n), dtype=dtype)
            if n:
                assert hip.hipMemcpy(out.ctypes.data, C.c_void_p(ptr), out.nbytes, 4) == 0
            return out
        n = int(offs[-1])
        pc, px, pid = read(bufs[0][0], n, np.uint16), read(bufs[1][0], n, np.uint32), read(bufs[2][0], n, np.uint32)
        gc, gx = read(gbufs[0][0], int(goffs[-1]), np.uint16), read(gbufs[1][0], int(goffs[-1]), np.uint32)
        owner_of = {}
        for part in range(nparts):
            for k in range(int(goffs[part]), int(goffs[part + 1])):
                owner_of[(int(gc[k]), int(gx[k]))] = part
        got = set()
        for part in range(nparts):
            for k in range(int(offs[part]), int(offs[part + 1])):
                key = (int(pc[k]), int(px[k]))
                assert owner_of[key] == part          # a pair travels to the owner of its group
                got.add((key, int(pid[k])))
        # the oracle's sets: every passing row's ids under its group key
        want = set()
        for seg in tab.segments:
            c, x = seg["d"][0][:seg["size"]], seg["d"][1][:seg["size"]]
            for r in np.nonzero(x < 80)[0]:
                for i in seg["m"][0][r]:
                    want.add(((int(c[r]), int(x[r])), int(i)))
        assert got == want and len(got) > 10_000
    finally:
        dt.discard(h)


def test_streamed_delivery_on_small_tables():
    """Big results of the hashed partitioning leave in chunks — the aggregation runs as eight launches over consecutive level-A partitions,
    each with its own region of the output columns, and finished chunks are copied out while the next ones aggregate. The full-size C5 test
    takes that path by itself; here VH_TEST_HP_STREAM asks for it on the small tables of this file (fresh process: one knob, read by the
    planner per query)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VH_TEST_HP_STREAM="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k",
                        "any_number_of_ids or payloads_and_key or ragged or widest_ids or skew or overflow"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]

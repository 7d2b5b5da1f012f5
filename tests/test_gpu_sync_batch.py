"""vh_table_sync_batch: what ONE upsert batch did to the table — rows appended to the last segments, metrics of
existing rows updated in place anywhere (src/codegen/db/upsert.cc:384-411) — reaches the mirror in one call and one kernel launch:
every element type, ranges that start and end anywhere (the pull kernel copies 16-byte pieces of the SOURCE's alignment), host memory
that is registered (read in place over PCIe), small unregistered runs (pinned ring) and big ones (DMA + stats in place). After it the
arenas hold the host's bytes, the per-segment stats cover them, derived layouts follow, and queries answer like the oracle."""
import ctypes as C
import mmap

import numpy as np
import pytest

from oracle import viya_oracle as vo
from tests.planner import col_descs, mirror_table, storage_index
from tests.test_gpu_typed import F, TYPES, _rand, run, typed_table
from viyadb_amd import capi
from viyadb_amd.executor import DeviceTable

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _init():
    from viyadb_amd import executor
    executor.init(0)


def seg_columns(tab, seg):
    """The segment's FULL column arrays in storage order (what the generated text takes addresses of)."""
    cols = list(seg["d"])
    for m in tab.metrics:
        cols.append(None if m.agg == "bitset" else seg["m"][m.index])
    if tab.has_hidden_count:
        cols.append(seg["count"])
    return cols


def restat(tab, seg):
    """SegmentStats of the rows the segment holds now (the reference widens them row by row at Insert: store.cc:171-201)."""
    n = seg["size"]
    for d in tab.dims:
        if d.dim_type in ("numeric", "time"):
            col = seg["d"][d.index][:n]
            ident_max = d.num_type.dtype.type(d.num_type.cpp_min_value)
            ident_min = d.num_type.dtype.type(d.num_type.cpp_max_value)
            seg["dmax"][d.index] = max(col.max(), ident_max) if n else ident_max
            seg["dmin"][d.index] = min(col.min(), ident_min) if n else ident_min


def check_mirror(tab, dt):
    for s, seg in enumerate(tab.segments):
        n = seg["size"]
        for ci, a in enumerate(seg_columns(tab, seg)):
            if a is None:
                continue
            dev = dt.read_column(s, ci, n)
            assert np.array_equal(dev.view(np.uint8), np.ascontiguousarray(a[:n]).view(np.uint8)), (s, ci)


def capacity_table(nseg, rows, seg_size, seed):
    """typed_table whose arrays have the segment's full CAPACITY (rows beyond size() are there to be appended into)."""
    tab = typed_table(nseg=nseg, rows=seg_size, seg_size=seg_size, seed=seed)
    for seg in tab.segments:
        seg["size"] = rows
        restat(tab, seg)
    return tab


def test_batch_initial_load_updates_and_appends():
    tab = capacity_table(nseg=4, rows=30_000, seg_size=50_000, seed=21)
    dt = DeviceTable(col_descs(tab), tab.segment_size, reserve_segments=1)      # grows inside the batch
    rng = np.random.default_rng(5)
    try:
        dt.sync_batch([(s, 0, seg["size"], seg["size"], seg_columns(tab, seg), 0) for s, seg in enumerate(tab.segments)])
        check_mirror(tab, dt)
        q = {"dimensions": ["s8", "flag"], "metrics": ["count", "long_sum", "double_sum", "int_min", "ushort_max", "float_avg"], "filter": F("lt", "d_uint", "20")}
        run(tab, dt, q)
        st0 = dt.sync_stats()
        assert st0["batches"] == 1 and st0["bytes_pulled"] == 0 and st0["bytes_staged"] + st0["bytes_dma"] > 0
        # one upsert batch: metrics of rows here and there in every segment (odd starts, odd lengths), rows appended to the last two
        items = []
        nd = len(tab.dims)
        for s, seg in enumerate(tab.segments):
            for first, n in ((1, 1), (13, 7), (4097, 333), (29_000 - s, 999 + s)):
                for m in tab.metrics:
                    a = seg["m"][m.index]
                    a[first:first + n] = _rand(rng, a.dtype, n)
                items.append((s, first, n, seg["size"], seg_columns(tab, seg), capi.SYNC_METRICS_ONLY))
        for s in (2, 3):
            seg = tab.segments[s]
            seg["size"] = 41_234 + s
            restat(tab, seg)
            items.append((s, 30_000, seg["size"] - 30_000, seg["size"], seg_columns(tab, seg), 0))
        dt.sync_batch(items)
        check_mirror(tab, dt)
        run(tab, dt, q)
        run(tab, dt, q, flags=capi.PLAN_FORCE_PACK)
        st1 = dt.sync_stats()
        assert st1["batches"] == 2 and st1["runs"] > st0["runs"]
        # a dimension's stats were widened by the appended rows: a value outside the old range is found, segments without it are skipped
        seg = tab.segments[3]
        seg["d"][5][seg["size"]] = 4_000_000                 # d_uint of one more appended row
        seg["size"] += 1
        restat(tab, seg)
        dt.sync_batch([(3, seg["size"] - 1, 1, seg["size"], seg_columns(tab, seg), 0)])
        res, st = run(tab, dt, {"dimensions": ["s8"], "metrics": ["count"], "filter": F("eq", "d_uint", "4000000")})
        assert res.ngroups == 1 and res.scanned_segments == 1
    finally:
        dt.close()


def test_batch_rejects_gaps_and_overflows():
    tab = capacity_table(nseg=1, rows=1000, seg_size=2000, seed=3)
    dt = DeviceTable(col_descs(tab), tab.segment_size, reserve_segments=1)
    try:
        cols = seg_columns(tab, tab.segments[0])
        dt.sync_batch([(0, 0, 1000, 1000, cols, 0)])
        with pytest.raises(capi.VhError, match="gap"):
            dt.sync_batch([(0, 1001, 10, 1011, cols, 0)])
        with pytest.raises(capi.VhError, match="do not fit"):
            dt.sync_batch([(0, 1000, 1001, 2001, cols, 0)])
        # two items on one segment chain: the second starts where the first ended
        tab.segments[0]["size"] = 1500
        restat(tab, tab.segments[0])
        dt.sync_batch([(0, 1000, 200, 1200, cols, 0), (0, 1200, 300, 1500, cols, 0)])
        check_mirror(tab, dt)
    finally:
        dt.close()


def _page_aligned(dtype, n):
    """A numpy array on its own anonymous pages (what a `new Segment` of tens of megabytes gets from the allocator)."""
    nbytes = max(1, n * np.dtype(dtype).itemsize)
    buf = mmap.mmap(-1, (nbytes + 4095) // 4096 * 4096)
    return np.frombuffer(buf, dtype=dtype, count=n), buf


def test_registered_host_memory_is_read_in_place():
    """vh_host_register: the segment's memory is pinned and mapped once; every later range of it is PULLED by the sync kernel — no host copy."""
    lib = capi.load()
    rng = np.random.default_rng(9)
    desc = {"name": "t", "segment_size": 100_000, "dimensions": [{"name": "a", "type": "uint"}, {"name": "b", "type": "ubyte"}, {"name": "c", "type": "short"}],
            "metrics": [{"name": "v", "type": "long_sum"}, {"name": "w", "type": "double_max"}, {"name": "count", "type": "count"}]}
    tab = vo.Table(desc)
    keep = []
    for s in range(3):
        arrs = []
        for dt_, gen in ((np.uint32, lambda n: rng.integers(0, 50, n)), (np.uint8, lambda n: rng.integers(0, 9, n)), (np.int16, lambda n: rng.integers(-300, 300, n)),
                         (np.int64, lambda n: rng.integers(-10 ** 6, 10 ** 6, n)), (np.float64, lambda n: rng.integers(-999, 999, n) / 4.0), (np.uint32, lambda n: rng.integers(1, 4, n))):
            a, buf = _page_aligned(dt_, 100_000)
            a[:] = gen(100_000).astype(dt_)
            keep.append(buf)
            arrs.append(a)
        tab.add_segment_arrays(arrs[:3], arrs[3:], None, 60_000)
    dt = DeviceTable(col_descs(tab), tab.segment_size, reserve_segments=3)
    try:
        for seg in tab.segments:
            for a in seg_columns(tab, seg):
                capi.check(lib.vh_host_register(a.ctypes.data, a.nbytes))
        dt.sync_batch([(s, 0, 60_000, 60_000, seg_columns(tab, seg), 0) for s, seg in enumerate(tab.segments)])
        st = dt.sync_stats()
        assert st["bytes_pulled"] == 3 * 60_000 * (4 + 1 + 2 + 8 + 8 + 4) and st["bytes_staged"] == 0 and st["bytes_dma"] == 0
        check_mirror(tab, dt)
        q = {"dimensions": ["b", "c"], "metrics": ["v", "w", "count"], "filter": F("lt", "a", "25")}
        run(tab, dt, q)
        # in-place updates at odd offsets + an append, all pulled
        items = []
        for s, seg in enumerate(tab.segments):
            seg["m"][0][777:777 + 41] += 5
            seg["m"][1][3:4] = 12345.25
            items += [(s, 777, 41, 60_000, seg_columns(tab, seg), capi.SYNC_METRICS_ONLY), (s, 3, 1, 60_000, seg_columns(tab, seg), capi.SYNC_METRICS_ONLY)]
        tab.segments[2]["size"] = 99_999
        restat(tab, tab.segments[2])
        items.append((2, 60_000, 39_999, 99_999, seg_columns(tab, tab.segments[2]), 0))
        dt.sync_batch(items)
        st2 = dt.sync_stats()
        assert st2["bytes_staged"] == 0 and st2["bytes_dma"] == 0 and st2["bytes_pulled"] > st["bytes_pulled"]
        check_mirror(tab, dt)
        run(tab, dt, q)
    finally:
        dt.close()
        for seg in tab.segments:
            for a in seg_columns(tab, seg):
                capi.check(lib.vh_host_unregister(a.ctypes.data))


def test_derived_layouts_follow_a_batch():
    """Narrow predicate copies, payload projections (bit-field records included) and the compiled kernels' views of them are refreshed
    for the segments a batch touched — and dropped when a shipped value no longer fits its stored width."""
    from viyadb_amd import synth
    from tests.parity import build_oracle_table, compare
    from viyadb_amd.executor import AggPlan
    w = synth.c3(segment_rows=100_000)
    nseg, rows = 4, 100_000
    dt = synth.create_device_table(w, nseg, rows)
    try:
        ot = build_oracle_table(w, nseg, rows)
        plan = lambda: AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, flags=capi.PLAN_FORCE_JIT | capi.PLAN_FORCE_PACK)
        dt.narrow([2, 3, 4])
        res = dt.query_agg(plan())
        compare(res, vo.scan_aggregate(vo.parse_query(ot, w.query)), "before")
        assert res.packed and res.narrow
        # metrics of some rows change in place in every segment; values stay inside their recorded widths
        items = []
        for s, seg in enumerate(ot.segments):
            for j in (0, 2):                       # m0 (long_sum), count
                seg["m"][j][5000 + s:5100 + s] = 1 + (np.arange(100) % 3)
            items.append((s, 5000 + s, 100, rows, list(seg["d"]) + list(seg["m"]), capi.SYNC_METRICS_ONLY))
        dt.sync_batch(items)
        res = dt.query_agg(plan())
        compare(res, vo.scan_aggregate(vo.parse_query(ot, w.query)), "after in-place updates")
        assert res.packed and res.narrow
        # a value that needs more bits than the records give it: the projection is rebuilt wider, the answer stays right
        seg = ot.segments[1]
        seg["m"][0][7] = 1 << 40
        dt.sync_batch([(1, 7, 1, rows, list(seg["d"]) + list(seg["m"]), capi.SYNC_METRICS_ONLY)])
        res = dt.query_agg(plan())
        compare(res, vo.scan_aggregate(vo.parse_query(ot, w.query)), "after an outgrown width")
    finally:
        dt.close()


def test_a_batch_that_fails_half_way_leaves_the_table_as_it_found_it(monkeypatch):
    """a failure after some of a batch's runs were launched (injected at item 2 of 4) must not leave seg_rows advanced, stats
    reset or too narrow, or pending slots for the next batch to mis-pair. The same batch sent again goes through; the answers are the oracle's,
    segment skipping included."""
    tab = capacity_table(nseg=4, rows=20_000, seg_size=40_000, seed=77)
    dt = DeviceTable(col_descs(tab), tab.segment_size, reserve_segments=4)
    rng = np.random.default_rng(11)
    q = {"dimensions": ["s8", "flag"], "metrics": ["count", "long_sum", "int_min", "ushort_max"], "filter": F("lt", "d_uint", "20")}
    try:
        dt.sync_batch([(s, 0, seg["size"], seg["size"], seg_columns(tab, seg), 0) for s, seg in enumerate(tab.segments)])
        run(tab, dt, q)
        items = []
        for s, seg in enumerate(tab.segments):      # every segment: an in-place update of metrics and an append that widens d_uint's range
            for m in tab.metrics:
                a = seg["m"][m.index]
                a[100:400] = _rand(rng, a.dtype, 300)
            seg["d"][5][20_000:25_000] = 5_000_000 + s
            seg["size"] = 25_000
            restat(tab, seg)
            items.append((s, 100, 300, 20_000, seg_columns(tab, seg), capi.SYNC_METRICS_ONLY))
            items.append((s, 20_000, 5_000, 25_000, seg_columns(tab, seg), 0))
        monkeypatch.setenv("VH_TEST_SYNC_FAIL_AT", "5")
        with pytest.raises(capi.VhError, match="injected"):
            dt.sync_batch(items)
        monkeypatch.delenv("VH_TEST_SYNC_FAIL_AT")
        from tests.planner import plan_from_query
        aq = vo.parse_query(tab, {"type": "aggregate", "table": "t", "dimensions": ["s8"], "metrics": ["count"]})
        assert dt.query_agg(plan_from_query(tab, aq, now=1496570140)).scanned_recs == 4 * 20_000       # rows mirrored: as before the batch
        # the table still answers for the OLD rows (a snapshot of 20 000 per segment) — with the updated metrics or the old ones where the
        # pull did not happen; what must hold is that nothing crashes and a later batch is not mis-paired: send it again, whole
        dt.sync_batch(items)
        check_mirror(tab, dt)
        run(tab, dt, q)
        for s in range(4):                           # the appended values are found, and only in their own segment
            res, st = run(tab, dt, {"dimensions": ["s8"], "metrics": ["count"], "filter": F("eq", "d_uint", str(5_000_000 + s))})
            assert res.passed_recs == 5_000
    finally:
        dt.close()

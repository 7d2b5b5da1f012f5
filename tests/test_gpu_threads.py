"""SURVEY 8(b) threading row: the reference enters the aggregate function from several read_pool threads while one writer
appends (src/db/database.cc:28-34). The library plans and launches under a per-table lock and runs every query on its own
execution context (stream + scratch + staging), so queries of one table overlap on the device. Threads hammer their own tables and one shared table (ctypes drops the GIL inside the calls) while a writer
keeps syncing new segments into the shared one; every answer must equal the single-threaded answer for the snapshot used."""
import threading

import numpy as np
import os

import pytest

pytestmark = pytest.mark.gpu


def test_concurrent_queries_and_a_writer():
    from viyadb_amd import capi, executor
    from viyadb_amd.executor import AggPlan, DeviceTable, GroupSpec
    executor.init(0)
    rows, nseg = 50_000, 6
    rng = np.random.default_rng(11)

    def make_table(seed):
        r = np.random.default_rng(seed)
        t = DeviceTable([(capi.DIM_NUMERIC, capi.U32), (capi.DIM_NUMERIC, capi.U32), (capi.METRIC_SUM, capi.I64), (capi.METRIC_COUNT, capi.U32)],
                        rows, reserve_segments=nseg)
        data = []
        for s in range(nseg):
            cols = [r.integers(0, 300, rows).astype(np.uint32), r.integers(0, 1000, rows).astype(np.uint32),
                    r.integers(-1000, 1000, rows).astype(np.int64), np.ones(rows, dtype=np.uint32)]
            t.sync_segment(s, cols, rows)
            data.append(cols)
        return t, data

    def expected(data, nsegs, thresh):
        k = np.concatenate([d[0] for d in data[:nsegs]])
        f = np.concatenate([d[1] for d in data[:nsegs]])
        v = np.concatenate([d[2] for d in data[:nsegs]])
        m = f < thresh
        sums = np.zeros(300, dtype=np.int64)
        cnts = np.zeros(300, dtype=np.int64)
        np.add.at(sums, k[m], v[m])
        np.add.at(cnts, k[m], 1)
        return sums, cnts

    def check(t, data, nsegs, thresh, flags=0):
        res = t.query_agg(AggPlan(filter=[("rel", 1, capi.OP_LT, thresh)], groups=[GroupSpec(0)], metrics=[2, 3], seg_rows=[rows] * nsegs,
                                  flags=flags))
        sums, cnts = expected(data, nsegs, thresh)
        got_s = np.zeros(300, dtype=np.int64)
        got_c = np.zeros(300, dtype=np.int64)
        got_s[res.keys[0]] = res.states[0]
        got_c[res.keys[0]] = res.states[1]
        assert np.array_equal(got_s, sums) and np.array_equal(got_c, cnts), (nsegs, thresh, flags)

    own = [make_table(100 + i) for i in range(3)]
    shared, shared_data = make_table(7)
    extra = [[rng.integers(0, 300, rows).astype(np.uint32), rng.integers(0, 1000, rows).astype(np.uint32),
              rng.integers(-1000, 1000, rows).astype(np.int64), np.ones(rows, dtype=np.uint32)] for _ in range(6)]
    errors = []
    synced = [nseg]            # segments of the shared table that are completely synced (only the writer appends)

    def reader_own(i):
        try:
            t, data = own[i]
            for it in range(60):
                check(t, data, 1 + (it % nseg), 100 + 37 * (it % 20), flags=[0, 1, 2, 8][it % 4])
        except Exception as e:   # noqa: BLE001
            errors.append(("own", i, repr(e)))

    def reader_shared(i):
        try:
            for it in range(60):
                n = synced[0]     # size() snapshot: whatever the writer finished before this query
                check(shared, shared_data, n, 150 + 41 * ((it + i) % 20), flags=[0, 1][it % 2])
        except Exception as e:   # noqa: BLE001
            errors.append(("shared", i, repr(e)))

    def writer():
        try:
            for s, cols in enumerate(extra):
                shared_data.append(cols)
                shared.sync_segment(nseg + s, cols, rows)
                synced[0] = nseg + s + 1
        except Exception as e:   # noqa: BLE001
            errors.append(("writer", 0, repr(e)))

    threads = [threading.Thread(target=reader_own, args=(i,)) for i in range(3)]
    threads += [threading.Thread(target=reader_shared, args=(i,)) for i in range(3)] + [threading.Thread(target=writer)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    try:
        assert not errors, errors[:3]
        check(shared, shared_data, nseg + len(extra), 500)
    finally:
        for t, _ in own:
            t.close()
        shared.close()


@pytest.mark.skipif(os.environ.get("VH_JIT", "") in ("0", "off"), reason="VH_JIT=off: the ratio below was characterised with the per-query compiled kernels")
def test_queries_of_one_table_overlap():
    """Two threads on ONE table finish 2 x N queries no later than one thread finishes 2N (sooner, as a rule): planning is serialised per table,
    but a launched query waits for the device and reads its groups back outside the lock, on its own context."""
    import time
    from viyadb_amd import capi, executor, synth
    from viyadb_amd.executor import AggPlan
    executor.init(0)
    w = synth.c2()
    t = synth.create_device_table(w, 2)                      # 2 M rows: ~15 us of kernel under ~65 us of launches, emission, read-back and host work per query
                                                             # (two threads: 0.64-0.67 of the serial time on 1-2 M rows; with 16 M rows the kernels fill more of the query and the
                                                             # ratio wanders between 0.74 and 0.96 — round 4 took ~100 us of waiting out of every query)
    try:
        plans = [t.prepare(AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics)) for _ in range(2)]
        want = t.query_agg(plans[0])
        n = 400

        def loop(plan, count, out):
            for _ in range(count):
                r = t.query_agg(plan)
                if r.ngroups != want.ngroups or int(r.states[0].sum()) != int(want.states[0].sum()):
                    out.append("mismatch")

        for _ in range(50):
            t.query_agg(plans[0])
        best_serial, best_par = 1e9, 1e9
        for _ in range(3):
            errs = []
            t0 = time.perf_counter()
            loop(plans[0], 2 * n, errs)
            best_serial = min(best_serial, time.perf_counter() - t0)
            ths = [threading.Thread(target=loop, args=(plans[i], n, errs)) for i in range(2)]
            t0 = time.perf_counter()
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            best_par = min(best_par, time.perf_counter() - t0)
            assert not errs
        print("serial %.1f ms, two threads %.1f ms" % (best_serial * 1e3, best_par * 1e3))
        # (a ratio, not a correctness property: 0.64-0.67 when a query cost ~65 us outside its kernel, 0.85-0.95 now that it costs ~40 —
        # the bound only says that a second thread does not make the pair slower than one thread alone)
        assert best_par < 1.05 * best_serial, (best_serial, best_par)
    finally:
        t.close()


def test_repeated_partitioned_query_visits_contexts():
    """A partitioned plan with a big tuple pool is timed on three execution contexts before the pool settles on the one whose
    scratch landed best (placement preference): every run must return the same groups whichever context served it, also while a
    held result pins one of the contexts."""
    from viyadb_amd import capi, executor, synth
    from viyadb_amd.executor import AggPlan
    executor.init(0)
    w = synth.c3()
    t = synth.create_device_table(w, 100)                    # 100 M rows, ~5 M survivors: a ~100 MB tuple pool
    try:
        plan = AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, flags=capi.PLAN_FORCE_PART, groups_hint=100000)
        first = t.query_agg(plan)
        assert first.path == "dense_part"
        o = np.lexsort([first.keys[1], first.keys[0]])
        want = [first.keys[0][o], first.keys[1][o], first.states[0][o], first.states[1][o]]
        held = None
        for i in range(8):
            r = t.query_agg(plan)
            o = np.lexsort([r.keys[1], r.keys[0]])
            for a, b in zip(want, [r.keys[0][o], r.keys[1][o], r.states[0][o], r.states[1][o]]):
                assert np.array_equal(a, b), i
            if i == 2:
                held = t.query_agg_keep(plan)            # pins whichever context it got
        t.discard(held)
    finally:
        t.close()

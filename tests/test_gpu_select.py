"""vh_query_select through the C-ABI on synthetic tables: ordered emission, skip/limit windows,
size() snapshots, segment sizes that are not multiples of the 1024-row wave step, every element type as an output
column, bitset cardinalities, and a full-size run checked through size-independent properties."""
import numpy as np
import pytest

from oracle import viya_oracle as vo
from tests.parity import build_oracle_table
from viyadb_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _init():
    from viyadb_amd import executor
    executor.init(0)


def oracle_select(ot, query, cols, skip, limit, seg_rows=None):
    aq = vo.parse_query(ot, dict(query, dimensions=[], metrics=[]))
    aq.skip, aq.limit = skip, limit
    picked, stats = vo.scan_select(aq, seg_rows)
    out = []
    nd = len(ot.dims)
    for c in cols:
        vals = []
        for si, i in picked:
            seg = ot.segments[si]
            if c < nd:
                vals.append(seg["d"][c][i])
            elif ot.metrics[c - nd].agg == "bitset":
                vals.append(len(seg["m"][c - nd][i]))
            else:
                vals.append(seg["m"][c - nd][i])
        out.append(np.array(vals))
    return out, stats


@pytest.mark.parametrize("wl,rps,nseg", [("C2", 50000, 5), ("C3", 33333, 4), ("C1", 1000, 3), ("C3", 1023, 7), ("C5", 5000, 3)])
def test_select_matches_oracle(wl, rps, nseg):
    w = synth.WORKLOADS[wl](segment_rows=rps)
    dt = synth.create_device_table(w, nseg, rps, 0, 42)
    try:
        ot = build_oracle_table(w, nseg, rps, 0, 42)
        cols = list(range(len(w.columns)))
        snap = [rps] * nseg
        snap[1] = rps // 3 + 1            # size() snapshot smaller than what is mirrored
        for skip, limit, seg_rows in ((0, 0, None), (0, 10, None), (17, 0, None), (5, 3, None), (10 ** 9, 0, None), (3, 2000, snap)):
            got, info = dt.query_select(w.plan.filter, cols, skip=skip, limit=limit, seg_rows=seg_rows)
            want, stats = oracle_select(ot, w.query, cols, skip, limit, seg_rows)
            assert info.nrows == stats["output_recs"], (skip, limit)
            assert info.scanned_recs == stats["scanned_recs"] and info.scanned_segments == stats["scanned_segments"]
            assert info.passed_recs == stats["passed_recs"]
            for c, (a, b) in enumerate(zip(got, want)):
                assert len(a) == len(b)
                if len(b):
                    assert np.array_equal(a, b.astype(a.dtype)), (w.columns[c].name, skip, limit)
    finally:
        dt.close()


def test_select_full_size_properties():
    """C3 at 100 M rows: the emitted ids are strictly increasing (storage order), every row satisfies the predicate,
    the window arithmetic holds, and limit-less emission returns exactly passed_recs rows."""
    w = synth.WORKLOADS["C3"](segment_rows=1_000_000)
    nseg = 100
    dt = synth.create_device_table(w, nseg, w.segment_rows, 0, 42)
    try:
        names = [c.name for c in w.columns]
        cols = [names.index(n) for n in ("d2", "d3", "d4", "id", "m0")]
        got, info = dt.query_select(w.plan.filter, cols, skip=1000, limit=0)
        d2, d3, d4, ids, m0 = got
        assert info.nrows == info.passed_recs - 1000 and len(ids) == info.nrows
        assert 0.04 * nseg * 1e6 < info.passed_recs < 0.06 * nseg * 1e6
        assert np.all(d2 == 1) and np.all(d3 < 447) and np.all(d4 >= 553)
        assert np.all(np.diff(ids.astype(np.int64)) > 0)
        # limit reached inside the first segment: one extra row from each of the 99 later segments (reference rule)
        got2, info2 = dt.query_select(w.plan.filter, cols, skip=1000, limit=50)
        assert info2.nrows == 50 + (nseg - 1)
        assert np.array_equal(got2[3][:50], ids[:50])
        later = got2[3][50:]
        assert np.array_equal(later // 1_000_000, np.arange(1, nseg))
        assert info2.kernel_ms < 50
    finally:
        dt.close()


def test_select_with_a_bitset_cardinality_predicate():
    """C5 rows whose count-distinct set holds exactly two ids (filter.cc:216: `tuple_metrics._j[idx].cardinality() == farg`):
    the synthetic generator draws two ids per row, equal now and then."""
    w = synth.WORKLOADS["C5"](segment_rows=4000)
    dt = synth.create_device_table(w, 3, 4000, 0, 42)
    try:
        ot = build_oracle_table(w, 3, 4000, 0, 42)
        users = w.col("users")
        from viyadb_amd.executor import anynum
        for filt, q in (([("rel", users, capi.OP_EQ, anynum(capi.U32, 2))], {"op": "eq", "column": "users", "value": "2"}),
                        ([("rel", users, capi.OP_GE, anynum(capi.U32, 2)), ("rel", w.col("u"), capi.OP_LT, 500000), ("and", 2)],
                         {"op": "and", "filters": [{"op": "ge", "column": "users", "value": "2"}, {"op": "lt", "column": "u", "value": "500000"}]})):
            cols = [w.col("u"), users]
            got, info = dt.query_select(filt, cols, skip=3, limit=500)
            want, stats = oracle_select(ot, dict(w.query, filter=q), cols, 3, 500)
            assert info.nrows == stats["output_recs"] and info.passed_recs == stats["passed_recs"] and info.nrows > 0
            for a, b in zip(got, want):
                assert np.array_equal(a, b.astype(a.dtype))
    finally:
        dt.close()

"""Pins the oracle: every known-answer test the reference's suite holds for the aggregate path
(tests/golden/reference_cases.json, transcribed from /root/reference/test/*.cc) must pass on the
CPU restatement before it is trusted as the checker for the HIP path."""
import pytest

from oracle import viya_oracle as vo
from tests import golden_cases as gc


def run_oracle(tconf, loads, query, now):
    db = vo.Database({"tables": [tconf]})
    t = db.table(tconf["name"])
    for batch in loads:
        t.load(batch, now=now)
    try:
        rows, stats = db.query(query, now=now)
    except vo.InvalidArgument as e:
        e.reference_exception = "invalid_argument"
        raise
    info = {"segments": len(t.segments), "segment_sizes": [s["size"] for s in t.segments]}
    return rows, stats, info


@pytest.mark.parametrize("cid", gc.CASE_IDS)
def test_reference_case(cid):
    gc.check_case(gc.case_by_id(cid), run_oracle)


def test_twin_matches_numpy_oracle_on_reference_cases():
    """The emitted C++ twin (cpu_baseline) and the numpy interpreter agree group for group."""
    import numpy as np
    from oracle import cpu_twin
    from tests.parity import sort_rows
    n = 0
    for case in gc.CASES:
        if "throws" in case or case["query"].get("type") != "aggregate":
            continue
        seen_bitset = locals().get("seen_bitset", False) or case["table"] == "UserEvents"
        tconf = gc.table_conf(case)
        db = vo.Database({"tables": [tconf]})
        t = db.table(tconf["name"])
        for batch in gc.materialise_loads(case):
            t.load(batch, now=case.get("now"))
        q = gc.materialise_query(case)
        st = vo.scan_aggregate(vo.parse_query(t, q), now=case.get("now"))
        st2 = cpu_twin.Twin(t, q).run(now=case.get("now"))
        p1, p2 = sort_rows(st.keys, st.states), sort_rows(st2.keys, st2.states)
        assert st.ngroups == st2.ngroups, case["id"]
        for a, b in zip(st.keys + st.states, st2.keys + st2.states):
            if a.dtype.kind == "f":
                np.testing.assert_allclose(a[p1], b[p2], rtol=1e-12, err_msg=case["id"])
            else:
                assert np.array_equal(a[p1], b[p2]), case["id"]
        n += 1
    assert n > 40 and seen_bitset      # bitset.BitsetMetric (test/bitset.cc:32-53) goes through the twin's Roaring stand-in


def test_twin_matches_numpy_oracle_on_a_c5_window():
    """C5's query (time rollup + hour granularity + COUNT DISTINCT + COUNT) on a seeded window of the synthetic rows: the twin reading the
    bitset column as CSR arrays (what bench.py's cpu_baseline times) against the numpy oracle reading a Python set per row — and the
    hot-key twin of it, whose one group holds a fifth of the survivors."""
    import numpy as np
    from oracle import cpu_twin
    from tests.parity import build_oracle_table, sort_rows
    from viyadb_amd import synth
    for name in ("C5", "C5h", "C5t"):
        w = synth.WORKLOADS[name](segment_rows=30_000)
        st = vo.scan_aggregate(vo.parse_query(build_oracle_table(w, 2, 30_000, row_base=77_000), w.query), now=w.now)
        tw = cpu_twin.Twin(build_oracle_table(w, 2, 30_000, row_base=77_000, csr=True), w.query)
        st2 = tw.run(now=w.now)
        assert st.ngroups == st2.ngroups > 10_000, name
        p1, p2 = sort_rows(st.keys, st.states), sort_rows(st2.keys, st2.states)
        for a, b in zip(st.keys + st.states, st2.keys + st2.states):
            assert np.array_equal(a[p1], b[p2]), name
        # and once more from the oracle's own per-row sets (the conversion the reference-case test relies on)
        st3 = cpu_twin.Twin(build_oracle_table(w, 2, 30_000, row_base=77_000), w.query).run(now=w.now)
        p3 = sort_rows(st3.keys, st3.states)
        for a, b in zip(st.keys + st.states, st3.keys + st3.states):
            assert np.array_equal(a[p1], b[p3]), name

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
        if "throws" in case or case["table"] in ("UserEvents",) or case["query"].get("type") != "aggregate":
            continue
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
    assert n > 40

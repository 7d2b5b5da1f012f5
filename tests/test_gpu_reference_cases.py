"""The reference's own known-answer tests (tests/golden/reference_cases.json), run through the
product: C++ host shim (JSON -> plan) -> C-ABI -> HIP kernels -> host post-aggregation.
They read like the reference's tests: create the database from JSON, SimpleLoader-style loads,
Query into a row list, compare (sorted unless the reference compares in order)."""
import pytest

from tests import golden_cases as gc

pytestmark = pytest.mark.gpu


def run_gpu(tconf, loads, query, now):
    from viyadb_amd import hostdb
    db = hostdb.Database({"tables": [tconf]})
    try:
        for batch in loads:
            db.load(tconf["name"], batch, now=now)
        rows, stats = db.query(query, now=now)
        ti = db.table_info(tconf["name"])
        return rows, stats, {"segments": ti["segments"], "segment_sizes": [ti["first_segment_size"]]}
    finally:
        db.close()


@pytest.mark.parametrize("cid", gc.CASE_IDS)
def test_reference_case_on_gpu(cid):
    gc.check_case(gc.case_by_id(cid), run_gpu)


@pytest.mark.parametrize("flags", ["1", "2", "4"])
@pytest.mark.parametrize("cid", ["aggregation.BasicQuery", "aggregation.NumericDimensions", "metrics.long.AggregateMetrics",
                                 "metrics.float.AggregateMetrics", "time.TimeEvents.QueryGranularity",
                                 "time.DynamicRollup.TimestampMicroIngestion", "boolean.QueryTest"])
def test_reference_case_forced_table_organisation(cid, flags, monkeypatch):
    """Same answers from the hash table (1), the HBM dense table (2) and without XCD-private copies (4)."""
    monkeypatch.setenv("VIYA_HIP_PLAN_FLAGS", flags)
    gc.check_case(gc.case_by_id(cid), run_gpu)

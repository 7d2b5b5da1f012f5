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


def test_mirror_follows_upserts():
    """queries interleaved with loads. Upsert appends rows AND updates metrics of existing rows in
    place; the HBM mirror must track both (dirty-range sync), across a segment boundary."""
    import random
    from oracle import viya_oracle as vo
    from viyadb_amd import hostdb
    tconf = {"name": "events", "segment_size": 500,
             "dimensions": [{"name": "country"}, {"name": "event_name"}, {"name": "day", "type": "uint"}],
             "metrics": [{"name": "count", "type": "count"}, {"name": "revenue", "type": "double_sum"},
                         {"name": "best", "type": "int_max"}, {"name": "users", "type": "bitset"}]}
    q = {"type": "aggregate", "table": "events", "dimensions": ["country", "event_name"],
         "metrics": ["count", "revenue", "best", "users"], "filter": {"op": "ge", "column": "day", "value": "3"}}
    rnd = random.Random(4)
    gdb = hostdb.Database({"tables": [tconf]})
    odb = vo.Database({"tables": [tconf]})
    try:
        for batch in range(6):
            rows = [[rnd.choice(["US", "IL", "KZ", "RU", "AZ"]), rnd.choice(["open", "buy", "quit"]), str(rnd.randrange(0, 90)),
                     str(rnd.randrange(0, 500) / 4), str(rnd.randrange(-50, 50)), str(rnd.randrange(0, 40))] for _ in range(350)]
            gdb.load("events", rows)
            odb.table("events").load(rows)
            got, gst = gdb.query(q)
            want, ost = odb.query(q)
            assert sorted(got) == sorted(want), batch
            assert gst["scanned_recs"] == ost["scanned_recs"] and gst["aggregated_recs"] == ost["aggregated_recs"]
        assert gdb.table_info("events")["segments"] >= 2
    finally:
        gdb.close()

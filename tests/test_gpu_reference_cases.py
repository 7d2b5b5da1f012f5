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


def _events_pair(segment_size=300, nrows=2000, seed=9):
    """The same random events loaded into the product (host shim + GPU) and the oracle; several segments."""
    import random
    from oracle import viya_oracle as vo
    from viyadb_amd import hostdb
    tconf = {"name": "events", "segment_size": segment_size,
             "dimensions": [{"name": "country"}, {"name": "event_name"}, {"name": "day", "type": "uint"},
                            {"name": "ok", "type": "boolean"}, {"name": "ts", "type": "time", "format": "%Y-%m-%d %H:%M:%S"},
                            {"name": "id", "type": "uint"}],
             "metrics": [{"name": "count", "type": "count"}, {"name": "revenue", "type": "double_sum"},
                         {"name": "best", "type": "int_max"}, {"name": "avg_len", "type": "long_avg"}, {"name": "users", "type": "bitset"}]}
    rnd = random.Random(seed)
    names = ["open", "buy", "quit", "refund", "review", "rate", "share", "purchase", "donate", "browse"]
    rows = [[rnd.choice(["US", "IL", "KZ", "RU", "AZ", "CH"]), rnd.choice(names[: 3 + (i * 7 // nrows)]), str(rnd.randrange(0, 90)),
             rnd.choice(["true", "false"]), "2017-06-%02d 10:%02d:00" % (1 + i % 28, i % 60), str(i),
             str(rnd.randrange(0, 500) / 4), str(rnd.randrange(-50, 50)), str(rnd.randrange(1, 30)), str(rnd.randrange(0, 40))]
            for i in range(nrows)]
    gdb = hostdb.Database({"tables": [tconf]})
    odb = vo.Database({"tables": [tconf]})
    gdb.load("events", rows)
    odb.table("events").load(rows)
    return gdb, odb


def test_select_matches_oracle_across_segments():
    """select = the passing rows in storage order through skip/limit, including the reference's rule
    that `break` leaves the tuple loop only (every later segment still sends one row once the limit is reached)."""
    gdb, odb = _events_pair()
    try:
        assert gdb.table_info("events")["segments"] >= 6
        base = {"type": "select", "table": "events", "dimensions": ["country", "event_name", "day", "ok", "ts", "id"],
                "metrics": ["count", "revenue", "best", "avg_len", "users"]}
        flt = {"op": "and", "filters": [{"op": "lt", "column": "day", "value": "30"}, {"op": "ne", "column": "country", "value": "US"}]}
        for extra in ({}, {"limit": 7}, {"skip": 5}, {"skip": 450, "limit": 20}, {"limit": 1}, {"skip": 100000}, {"header": True, "limit": 3},
                      {"filter": {"op": "eq", "column": "country", "value": "nowhere"}},
                      {"filter": {"op": "gt", "column": "id", "value": "1500"}, "limit": 2}):   # segment skipping by id range
            q = dict(base, filter=flt)
            q.update(extra)
            got, gst = gdb.query(q)
            want, ost = odb.query(q)
            assert got == want, extra
            for k in ("scanned_recs", "scanned_segments", "output_recs"):
                assert gst[k] == ost[k], (extra, k)
        # the `select` column form with a time format override and `*`
        q = {"type": "select", "table": "events", "select": [{"column": "ts", "format": "%d/%m"}, {"column": "*"}], "limit": 4, "skip": 298}
        assert gdb.query(q)[0] == odb.query(q)[0]
    finally:
        gdb.close()


def test_search_matches_oracle_across_segments():
    """search = distinct values of a dimension among passing rows in first-occurrence order, term match,
    limit (with the same tuple-loop-only `break`), on string / numeric / boolean dimensions."""
    gdb, odb = _events_pair()
    try:
        flt = {"op": "ge", "column": "day", "value": "10"}
        for dim, term, limit in (("event_name", "", 0), ("event_name", "r", 0), ("event_name", "", 2), ("event_name", "e", 1),
                                 ("event_name", "zzz", 3), ("country", "", 4), ("day", "1", 0), ("day", "", 5), ("ok", "", 0),
                                 ("ok", "tr", 1), ("id", "99", 3)):
            q = {"type": "search", "table": "events", "dimension": dim, "term": term, "filter": flt}
            if limit:
                q["limit"] = limit
            got, gst = gdb.query(q)
            want, ost = odb.query(q)
            assert got == want, (dim, term, limit)        # push order is storage order of first occurrence: exact
            for k in ("scanned_recs", "scanned_segments", "aggregated_recs", "output_recs"):
                assert gst[k] == ost[k], (dim, term, limit, k)
        q = {"type": "search", "table": "events", "dimension": "event_name", "term": "o", "header": True}
        assert gdb.query(q)[0] == odb.query(q)[0]
    finally:
        gdb.close()


def test_sorted_limit_with_device_top_n(monkeypatch):
    """`sort` + `limit` over ~400 K group slots. The device keeps only the groups that can make the
    window; rows must equal the oracle's exactly (the second sort column makes the order total), with and without it."""
    import random
    from oracle import viya_oracle as vo
    from viyadb_amd import hostdb
    tconf = {"name": "events", "segment_size": 40000,
             "dimensions": [{"name": "country"}, {"name": "id", "type": "uint"}],
             "metrics": [{"name": "count", "type": "count"}, {"name": "revenue", "type": "double_sum"}, {"name": "best", "type": "int_max"},
                         {"name": "spread", "type": "long_avg"}]}
    rnd = random.Random(21)
    rows = [[rnd.choice(["US", "IL", "KZ", "RU", "AZ", "CH"]), str(i % 70000), str(rnd.randrange(0, 100000) / 8), str(rnd.randrange(-3000, 3000)),
             str(rnd.randrange(1, 50))] for i in range(90000)]
    gdb = hostdb.Database({"tables": [tconf]})
    odb = vo.Database({"tables": [tconf]})
    try:
        gdb.load("events", rows)
        odb.table("events").load(rows)
        base = {"type": "aggregate", "table": "events", "dimensions": ["id", "country"], "metrics": ["count", "revenue", "best", "spread"]}
        for sort, extra in (([{"column": "best"}, {"column": "id", "ascending": True}, {"column": "country"}], {"limit": 20}),
                            ([{"column": "best", "ascending": True}, {"column": "id"}, {"column": "country"}], {"limit": 7, "skip": 5}),
                            ([{"column": "revenue"}, {"column": "id"}, {"column": "country"}], {"limit": 15}),
                            ([{"column": "id"}, {"column": "country"}], {"limit": 9, "skip": 3}),
                            ([{"column": "count"}, {"column": "id", "ascending": True}, {"column": "country"}], {"limit": 11, "having": {"op": "lt", "column": "best", "value": "0"}}),
                            ([{"column": "spread"}, {"column": "id"}, {"column": "country"}], {"limit": 5}),      # AVG: host-side order
                            ([{"column": "country"}, {"column": "id"}], {"limit": 5})):                             # string: host-side order
            q = dict(base, sort=sort, **extra)
            want, ost = odb.query(q)
            for host_only in ("", "1"):
                if host_only:
                    monkeypatch.setenv("VIYA_HOST_TOPN", "1")
                else:
                    monkeypatch.delenv("VIYA_HOST_TOPN", raising=False)
                got, gst = gdb.query(q)
                assert got == want, (sort, extra, host_only)
                assert gst["aggregated_recs"] == ost["aggregated_recs"] and gst["output_recs"] == ost["output_recs"]
    finally:
        gdb.close()

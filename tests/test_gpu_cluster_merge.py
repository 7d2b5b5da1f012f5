"""the cluster aggregate with binary partial states (viyadb_amd/host/partial_state.h).

The reference's controller (src/cluster/query/agg_runner.cc:83-140) strips header/having/sort/skip/limit, collects the
workers' answers and re-aggregates them in a temporary table; its TSV hop loses AVG (divided too early) and bitsets
(cardinalities do not merge). With typed partial states the merged answer must EQUAL what one database holding all the
rows answers — that is the oracle here: the CPU restatement loaded with the union of the workers' rows."""
import random

import pytest

from tests import partial_wire as pw

pytestmark = pytest.mark.gpu

TCONF = {"name": "events", "segment_size": 400,
         "dimensions": [{"name": "country"}, {"name": "event_name"}, {"name": "day", "type": "uint"},
                        {"name": "ok", "type": "boolean"}, {"name": "ts", "type": "time", "format": "%Y-%m-%d %H:%M:%S"},
                        {"name": "id", "type": "uint"}],
         "metrics": [{"name": "count", "type": "count"}, {"name": "revenue", "type": "double_sum"},
                     {"name": "best", "type": "int_max"}, {"name": "worst", "type": "float_min"},
                     {"name": "avg_len", "type": "long_avg"}, {"name": "users", "type": "bitset"}]}


# the same table without a COUNT metric: an AVG then divides by the hidden per-row count (store.cc:126-129)
TCONF_NOCOUNT = dict(TCONF, metrics=[m for m in TCONF["metrics"] if m["type"] != "count"])


def _rows(nrows=3000, seed=17):
    rnd = random.Random(seed)
    names = ["open", "buy", "quit", "refund", "review", "rate", "share", "purchase", "donate", "browse"]
    return [[rnd.choice(["US", "IL", "KZ", "RU", "AZ", "CH"]), rnd.choice(names), str(rnd.randrange(0, 90)),
             rnd.choice(["true", "false"]), "2017-06-%02d 10:%02d:00" % (1 + i % 28, i % 60), str(i % 1500),
             str(rnd.randrange(0, 500) / 4), str(rnd.randrange(-50, 50)), str(rnd.randrange(-40, 40) / 8), str(rnd.randrange(1, 30)),
             str(rnd.randrange(0, 60))] for i in range(nrows)]


def _cluster(nworkers=3, nrows=3000, seed=17, tconf=TCONF):
    """nworkers product databases holding a random split of the rows (the same dimension tuple can live on several
    workers; their dictionaries assign different codes), one empty controller, and the oracle holding everything."""
    from oracle import viya_oracle as vo
    from viyadb_amd import hostdb
    rows = _rows(nrows, seed)
    rnd = random.Random(seed + 1)
    parts = [[] for _ in range(nworkers)]
    for r in rows:
        parts[rnd.randrange(nworkers)].append(r)
    workers = []
    for p in parts:
        rnd.shuffle(p)
        w = hostdb.Database({"tables": [tconf]})
        w.load("events", p)
        workers.append(w)
    controller = hostdb.Database({"tables": [tconf]})
    odb = vo.Database({"tables": [tconf]})
    odb.table("events").load(rows)
    return workers, controller, odb


QUERIES = [
    {"dimensions": ["country", "event_name"], "metrics": ["count", "revenue", "best", "worst", "avg_len", "users"],
     "filter": {"op": "ge", "column": "day", "value": "3"}},
    {"dimensions": ["event_name"], "metrics": ["avg_len", "users"], "nocount": True},      # AVG over the hidden count
    {"dimensions": [], "metrics": ["users", "avg_len", "revenue"], "nocount": True},       # one global group
    {"dimensions": [], "metrics": ["count", "users"]},
    {"dimensions": ["ok", "day"], "metrics": ["best", "count"], "filter": {"op": "lt", "column": "day", "value": "20"}},
    {"select": [{"column": "ts", "granularity": "day", "format": "%d/%m"}, {"column": "users"}, {"column": "count"}]},
    {"dimensions": ["country", "id"], "metrics": ["count", "revenue", "users"], "header": True,
     "having": {"op": "and", "filters": [{"op": "gt", "column": "count", "value": "1"}, {"op": "ne", "column": "country", "value": "US"}]},
     "sort": [{"column": "revenue"}, {"column": "id", "ascending": True}, {"column": "country"}], "skip": 3, "limit": 25},
    {"dimensions": ["country"], "metrics": ["users"], "having": {"op": "eq", "column": "country", "value": "KZ"}},
    {"dimensions": ["id"], "metrics": ["best"], "sort": [{"column": "best"}, {"column": "id"}], "limit": 10},
    {"dimensions": ["country"], "metrics": ["count"], "filter": {"op": "eq", "column": "country", "value": "nowhere"}},   # empty partials
]


@pytest.mark.parametrize("qi", range(len(QUERIES)))
def test_merged_partials_equal_one_database(qi):
    q = dict(QUERIES[qi], type="aggregate", table="events")
    workers, controller, odb = _cluster(tconf=TCONF_NOCOUNT if q.pop("nocount", False) else TCONF)
    try:
        want, ost = odb.query(q)
        blobs, scanned = [], 0
        for w in workers:
            blob, st = w.query_partial(q)
            scanned += st["scanned_recs"]
            blobs.append(blob)
        got, gst = controller.query_merge(q, blobs)
        if "sort" in q:
            assert got == want
        else:
            assert sorted(got) == sorted(want)
        assert gst["aggregated_recs"] == ost["aggregated_recs"]
        assert gst["output_recs"] == ost["output_recs"]
        assert gst["scanned_recs"] == scanned
        # any worker can act as the controller as well (its dictionaries already hold some of the strings)
        again, _ = workers[1].query_merge(q, blobs)
        assert sorted(again) == sorted(want)
    finally:
        for d in workers + [controller]:
            d.close()


@pytest.mark.parametrize("flags", ["1", "2049"])      # hash table as separate arrays / as records, on workers and controller alike
def test_merged_partials_through_the_hash_table(flags, monkeypatch):
    monkeypatch.setenv("VIYA_HIP_PLAN_FLAGS", flags)
    for qi in (0, 3, 6):
        test_merged_partials_equal_one_database(qi)


def test_partial_state_content_matches_oracle_partials():
    """A worker's blob, decoded by the independent reader in tests/partial_wire.py, holds exactly the oracle's
    aggregation states for that worker's rows: sums (not averages), the dividing count, the distinct (group, id) pairs."""
    from oracle import viya_oracle as vo
    from viyadb_amd import hostdb
    rows = _rows(1200, 5)
    w = hostdb.Database({"tables": [TCONF_NOCOUNT]})
    odb = vo.Database({"tables": [TCONF_NOCOUNT]})
    try:
        w.load("events", rows)
        odb.table("events").load(rows)
        q = {"type": "aggregate", "table": "events", "dimensions": ["country", "day"], "metrics": ["avg_len", "users", "revenue", "worst"],
             "filter": {"op": "ge", "column": "day", "value": "45"}}
        blob, _ = w.query_partial(q)
        got = pw.decode(blob)
        want = pw.oracle_partial(odb, q)
        assert got["has_hidden"] == 1
        assert pw.canonical(got) == pw.canonical(want)
        # ... and the product merges a blob written by that independent encoder (oracle states -> wire bytes)
        c = hostdb.Database({"tables": [TCONF_NOCOUNT]})
        try:
            rows_out, _ = c.query_merge(q, [pw.encode(want)])
            assert sorted(rows_out) == sorted(odb.query(q)[0])
        finally:
            c.close()
    finally:
        w.close()


def test_merge_of_many_groups_takes_the_hash_path():
    """~60 K distinct (id, day) groups spread over 4 workers: the temporary table's re-aggregation is a hash GROUP BY."""
    from oracle import viya_oracle as vo
    from viyadb_amd import hostdb
    tconf = {"name": "wide", "segment_size": 20000, "dimensions": [{"name": "id", "type": "ulong"}, {"name": "day", "type": "uint"}],
             "metrics": [{"name": "count", "type": "count"}, {"name": "v", "type": "long_sum"}, {"name": "users", "type": "bitset"}]}
    rnd = random.Random(3)
    rows = [[str(rnd.randrange(0, 1 << 40) if i % 3 else i % 20000), str(rnd.randrange(0, 5)), str(rnd.randrange(-1000, 1000)), str(rnd.randrange(0, 9))]
            for i in range(80000)]
    workers = [hostdb.Database({"tables": [tconf]}) for _ in range(4)]
    odb = vo.Database({"tables": [tconf]})
    try:
        for i, w in enumerate(workers):
            w.load("wide", rows[i::4])
        odb.table("wide").load(rows)
        q = {"type": "aggregate", "table": "wide", "dimensions": ["id", "day"], "metrics": ["count", "v", "users"],
             "sort": [{"column": "count"}, {"column": "v"}, {"column": "id"}, {"column": "day"}], "limit": 50}
        blobs = [w.query_partial(q)[0] for w in workers]
        got, gst = workers[0].query_merge(q, blobs)
        want, ost = odb.query(q)
        assert got == want
        assert gst["aggregated_recs"] == ost["aggregated_recs"] > 50000
    finally:
        for w in workers:
            w.close()

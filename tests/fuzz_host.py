"""Bug hunt with the JSON-level differential fuzzer of tests/test_gpu_host_fuzz.py over many seeds (GPU box):
python tests/fuzz_host.py [first_seed] [nseeds] [queries_per_seed]  — prints every discrepancy with its query.
FUZZ_CLUSTER=1: every aggregate query goes through three workers' partial states and the GPU merge instead."""
import json
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import viya_oracle as vo            # noqa: E402  (lives under tests/: only test code may use the oracle)
from tests import test_gpu_host_fuzz as f       # noqa: E402
from viyadb_amd import hostdb                   # noqa: E402

first, nseeds, nq = (int(a) for a in (sys.argv[1:4] + ["5000", "40", "60"][len(sys.argv) - 1:]))
CLUSTER = bool(os.environ.get("FUZZ_CLUSTER"))


class Cluster:
    """Three workers behind the Database interface: load = random split, query = partial states + merge."""

    def __init__(self, tconf, rnd):
        self.rnd = rnd
        self.workers = [hostdb.Database({"tables": [tconf]}) for _ in range(3)]

    def load(self, table, rows, now=None):
        parts = [[], [], []]
        for r in rows:
            parts[self.rnd.randrange(3)].append(r)
        for w, p in zip(self.workers, parts):
            w.load(table, p, now=now)

    def query(self, q, now=None):
        return self.workers[1].query_merge(q, [w.query_partial(q, now=now)[0] for w in self.workers])

    def close(self):
        for w in self.workers:
            w.close()


bad = checked = 0
for seed in range(first, first + nseeds):
    rnd = random.Random(seed)
    tconf = f.make_table(rnd)
    if CLUSTER:
        for d in tconf["dimensions"]:
            if d["name"] == "event":
                d["cardinality"] = 200
    rows = f.make_rows(rnd, tconf, rnd.choice([400, 2500, 9000]))
    gdb = Cluster(tconf, random.Random(seed + 1)) if CLUSTER else hostdb.Database({"tables": [tconf]})
    odb = vo.Database({"tables": [tconf]})
    third = len(rows) // 3
    for b in (rows[:third], rows[third:2 * third], rows[2 * third:]):
        gdb.load("t", b, now=f.NOW)
        odb.table("t").load(b, now=f.NOW)
    for qi in range(nq):
        q = f.make_query(rnd, tconf, rows)
        if CLUSTER and not f._cluster_comparable(q, tconf):
            continue
        try:
            want, ost = odb.query(q, now=f.NOW)
        except vo.OutOfRange:
            continue
        except (vo.Unsupported, vo.InvalidArgument, ValueError, OverflowError) as e:
            try:
                gdb.query(q, now=f.NOW)
                bad += 1
                print("SEED", seed, qi, "oracle rejected (%s), product accepted:" % e, json.dumps(q), json.dumps(tconf))
            except hostdb.HostError:
                pass
            continue
        try:
            got, gst = gdb.query(q, now=f.NOW)
        except hostdb.HostError as e:
            if "stod" in str(e) and any(f._stod_throws(c) for r in want for c in r):
                continue
            bad += 1
            print("SEED", seed, qi, "product error:", e, json.dumps(q), json.dumps(tconf))
            continue
        checked += 1
        if q["type"] == "aggregate" and "sort" not in q:
            ok = len(got) == len(want) if ("limit" in q or "skip" in q) else sorted(got) == sorted(want)
        else:
            ok = got == want
        if not ok:
            bad += 1
            print("SEED", seed, qi, "MISMATCH", json.dumps(q), json.dumps(tconf))
            print("  got ", got[:5], len(got))
            print("  want", want[:5], len(want))
        for k in ("aggregated_recs", "output_recs") if CLUSTER else ("scanned_recs", "scanned_segments", "aggregated_recs", "output_recs"):
            if gst[k] != ost[k]:
                bad += 1
                print("SEED", seed, qi, "STAT", k, gst[k], ost[k], json.dumps(q))
    gdb.close()
print("checked", checked, "queries over", nseeds, "seeds;", bad, "discrepancies")

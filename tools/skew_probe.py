#!/usr/bin/env python3
"""What skew costs: the bench's queries on tables whose keys are NOT the uniform draw of SURVEY 8(d) —
  C3z  C3 with a Zipf-like d0 (one of DENSE_PART's partitions receives several times its share of the tuples),
  C3s  C3 loaded in d3 order (the reference's own scenario, test/index.cc:44-75: segments skipped by min / max, survivors clustered),
  C5h  C5 with one (t, u) pair on a tenth of the rows (one digit of every level of the hashed partitioning, one LDS range),
each next to its uniform twin IN THE SAME PROCESS (same box, same kind of placement), prepared exactly as bench.py prepares its table.
Per table: the FIRST query's wall time (whatever attempts it needs), then the steady state (median wall time and kernel time of `steps`
queries), attempts, table path and kernels. One JSON line per table, then one summary line with the ratios.
usage: skew_probe.py [c3_segments=1000] [c5_segments=125] [steps=15]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viyadb_amd import capi, executor, synth   # noqa: E402
from viyadb_amd.executor import AggPlan        # noqa: E402

c3seg = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
c5seg = int(sys.argv[2]) if len(sys.argv) > 2 else 125
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 15
executor.init(0)
import torch   # noqa: E402  (device synchronisation only)


def run(w, nseg, prepared=True):
    t0 = time.time()
    table = synth.create_device_table(w, nseg, w.segment_rows)
    plan = AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, flags=capi.PLAN_CARD32, groups_hint=w.plan.groups_hint)
    table.prepare(plan)
    if prepared:
        table.pack(table.gather_columns(plan))
        table.predpack(table.filter_columns(plan))
    torch.cuda.synchronize()
    t_build = time.time() - t0
    # the first query of this shape on this table: no vh_table_prepare before it, so whatever the planner learns it learns here
    q0 = time.perf_counter()
    r = table.query_agg(plan, copy=False)
    first_ms = (time.perf_counter() - q0) * 1e3
    first = {"ms": round(first_ms, 3), "attempts": r.retries + 1, "kernel": r.kernel}
    q0 = time.perf_counter()
    r = table.query_agg(plan, copy=False)
    second_ms = (time.perf_counter() - q0) * 1e3
    wall, kern = [], []
    for _ in range(steps):
        q0 = time.perf_counter()
        r = table.query_agg(plan, copy=False)
        wall.append((time.perf_counter() - q0) * 1e3)
        kern.append(r.scan_kernel_ms)
    wall.sort(); kern.sort()
    out = {"workload": w.name, "rows": nseg * w.segment_rows, "segments": nseg, "scanned_segments": r.scanned_segments, "passed": r.passed_recs,
           "groups": r.ngroups, "first_query": first, "second_query_ms": round(second_ms, 3),
           "steady_ms": round(wall[len(wall) // 2], 3), "steady_kernel_ms": round(kern[len(kern) // 2], 3), "steady_attempts": r.retries + 1,
           "path": r.path, "kernel": r.kernel, "build_seconds": round(t_build, 2)}
    table.close()
    print(json.dumps(out), flush=True)
    return out


res = {}
for name, nseg in (("C3", c3seg), ("C3z", c3seg), ("C3s", c3seg), ("C5", c5seg), ("C5h", c5seg)):
    if nseg <= 0:
        continue
    w = synth.WORKLOADS[name](segment_rows=1_000_000) if name != "C3s" else synth.c3s(1_000_000, nseg)
    try:
        res[name] = run(w, nseg)
    except Exception as e:   # noqa: BLE001
        print(json.dumps({"workload": name, "error": str(e)[:300]}), flush=True)
summ = {}
for sk, base in (("C3z", "C3"), ("C3s", "C3"), ("C5h", "C5")):
    if sk in res and base in res:
        summ[sk] = {"steady_over_uniform": round(res[sk]["steady_ms"] / res[base]["steady_ms"], 3),
                    "first_over_uniform_steady": round(res[sk]["first_query"]["ms"] / res[base]["steady_ms"], 3),
                    "first_over_uniform_first": round(res[sk]["first_query"]["ms"] / res[base]["first_query"]["ms"], 3)}
print(json.dumps({"summary": summ, "note": "ratios of wall times; uniform twin measured in the same process; C3s scans fewer segments than C3 (skipped by min / max)"}), flush=True)

#!/usr/bin/env python3
"""What device top-N buys at C5t scale: 100 M rows, ~29 M groups, ORDER BY count DESC LIMIT 10 (C-ABI plan.top)."""
import json, os, sys, time, dataclasses
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viyadb_amd import executor, synth
executor.init(0)
w = synth.c5t()
t = synth.create_device_table(w, int(sys.argv[1]) if len(sys.argv) > 1 else 100)
base = executor.AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics)
for label, plan in (("all groups to the host", base), ("top-10 superset on the device", dataclasses.replace(base, top=(2, True, 10)))):
    for _ in range(3):
        t0 = time.perf_counter()
        r = t.query_agg(plan, copy=False)
        wall = (time.perf_counter() - t0) * 1e3
    print(json.dumps({"case": label, "kernel_ms": round(r.scan_kernel_ms, 2), "device_total_ms": round(r.total_ms, 2), "wall_ms": round(wall, 2),
                      "groups": r.ngroups, "rows_returned": r.returned}))

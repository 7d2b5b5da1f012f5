#!/usr/bin/env python3
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viyadb_amd import executor, synth
from viyadb_amd.executor import AggPlan
seg = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
executor.init(0)
w = synth.c3()
t = synth.create_device_table(w, seg)
npred = int(sys.argv[3]) if len(sys.argv) > 3 else 3
flt = list(w.plan.filter[:npred]) + ([("and", npred)] if npred > 1 else [])
if npred == 0:
    flt = [("rel", 0, 5, 0)]   # d0 >= 0: everything passes, still the fast kernel
plan = AggPlan(filter=flt, groups=w.plan.groups, metrics=w.plan.metrics, flags=flags, groups_hint=100000)
for _ in range(6):
    r = t.query_agg(plan)
print(json.dumps({"kernel_ms": round(r.scan_kernel_ms, 3), "total_ms": round(r.total_ms, 3), "path": r.path, "passed": r.passed_recs, "retries": r.retries}))

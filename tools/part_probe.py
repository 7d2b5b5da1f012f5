#!/usr/bin/env python3
"""DENSE_PART at high selectivity on the C3 table: run under `rocprofv3 --kernel-trace --stats` to split phase 1
(scan + staging) from phase 2 (part_agg_kernel)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viyadb_amd import capi, executor, synth
from viyadb_amd.executor import AggPlan, GroupSpec
seg = int(sys.argv[1]) if len(sys.argv) > 1 else 100
executor.init(0)
w = synth.c3()
t = synth.create_device_table(w, seg)
for name, plan in (("nofilter", AggPlan(filter=[], groups=[GroupSpec(0), GroupSpec(1)], metrics=[7, 9])),
                   ("25pct", AggPlan(filter=[w.plan.filter[0]], groups=[GroupSpec(0), GroupSpec(1)], metrics=[7, 9]))):
    for _ in range(5):
        r = t.query_agg(plan)
    print(json.dumps({"case": name, "kernel_ms": r.scan_kernel_ms, "path": r.path, "passed": r.passed_recs}))

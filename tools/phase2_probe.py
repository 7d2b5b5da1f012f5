#!/usr/bin/env python3
"""C3 table, partitioned plan only, d3 < X: run under rocprofv3 --kernel-trace --stats to see how part_agg_kernel's time moves with
the number of tuples. usage: phase2_probe.py <threshold> [segments]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viyadb_amd import capi, executor, synth
from viyadb_amd.executor import AggPlan
th = int(sys.argv[1]); seg = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
executor.init(0)
w = synth.c3()
t = synth.create_device_table(w, seg)
t.pack(t.gather_columns(w.plan))
filt = [("rel", 2, capi.OP_EQ, 1), ("rel", 3, capi.OP_LT, th), ("rel", 4, capi.OP_GE, 553), ("and", 3)]
plan = AggPlan(filter=filt, groups=w.plan.groups, metrics=w.plan.metrics, flags=capi.PLAN_FORCE_PART | capi.PLAN_FORCE_PACK, groups_hint=100000)
for _ in range(8):
    r = t.query_agg(plan)
print("threshold", th, "passed", r.passed_recs, "kernel_ms", round(r.scan_kernel_ms, 3), r.kernel, flush=True)

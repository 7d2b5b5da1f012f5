#!/usr/bin/env python3
"""Same table, same query, different SCRATCH buffers: every held result keeps its execution context, so the next query gets
a fresh one (its own hipMalloc'ed tuple pool). Prints the partitioned C3 kernel time per context next to the allocation trace.
usage: VH_TRACE_ALLOC=1 scratch_probe.py [contexts]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viyadb_amd import capi, executor, synth
from viyadb_amd.executor import AggPlan

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
executor.init(0)
w = synth.c3()
t = synth.create_device_table(w, 1000)
t.pack(t.gather_columns(w.plan))
plan = AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, flags=capi.PLAN_FORCE_PART, groups_hint=100000)
held = []
for k in range(n):
    ms = []
    for _ in range(6):                      # released contexts are reused first: these six runs share one context
        h = t.query_agg_keep(plan)
        ms.append(t.collect(h, plan).scan_kernel_ms)
        t.discard(h)
    held.append(t.query_agg_keep(plan))     # ... which this one now holds on to
    print(json.dumps({"context": k, "kernel_ms": [round(x, 3) for x in ms]}), flush=True)
for h in held:
    t.discard(h)
t.close()

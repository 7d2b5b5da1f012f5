#!/usr/bin/env python3
"""C5 / C5t on one GPU's share (125 segments): kernel time (HIP events) and wall time per query, knobs from the environment.
usage: c5_probe.py [C5|C5t] [segments] [runs]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viyadb_amd import capi, executor, synth
from viyadb_amd.executor import AggPlan

name = sys.argv[1] if len(sys.argv) > 1 else "C5"
seg = int(sys.argv[2]) if len(sys.argv) > 2 else 125
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 6
executor.init(0)
w = synth.WORKLOADS[name]()
t = synth.create_device_table(w, seg)
plan = AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, groups_hint=w.plan.groups_hint,
               flags=0 if os.environ.get("C5_CARD64") else capi.PLAN_CARD32)
t.prepare(plan)
km, wall = [], []
for i in range(runs + 2):
    t0 = time.perf_counter()
    r = t.query_agg(plan, copy=False)
    dt = time.perf_counter() - t0
    if i >= 2:
        km.append(r.scan_kernel_ms); wall.append(dt * 1e3)
print(json.dumps({"workload": name, "segments": seg, "kernel_ms": round(min(km), 3), "kernel_ms_mean": round(sum(km) / len(km), 3),
                  "wall_ms": round(min(wall), 3), "groups": int(r.ngroups), "passed": int(r.passed_recs), "retries": int(r.retries),
                  "kernel": r.kernel, "env": {k: v for k, v in os.environ.items() if k.startswith("VH_")}}), flush=True)
t.close()

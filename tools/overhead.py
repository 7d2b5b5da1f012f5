#!/usr/bin/env python3
"""Fixed per-query overhead: wall clock vs device total vs dominant kernel, for small shards (what one rank
of an 8-GPU strong-scaling run sees)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from viyadb_amd import executor, synth
from viyadb_amd.executor import AggPlan
executor.init(0, stream=torch.cuda.current_stream().cuda_stream)
w = synth.c3()
for seg in (125, 250, 500, 1000):
    t = synth.create_device_table(w, seg)
    plan = AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, groups_hint=100000)
    for _ in range(3):
        t.query_agg(plan)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    k = tot = 0.0
    for _ in range(n):
        r = t.query_agg(plan)
        k += r.scan_kernel_ms
        tot += r.total_ms
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    print(json.dumps({"segments": seg, "wall_ms": round(wall, 3), "device_total_ms": round(tot / n, 3), "kernel_ms": round(k / n, 3),
                      "overhead_ms": round(wall - k / n, 3), "path": r.path}), flush=True)
    t.close()

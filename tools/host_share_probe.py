#!/usr/bin/env python3
"""Where a query's time goes outside its kernels: per workload, as bench.py prepares and times it — wall time per query in
the Python loop, the library's own total (HIP events), kernel time, and with VH_TIMES=1 the host's plan + enqueue and finalize shares.
usage: host_share_probe.py [C1,C2,C3] [steps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viyadb_amd import capi, executor, synth
from viyadb_amd.executor import AggPlan
names = (sys.argv[1] if len(sys.argv) > 1 else "C1,C2,C3").split(",")
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
executor.init(0)
import torch
for name in names:
    w = synth.WORKLOADS[name]()
    nseg = {"C1": 10, "C2": 100, "C3": 1000, "C3e": 125}.get(name, 100)
    t = synth.create_device_table(w, nseg)
    plan = AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, flags=capi.PLAN_CARD32, groups_hint=w.plan.groups_hint)
    t.prepare(plan)
    t.pack(t.gather_columns(plan)); t.predpack(t.filter_columns(plan)); t.warm(plan)
    for _ in range(5):
        r = t.query_agg(plan, copy=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); k = 0.0; tot = 0.0
    for _ in range(steps):
        r = t.query_agg(plan, copy=False); k += r.scan_kernel_ms; tot += r.total_ms
    wall = (time.perf_counter() - t0) / steps * 1e3
    # the C calls alone (no result views)
    import ctypes as C
    p, keep = t._build_plan(plan)
    t0 = time.perf_counter()
    for _ in range(steps):
        res = C.c_void_p(); t.lib.vh_query_agg(t.handle, C.byref(p), C.byref(res)); t.lib.vh_result_free(res)
    bare = (time.perf_counter() - t0) / steps * 1e3
    print(json.dumps({"workload": name, "segments": nseg, "wall_ms": round(wall, 4), "c_calls_only_ms": round(bare, 4), "events_total_ms": round(tot / steps, 4),
                      "kernel_ms": round(k / steps, 4), "outside_kernels_ms": round(wall - k / steps, 4), "python_share_ms": round(wall - bare, 4), "kernel": r.kernel}), flush=True)
    t.close()

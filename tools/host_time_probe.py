#!/usr/bin/env python3
"""Where a query's HOST time goes (C3 or C2 plan, copy=False): the C call (plan + launches + wait), the device's own span inside it
(vh_result_info.total_ms: first to last event), Python's collect, vh_result_free. usage: python tools/host_time_probe.py [C3|C2] [segments]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                              # noqa: E402
from viyadb_amd import capi, executor, synth   # noqa: E402

torch.cuda.set_device(0)
executor.init(0)
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
w = synth.c3() if name == "C3" else synth.c2()
segs = int(sys.argv[2]) if len(sys.argv) > 2 else (1000 if name == "C3" else 100)
t = synth.create_device_table(w, segs)
plan = executor.AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, groups_hint=w.plan.groups_hint)
t.prepare(plan)
for _ in range(5):
    t.query_agg(plan, copy=False)
p, keep = t._build_plan(plan)
N = 200
tc = tk = tf = dev = 0.0
t_all0 = time.perf_counter()
for _ in range(N):
    a = time.perf_counter()
    p, keep = t._build_plan(plan)
    res = C.c_void_p()
    capi.check(t.lib.vh_query_agg(t.handle, C.byref(p), C.byref(res)))
    b = time.perf_counter()
    r = t._collect(res, plan, False)
    c = time.perf_counter()
    t.lib.vh_result_free(res)
    d = time.perf_counter()
    tc += b - a; tk += c - b; tf += d - c; dev += r.total_ms
t_all = time.perf_counter() - t_all0
print(json.dumps({"workload": name, "segments": segs, "per_query_us": round(t_all / N * 1e6, 1), "c_call_us": round(tc / N * 1e6, 1),
                  "device_span_us": round(dev / N * 1e3, 1), "c_call_minus_device_us": round((tc / N - dev / N * 1e-3) * 1e6, 1),
                  "collect_us": round(tk / N * 1e6, 1), "free_us": round(tf / N * 1e6, 1)}))
t.close()

#!/usr/bin/env python3
"""Where the host-side part of the fixed per-query cost goes (plan marshalling, the C call, result views)."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viyadb_amd import capi, executor, synth
from viyadb_amd.executor import AggPlan
executor.init(0)
w = synth.c3()
t = synth.create_device_table(w, 125)
plan = AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, groups_hint=100000)
t.prepare(plan)
t.pack(t.gather_columns(plan))
t.narrow(t.filter_columns(plan))
for _ in range(5):
    t.query_agg(plan, copy=False)
n = 200
tb = tc = tv = 0.0
for _ in range(n):
    a = time.perf_counter()
    p, keep = t._build_plan(plan)
    b = time.perf_counter()
    res = C.c_void_p()
    capi.check(t.lib.vh_query_agg(t.handle, C.byref(p), C.byref(res)))
    c = time.perf_counter()
    r = t._collect(res, plan, False)
    t.lib.vh_result_free(res)
    d = time.perf_counter()
    tb += b - a; tc += c - b; tv += d - c
print(json.dumps({"build_plan_us": tb / n * 1e6, "vh_query_agg_us": tc / n * 1e6, "collect_us": tv / n * 1e6, "kernel_ms": r.scan_kernel_ms, "device_total_ms": r.total_ms}))

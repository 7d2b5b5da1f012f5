#!/usr/bin/env python3
"""Where the C3 scan's predicate columns come from, A/B/C inside ONE process (boxes and processes differ by more than the variants do): narrow
copies (5 bytes per row), the byte-plane predicate projection (3), the bit-sliced one (2.75, comparisons bit-serial on 32 rows per lane), each
with the payload gathered from the 4-byte bit-field records; rounds interleaved, median kernel time (both phases, HIP events) and per-query
wall time per variant. usage: pred_ab.py [segments] [rounds]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viyadb_amd import capi, executor, synth
from viyadb_amd.executor import AggPlan
seg = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
executor.init(0)
w = synth.c3()
t = synth.create_device_table(w, seg)
base = AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, groups_hint=w.plan.groups_hint)
t.pack(t.gather_columns(base), compressed=True)
cols = t.filter_columns(base)
t.narrow(cols); t.predpack(cols, sliced=False); t.predpack(cols, sliced=True)
variants = [("narrow", capi.PLAN_NO_PREDPACK), ("byte_planes", capi.PLAN_NO_SLICED), ("bit_sliced", 0), ("arenas", capi.PLAN_NO_NARROW)]
plans = {n: AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, groups_hint=w.plan.groups_hint, flags=f) for n, f in variants}
for n, p in plans.items():
    t.prepare(p)
    for _ in range(3):
        t.query_agg(p, copy=False)
ms = {n: [] for n in plans}
wall = {n: [] for n in plans}
last = {}
for _ in range(rounds):
    for n, p in plans.items():
        k = []
        t0 = time.perf_counter()
        for _ in range(5):
            r = t.query_agg(p, copy=False)
            k.append(r.scan_kernel_ms)
        wall[n].append((time.perf_counter() - t0) / 5 * 1e3)
        ms[n].append(sorted(k)[2])
        last[n] = r
for n in plans:
    r = last[n]
    print(json.dumps({"variant": n, "kernel_ms": round(sorted(ms[n])[len(ms[n]) // 2], 4), "min": round(min(ms[n]), 4), "max": round(max(ms[n]), 4),
                      "per_query_ms": round(sorted(wall[n])[len(wall[n]) // 2], 4), "predpack": r.predpack, "sliced": r.sliced, "narrow": r.narrow, "packed": r.packed,
                      "kernel": r.kernel, "path": r.path}), flush=True)
t.close()

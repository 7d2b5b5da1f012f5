#!/usr/bin/env python3
"""Streamed payload records against gathered ones (VhJitShape::qpay), per selectivity, on the C3 table at full size: the compiled compacting
scan with the bit-field projection and the predicate projection built, VH_PLAN_FORCE_QPAY against VH_PLAN_NO_QPAY, kernel time of both
phases (HIP events). The crossover decides VH_QPAY_MIN_SEL. usage: qpay_probe.py [segments]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viyadb_amd import capi, executor, synth
from viyadb_amd.executor import AggPlan
seg = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
executor.init(0)
w = synth.c3()
t = synth.create_device_table(w, seg)
base = AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, groups_hint=w.plan.groups_hint)
t.pack(t.gather_columns(base), compressed=True)
t.predpack(t.filter_columns(base))
# d2 == 1 (25 %) & d3 < a (a / 1000) & d4 >= b ((1000 - b) / 1000)
cases = [("0.25pct", 100, 900), ("1pct", 200, 800), ("2.5pct", 316, 684), ("5pct", 447, 553), ("10pct", 632, 368), ("25pct", 1000, 0)]
for name, a, b in cases:
    flt = [("rel", 2, capi.OP_EQ, 1), ("rel", 3, capi.OP_LT, a), ("rel", 4, capi.OP_GE, b), ("and", 3)]
    for label, flags in (("stream", capi.PLAN_FORCE_QPAY), ("gather", capi.PLAN_NO_QPAY)):
        plan = AggPlan(filter=flt, groups=w.plan.groups, metrics=w.plan.metrics, flags=flags | capi.PLAN_FORCE_PACK, groups_hint=w.plan.groups_hint)
        ms = []
        for _ in range(6):
            r = t.query_agg(plan, copy=False)
            ms.append(r.scan_kernel_ms)
        print(json.dumps({"case": name, "variant": label, "kernel_ms": round(sorted(ms[1:])[2], 3), "path": r.path, "streamed": r.streamed_payload, "predpack": r.predpack,
                          "sel": round(r.passed_recs / max(1, r.scanned_recs), 4), "kernel": r.kernel}), flush=True)
for name, flt in (("50pct", [("rel", 3, capi.OP_LT, 500)]), ("100pct", [])):
    for label, flags in (("stream", capi.PLAN_FORCE_QPAY | capi.PLAN_FORCE_PACK | capi.PLAN_FORCE_PART), ("arenas_default", capi.PLAN_FORCE_PART | capi.PLAN_NO_PACK)):
        plan = AggPlan(filter=flt, groups=w.plan.groups, metrics=w.plan.metrics, flags=flags, groups_hint=w.plan.groups_hint)
        ms = []
        for _ in range(5):
            r = t.query_agg(plan, copy=False)
            ms.append(r.scan_kernel_ms)
        print(json.dumps({"case": name, "variant": label, "kernel_ms": round(sorted(ms[1:])[1], 3), "path": r.path, "streamed": r.streamed_payload, "lanes": r.lanes,
                          "sel": round(r.passed_recs / max(1, r.scanned_recs), 4), "kernel": r.kernel}), flush=True)
t.close()

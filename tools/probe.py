#!/usr/bin/env python3
"""Ablation on the C3 table: which part of the fused kernel costs what (1 GPU)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viyadb_amd import capi, executor, synth
from viyadb_amd.executor import AggPlan, GroupSpec

seg = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
executor.init(0)
w = synth.c3()
t = synth.create_device_table(w, seg)
F3 = w.plan.filter
variants = {
    "scan_only(3 pred cols, no payload)": AggPlan(filter=F3, groups=[], metrics=[]),
    "scan+1 metric gather (m0 i64)": AggPlan(filter=F3, groups=[], metrics=[7]),
    "scan+count gather (u32)": AggPlan(filter=F3, groups=[], metrics=[9]),
    "scan+2 group gathers, no metrics": AggPlan(filter=F3, groups=[GroupSpec(0), GroupSpec(1)], metrics=[]),
    "full C3": AggPlan(filter=F3, groups=[GroupSpec(0), GroupSpec(1)], metrics=[7, 9], groups_hint=100000),
    "full C3 direct atomics (NO_PART)": AggPlan(filter=F3, groups=[GroupSpec(0), GroupSpec(1)], metrics=[7, 9], flags=16),
    "full C3, 1 pred col (d2==1, 25%)": AggPlan(filter=[F3[0]], groups=[GroupSpec(0), GroupSpec(1)], metrics=[7, 9]),
    "no filter, full payload (100%)": AggPlan(filter=[], groups=[GroupSpec(0), GroupSpec(1)], metrics=[7, 9]),
    "sel 0.5% (d3<5 & d2==1 ...)": AggPlan(filter=[("rel", 2, capi.OP_EQ, 1), ("rel", 3, capi.OP_LT, 45), ("rel", 4, capi.OP_GE, 553), ("and", 3)],
                                         groups=[GroupSpec(0), GroupSpec(1)], metrics=[7, 9]),
}
rows = seg * w.segment_rows
for name, plan in variants.items():
    ms = []
    for _ in range(5):
        r = t.query_agg(plan)
        ms.append(r.scan_kernel_ms)
    k = sorted(ms)[2]
    print(json.dumps({"variant": name, "kernel_ms": round(k, 3), "Grows_s": round(rows / k / 1e6, 1), "bref_GBs": round(r.algorithmic_bytes / k / 1e6),
                      "total_ms": round(r.total_ms, 3), "passed": r.passed_recs, "groups": r.ngroups, "path": r.path, "fast": r.fast,
                      "retries": r.retries}), flush=True)

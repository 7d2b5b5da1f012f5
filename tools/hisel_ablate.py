#!/usr/bin/env python3
"""Where phase 1 of DENSE_PART goes when every row passes (C3 table, GROUP BY d0, d1, SUM + COUNT from the arenas, compiled compacting
kernel): run under VH_JIT_ABLATE=1 (no gathers) / 2 (gathers only), VH_JIT_FLAGS=-DVJ_ABL=8 (no tuple append), VH_ABLATE_NO_PHASE2=1.
Timing only — the ablated runs answer wrongly. usage: hisel_ablate.py [segments]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viyadb_amd import capi, executor, synth
from viyadb_amd.executor import AggPlan, GroupSpec
seg = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
executor.init(0)
w = synth.c3()
t = synth.create_device_table(w, seg)
plan = AggPlan(filter=[], groups=[GroupSpec(0), GroupSpec(1)], metrics=[7, 9], flags=capi.PLAN_FORCE_PART | capi.PLAN_NO_PACK | capi.PLAN_NO_LANES, groups_hint=100000)
ms = []
for _ in range(5):
    r = t.query_agg(plan)
    ms.append(r.scan_kernel_ms)
print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("VH_")}, "kernel_ms": round(sorted(ms[1:])[1], 3), "kernel": r.kernel, "jit": r.jit}))

#!/usr/bin/env python3
"""What would the C3 scan cost if the survivors' payload were STREAMED (a 4-byte record per row read in full, like a predicate column)
instead of gathered (one 128-byte line per survivor)? Emulated with what exists: an always-true fourth predicate on a 4-byte arena adds
the stream; VH_JIT_ABLATE=1 (environment) removes the gathers; VH_ABLATE_NO_PHASE2=1 times phase 1 alone.
usage: [VH_JIT_ABLATE=1] [VH_ABLATE_NO_PHASE2=1] c3_stream_probe.py [extra4]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viyadb_amd import capi, executor, synth
from viyadb_amd.executor import AggPlan

extra = len(sys.argv) > 1 and sys.argv[1] == "extra4"
executor.init(0)
w = synth.c3()
t = synth.create_device_table(w, 1000)
t.pack(t.gather_columns(w.plan)); t.narrow(t.filter_columns(w.plan))
flt = list(w.plan.filter[:3]) + ([("rel", 5, capi.OP_GE, 0)] if extra else []) + [("and", 4 if extra else 3)]
plan = AggPlan(filter=flt, groups=w.plan.groups, metrics=w.plan.metrics, groups_hint=100000, flags=capi.PLAN_FORCE_PART)
ms = []
for _ in range(6):
    r = t.query_agg(plan, copy=False)
    ms.append(r.scan_kernel_ms)
print(json.dumps({"extra4": extra, "kernel_ms": round(min(ms[2:]), 3), "passed": int(r.passed_recs), "kernel": r.kernel,
                  "env": {k: v for k, v in os.environ.items() if k.startswith("VH_")}}), flush=True)
t.close()

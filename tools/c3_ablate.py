#!/usr/bin/env python3
"""C3 (1 B rows) through one library build (VIYA_HIP_LIB): kernel time of direct / partitioned aggregation, packed or not.
Used with the VH_ABLATE variant builds of tools/build_variant.py (results of those builds are wrong by design)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viyadb_amd import capi, executor, synth
from viyadb_amd.executor import AggPlan
executor.init(0)
w = synth.c3()
t = synth.create_device_table(w, int(sys.argv[1]) if len(sys.argv) > 1 else 1000)
t.pack(t.gather_columns(w.plan))
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
for label, flags in (("direct", capi.PLAN_NO_PART | capi.PLAN_NO_PACK), ("direct+pack", capi.PLAN_NO_PART | capi.PLAN_FORCE_PACK),
                     ("part", capi.PLAN_FORCE_PART | capi.PLAN_NO_PACK), ("part+pack", capi.PLAN_FORCE_PART | capi.PLAN_FORCE_PACK)):
    if only and label not in only:
        continue
    plan = AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, flags=flags, groups_hint=100000)
    ms = []
    for _ in range(6):
        r = t.query_agg(plan)
        ms.append(r.scan_kernel_ms)
    print(json.dumps({"lib": os.environ.get("VIYA_HIP_LIB", "default").split("/")[-2:-1], "variant": label, "kernel_ms": round(sorted(ms[1:])[2], 3), "path": r.path, "passed": r.passed_recs}), flush=True)

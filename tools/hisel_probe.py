#!/usr/bin/env python3
"""DENSE_PART at high selectivity on the C3 table (item 8 of VERDICT r02): phase 1 as the pre-built no-compaction "lanes" kernel (the
default from 50 % pass up) against the per-query compiled compacting kernel with whole-line tuple writes (PLAN_NO_LANES), per
selectivity. Prints kernel time (both phases, HIP events) per variant. usage: hisel_probe.py [segments]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viyadb_amd import capi, executor, synth
from viyadb_amd.executor import AggPlan, GroupSpec
seg = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
executor.init(0)
w = synth.c3()
t = synth.create_device_table(w, seg)
G = [GroupSpec(0), GroupSpec(1)]
cases = (("100pct", []), ("50pct", [("rel", 3, capi.OP_LT, 500)]), ("25pct", [w.plan.filter[0]]))
for name, filt in cases:
    for label, flags in (("default", 0), ("no_lanes", capi.PLAN_NO_LANES), ("no_lanes_nojit", capi.PLAN_NO_LANES | capi.PLAN_NO_JIT), ("force_lanes", capi.PLAN_FORCE_LANES)):
        plan = AggPlan(filter=filt, groups=G, metrics=[7, 9], flags=flags | capi.PLAN_FORCE_PART | capi.PLAN_NO_PACK, groups_hint=100000)
        ms = []
        try:
            for _ in range(5):
                r = t.query_agg(plan)
                ms.append(r.scan_kernel_ms)
        except Exception as e:  # noqa: BLE001
            print(json.dumps({"case": name, "variant": label, "error": str(e)[:200]})); continue
        print(json.dumps({"case": name, "variant": label, "kernel_ms": round(sorted(ms[1:])[1], 3), "total_ms": round(r.total_ms, 3), "path": r.path,
                          "lanes": r.lanes, "jit": r.jit, "kernel": r.kernel, "sel": round(r.passed_recs / max(1, r.scanned_recs), 4)}), flush=True)

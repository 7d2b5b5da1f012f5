#!/usr/bin/env python3
"""C3 table (1 B rows by default), selectivity sweep d3 < X with the other two predicates fixed: scan kernel time of
direct global atomics vs radix-partitioned LDS aggregation, each gathering from the column arenas or from the payload
projection. Prints one JSON line per (selectivity, variant). usage: c3_sweep.py [segments] [thresholds,comma]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viyadb_amd import capi, executor, synth
from viyadb_amd.executor import AggPlan

seg = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ths = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [90, 180, 270, 447, 540, 720, 1000]
executor.init(0)
w = synth.c3()
t = synth.create_device_table(w, seg)
t.pack(t.gather_columns(w.plan))
NOPACK, PART, DIRECT = capi.PLAN_NO_PACK, capi.PLAN_FORCE_PART, capi.PLAN_NO_PART
for th in ths:
    filt = [("rel", 2, capi.OP_EQ, 1), ("rel", 3, capi.OP_LT, th), ("rel", 4, capi.OP_GE, 553), ("and", 3)]
    for label, flags in (("direct", DIRECT | NOPACK), ("direct+pack", DIRECT | capi.PLAN_FORCE_PACK), ("part", PART | NOPACK), ("part+pack", PART | capi.PLAN_FORCE_PACK), ("default", 0)):
        plan = AggPlan(filter=filt, groups=w.plan.groups, metrics=w.plan.metrics, flags=flags, groups_hint=100000)
        ms = []
        for _ in range(6):
            r = t.query_agg(plan)
            ms.append(r.scan_kernel_ms)
        k = sorted(ms[1:])[2]
        print(json.dumps({"d3_lt": th, "sel": round(r.passed_recs / r.scanned_recs, 4), "variant": label, "kernel_ms": round(k, 3), "total_ms": round(r.total_ms, 3),
                          "path": r.path, "packed": r.packed, "lanes": r.lanes, "retries": r.retries}), flush=True)

#!/usr/bin/env python3
"""C3 table, GROUP BY (d5, d2) = 4 M groups (512 LDS-sized ranges: two partition levels), filter d0 < X: scan kernel time of
direct global atomics vs two-level partitioning (+ what the library picks). One JSON line per (selectivity, variant).
usage: part2_probe.py [segments] [thresholds,comma]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viyadb_amd import capi, executor, synth
from viyadb_amd.executor import AggPlan, GroupSpec

seg = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ths = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [20, 50, 80, 120, 250, 500, 1000]
executor.init(0)
w = synth.c3()
t = synth.create_device_table(w, seg)
groups = [GroupSpec(5), GroupSpec(2)]
t.pack(t.gather_columns(AggPlan(filter=[], groups=groups, metrics=w.plan.metrics)))
for th in ths:
    filt = [("rel", 0, capi.OP_LT, th)]
    for label, flags in (("direct", capi.PLAN_NO_PART2), ("two-level", capi.PLAN_FORCE_PART), ("default", 0)):
        plan = AggPlan(filter=filt, groups=groups, metrics=w.plan.metrics, flags=flags, groups_hint=4 << 20)
        ms = []
        for _ in range(5):
            r = t.query_agg(plan)
            ms.append(r.scan_kernel_ms)
        k = sorted(ms[1:])[1]
        print(json.dumps({"d0_lt": th, "sel": round(r.passed_recs / r.scanned_recs, 4), "variant": label, "kernel_ms": round(k, 3), "total_ms": round(r.total_ms, 3),
                          "groups": r.ngroups, "path": r.path, "kernel": r.kernel, "packed": r.packed, "lanes": r.lanes, "retries": r.retries}), flush=True)

#!/usr/bin/env python3
"""400 K dense groups (d0 x d1 x d2 of the C3 table) at 100 % / 25 % selectivity: partitioned vs direct atomics."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viyadb_amd import capi, executor, synth
from viyadb_amd.executor import AggPlan, GroupSpec
seg = int(sys.argv[1]) if len(sys.argv) > 1 else 200
executor.init(0)
w = synth.c3()
t = synth.create_device_table(w, seg)
groups = [GroupSpec(0), GroupSpec(1), GroupSpec(2)]
for name, flt in (("nofilter", []), ("25pct", [w.plan.filter[0]])):
    for flags, label in ((0, "default"), (16, "no_part")):
        for _ in range(3):
            r = t.query_agg(AggPlan(filter=flt, groups=groups, metrics=[7, 9], flags=flags))
        print(json.dumps({"case": name, "variant": label, "kernel_ms": round(r.scan_kernel_ms, 3), "path": r.path, "groups": r.ngroups, "lanes": r.lanes,
                          "Grows_s": round(seg * 1e6 / r.scan_kernel_ms / 1e6, 1)}))

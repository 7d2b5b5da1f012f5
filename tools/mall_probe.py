#!/usr/bin/env python3
"""Does a SMALL tuple pool (one that fits the 256 MB Infinity Cache) run the partitioned C3 scan faster per row than the whole-table
pool? The same table and query through size() snapshots that show `win` segments at a time; kernel time per window summed over the
table against the whole-table kernel. (Phase 2's fixed cost is paid per window: rocprofv3 --kernel-trace separates the scan.)
usage: mall_probe.py [window segments ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viyadb_amd import capi, executor, synth
from viyadb_amd.executor import AggPlan

executor.init(0)
w = synth.c3()
NSEG = 1000
t = synth.create_device_table(w, NSEG)
t.pack(t.gather_columns(w.plan)); t.narrow(t.filter_columns(w.plan))
base = dict(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, groups_hint=100000, flags=capi.PLAN_FORCE_PART)
full = AggPlan(**base)
for _ in range(4):
    r = t.query_agg(full, copy=False)
print(json.dumps({"window": NSEG, "kernel_ms": round(r.scan_kernel_ms, 3), "path": r.path}), flush=True)
for win in [int(a) for a in sys.argv[1:]] or [100, 50, 200]:
    tot = 0.0
    for rep in range(2):
        tot = 0.0
        for lo in range(0, NSEG, win):
            snap = [0] * NSEG
            for s in range(lo, min(NSEG, lo + win)):
                snap[s] = w.segment_rows
            r = t.query_agg(AggPlan(seg_rows=snap, **base), copy=False)
            tot += r.scan_kernel_ms
    print(json.dumps({"window": win, "sum_kernel_ms": round(tot, 3), "windows": (NSEG + win - 1) // win, "path": r.path}), flush=True)
t.close()

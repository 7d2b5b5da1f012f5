#!/usr/bin/env python3
"""Does the partitioned C3 kernel's time depend on where this process' buffers land? Re-creates the table several times in ONE
process (optionally holding on to dummy allocations in between so that addresses move) and prints the kernel time next to the
allocation trace (VH_TRACE_ALLOC=1). usage: VH_TRACE_ALLOC=1 placement_probe.py [rounds] [segments]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from viyadb_amd import capi, executor, synth
from viyadb_amd.executor import AggPlan

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
seg = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
executor.init(0)
w = synth.c3()
hold = []
for i in range(rounds):
    print("round", i, file=sys.stderr, flush=True)
    t = synth.create_device_table(w, seg)
    t.pack(t.gather_columns(w.plan))
    out = {}
    for label, flags in (("part", capi.PLAN_FORCE_PART), ("direct", capi.PLAN_NO_PART)):
        plan = AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, flags=flags, groups_hint=100000)
        ms = [t.query_agg(plan).scan_kernel_ms for _ in range(8)]
        out[label] = round(sorted(ms[2:])[3], 3)
    print(json.dumps({"round": i, **out}), flush=True)
    t.close()
    hold.append(torch.empty((37 + 61 * i) << 20, dtype=torch.uint8, device="cuda"))     # shift what comes next

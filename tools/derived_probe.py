#!/usr/bin/env python3
"""Does the C3 kernel's class follow the DERIVED layouts (payload projection + narrow predicate copies) when the tuple pool stays where it
is? One process, one execution context (its scratch buffer is placed once): after every round the projection and the narrow copies are
dropped, a filler of growing size is held so that the next ones land elsewhere, and they are rebuilt. Prints kernel ms per round next to
where the buffers landed (VH_TRACE_ALLOC=1). usage: VH_TRACE_ALLOC=1 derived_probe.py [rounds]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from viyadb_amd import executor, synth
from viyadb_amd.executor import AggPlan
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
executor.init(0)
w = synth.c3()
t = synth.create_device_table(w, 1000)
plan = AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, groups_hint=w.plan.groups_hint)
hold = []
for k in range(rounds):
    t.pack(t.gather_columns(plan))
    t.narrow(t.filter_columns(plan))
    ms = [t.query_agg(plan, copy=False).scan_kernel_ms for _ in range(6)]
    print(json.dumps({"round": k, "kernel_ms": round(sorted(ms[2:])[1], 3), "held_gb": sum(h.numel() for h in hold) >> 30}), flush=True)
    t.unpack()
    hold.append(torch.empty((7 + 5 * k) << 30, dtype=torch.uint8, device="cuda"))
t.close()

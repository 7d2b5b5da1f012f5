#!/usr/bin/env python3
"""A/B of VH_TEST_* knobs inside ONE process (same table, same placement): the knob is switched between queries of C3's plan.
usage: python tools/env_ab_probe.py <segments> <spec> [<spec> ...]   spec: '-' (nothing set) or NAME=VALUE[,NAME=VALUE]
env PREPARE=0: no vh_table_prepare."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["VH_TEST_HOOKS"] = "1"          # the gate in front of the library's test hooks (viya_hip.hip test_env)
import torch                              # noqa: E402
from viyadb_amd import executor, synth   # noqa: E402

torch.cuda.set_device(0)
executor.init(0)
segs = int(sys.argv[1])
specs = sys.argv[2:] or ["-"]
w = synth.c3()
t = synth.create_device_table(w, segs)
plan = executor.AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, groups_hint=w.plan.groups_hint)
if os.environ.get("PREPARE", "1") == "1":
    t.prepare(plan)
t.pack(t.gather_columns(plan))
t.narrow(t.filter_columns(plan))
names = sorted({kv.split("=")[0] for s in specs if s != "-" for kv in s.split(",")})
for rnd in range(2):
    for spec in specs:
        for n in names:
            os.environ.pop(n, None)
        if spec != "-":
            for kv in spec.split(","):
                k, v = kv.split("=")
                os.environ[k] = v
        ks, qs = [], []
        for i in range(30):
            r = t.query_agg(plan, copy=False)
            if i >= 5:
                ks.append(r.scan_kernel_ms)
        ks.sort()
        print(json.dumps({"segments": segs, "spec": spec, "round": rnd, "kernel_ms_median": round(ks[len(ks) // 2], 4), "min": round(ks[0], 4),
                          "groups": int(r.ngroups)}), flush=True)
t.close()

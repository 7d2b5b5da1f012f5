#!/usr/bin/env python3
"""What one rank of the N-GPU strong-scaling run (C4) does per step, measured on ONE GPU: its share of the segments through
vh_query_agg (the single-GPU entry point) and through vh_query_agg_sharded with a one-rank RCCL communicator
(VH_TEST_SHARDED_WORLD1: plan agreement, verdict all-reduce, ncclReduce of the partial table, emission — launch and
sync cost of every collective, no wire). A model of the per-rank cost, not a scaling measurement: the driver's
`bench.py --gpus N` runs are the measurement (tools/scale_curve.sh).
usage: python tools/scale_proxy.py [N ...]   (default 1 2 4 8)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
os.environ["VH_TEST_HOOKS"] = "1"            # the gate in front of the library's test hooks (viya_hip.hip test_env)
os.environ["VH_TEST_SHARDED_WORLD1"] = "1"
import torch                              # noqa: E402
import torch.distributed as dist          # noqa: E402
from viyadb_amd import distributed, executor, synth   # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=0, world_size=1)
executor.init(0)
comm = distributed.Comm.rccl(dist)
w = synth.c3()
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    segs = 1000 // n
    t = synth.create_device_table(w, segs)
    plan = executor.AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, groups_hint=w.plan.groups_hint)
    t.prepare(plan)
    t.pack(t.gather_columns(plan))
    t.narrow(t.filter_columns(plan))
    out = {}
    for label, fn in (("query_only_ms", lambda: t.query_agg(plan, copy=False)),
                      ("sharded_1rank_ms", lambda: distributed.sharded_query(t, plan, comm, root=0, copy=False))):
        for _ in range(5):
            r = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            r = fn()
        torch.cuda.synchronize()
        out[label] = round((time.perf_counter() - t0) / 50 * 1e3, 4)
    out.update(n=n, segments=segs, kernel_ms=round(r.scan_kernel_ms, 4), path=r.path,
               implied_G_rows_s=round(1000 * w.segment_rows / (out["sharded_1rank_ms"] * 1e-3) / 1e9, 1))
    print(json.dumps(out), flush=True)
    t.close()
comm.close()
dist.destroy_process_group()

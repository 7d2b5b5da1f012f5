#!/usr/bin/env python3
"""What one rank of the N-GPU strong-scaling run (C4) does per step, measured on one GPU: its share of the segments,
the launch -> RCCL reduce (1-rank communicator: launch and sync cost, no wire) -> finalise sequence of bench.py.
usage: python tools/scale_proxy.py [N ...]   (default 1 2 4 8)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
import torch                              # noqa: E402
import torch.distributed as dist          # noqa: E402
from viyadb_amd import distributed, executor, synth   # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
executor.init(0, stream=torch.cuda.current_stream().cuda_stream)
w = synth.c3()
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    segs = 1000 // n
    t = synth.create_device_table(w, segs)
    plan = executor.AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, groups_hint=w.plan.groups_hint)
    out = {}
    for label, force in (("query_only", False), ("launch+reduce+finalise", True)):
        for _ in range(5):
            r = distributed.sharded_query(torch, dist, t, plan, 1, copy=False, force_collectives=force)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            r = distributed.sharded_query(torch, dist, t, plan, 1, copy=False, force_collectives=force)
        torch.cuda.synchronize()
        out[label] = (time.perf_counter() - t0) / 50 * 1e3
    out.update(n=n, segments=segs, kernel_ms=r.scan_kernel_ms,
               implied_G_rows_s=1000 * w.segment_rows / (out["launch+reduce+finalise"] * 1e-3) / 1e9)
    print(json.dumps(out), flush=True)
    t.close()
dist.destroy_process_group()

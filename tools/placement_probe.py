#!/usr/bin/env python3
"""Where C3's time depends on WHERE things lie (profiles/r06/NOTES.md, "Placement"). One table of 1 B rows per experiment, kernel time = median of
8 queries' HIP-event time:
  contexts  six execution contexts (six scratch allocations: tuple pool, tables) on one table — do they differ?
  rebuild   the derived layouts dropped and built again behind spacers of 3 GB, sixteen times, with their addresses (VH_TRACE_ALLOC)
  prepare   eight tables one after the other, vh_table_prepare's candidates as it tries them (VH_TIMES) and the time afterwards
usage: placement_probe.py contexts|rebuild|prepare"""
import json
import os
import sys
import time

mode = sys.argv[1] if len(sys.argv) > 1 else "prepare"
os.environ["VH_TIMES" if mode == "prepare" else "VH_TRACE_ALLOC"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viyadb_amd import capi, executor, synth   # noqa: E402
from viyadb_amd.executor import AggPlan        # noqa: E402
import torch                                    # noqa: E402

executor.init(0)
w = synth.c3()


def med(t, plan, n=10):
    ks = []
    for _ in range(n):
        r = t.query_agg(plan, copy=False)
        ks.append(r.scan_kernel_ms)
        del r
    ks = sorted(ks[2:])
    return round(ks[len(ks) // 2], 4)


def new_table():
    t = synth.create_device_table(w, 1000)
    plan = AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, flags=capi.PLAN_CARD32, groups_hint=w.plan.groups_hint)
    return t, plan


if mode == "contexts":
    os.environ["VH_PREPARE_PLACE"] = "0"
    t, plan = new_table()
    t.pack(t.gather_columns(plan)); t.predpack(t.filter_columns(plan)); t.warm(plan)
    held = []
    for ctx in range(6):
        ks = []
        for i in range(12):
            r = t.query_agg(plan, copy=False)
            ks.append(r.scan_kernel_ms)
            if i < 11:
                del r
        ks = sorted(ks[2:])
        print(json.dumps({"context": ctx, "kernel_ms_median": round(ks[len(ks) // 2], 4), "min": round(ks[0], 4)}), flush=True)
        held.append(r)          # keeps this context busy: the next queries take another one (another scratch allocation)
elif mode == "rebuild":
    os.environ["VH_PREPARE_PLACE"] = "0"
    t, plan = new_table()
    hold = []
    for rebuild in range(16):
        print("REBUILD", rebuild, file=sys.stderr, flush=True)
        t.pack(t.gather_columns(plan)); t.predpack(t.filter_columns(plan)); t.warm(plan)
        print(json.dumps({"rebuild": rebuild, "kernel_ms_median": med(t, plan)}), flush=True)
        t.unpack()
        hold.append(torch.empty((3 << 30) + rebuild * (2 << 20), dtype=torch.uint8, device="cuda"))
else:
    hold = []
    for tbl in range(8):
        t, plan = new_table()
        t.pack(t.gather_columns(plan)); t.predpack(t.filter_columns(plan))
        before = med(t, plan)
        t0 = time.time(); t.warm(plan); torch.cuda.synchronize(); tw = time.time() - t0
        print(json.dumps({"table": tbl, "before_prepare": before, "prepare_seconds": round(tw, 3), "after_prepare": med(t, plan)}), flush=True)
        t.close()
        hold.append(torch.empty((2 << 30) + tbl * (6 << 20), dtype=torch.uint8, device="cuda"))

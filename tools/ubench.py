#!/usr/bin/env python3
"""Kernel-level sweep on one GPU: streaming-read ceiling, then each workload through each
aggregate-table organisation. Prints one line per variant (kernel ms, rows/s, B_ref GB/s)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viyadb_amd import executor, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--segments", type=int, default=100)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--workloads", default="C3,C2,C1")
    ap.add_argument("--variants", default="")
    args = ap.parse_args()
    executor.init(0)
    bw = executor.measure_read_bandwidth(8 << 30, 5)
    print(json.dumps({"read_bw_GBs": bw / 1e9}), flush=True)
    for name in args.workloads.split(","):
        w = synth.WORKLOADS[name]()
        if name == "C5t" and args.segments > 200:
            pass
        t = synth.create_device_table(w, args.segments)
        want = set(args.variants.split(",")) if args.variants else None
        for flags, label in [(0, "default"), (16, "no_part"), (8, "generic_kernel"), (4, "no_xcd_private"), (2, "force_global"), (10, "generic_force_global"), (1, "hash"), (9, "generic_hash")]:
            if want and label not in want:
                continue
            plan = executor.AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, flags=flags,
                                    groups_hint=w.plan.groups_hint)
            ms, tot = [], []
            for _ in range(args.iters):
                r = t.query_agg(plan)
                ms.append(r.scan_kernel_ms)
                tot.append(r.total_ms)
            k = sorted(ms)[len(ms) // 2]
            rows = args.segments * w.segment_rows
            print(json.dumps({"workload": name, "variant": label, "path": r.path, "kernel_ms": round(k, 4),
                              "total_ms": round(sorted(tot)[len(tot) // 2], 4), "rows_per_s": rows / (k * 1e-3),
                              "bref_GBs": r.algorithmic_bytes / (k * 1e-3) / 1e9, "groups": r.ngroups,
                              "passed": r.passed_recs, "retries": r.retries, "fast": r.fast}), flush=True)
        t.close()


if __name__ == "__main__":
    main()

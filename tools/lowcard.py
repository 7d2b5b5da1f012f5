#!/usr/bin/env python3
"""Low-cardinality hash-path probe: time-bucketed GROUP BY (month / day / hour) over the C5t table, with and without a filter."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viyadb_amd import capi, executor, synth  # noqa: E402


def main():
    nseg = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    executor.init(0)
    w = synth.c5t()
    t = synth.create_device_table(w, nseg)
    rows = nseg * w.segment_rows
    roll = w.plan.groups[0].rollup
    for label, groups, flt in (
            ("month", [executor.GroupSpec(0, granularity=capi.T_MONTH, rollup=roll)], w.plan.filter),
            ("day", [executor.GroupSpec(0, granularity=capi.T_DAY, rollup=roll)], w.plan.filter),
            ("hour", [executor.GroupSpec(0, granularity=capi.T_HOUR, rollup=roll)], w.plan.filter),
            ("month_nofilter", [executor.GroupSpec(0, granularity=capi.T_MONTH, rollup=roll)], []),
            ("hour_u100", [executor.GroupSpec(0, granularity=capi.T_HOUR, rollup=roll), executor.GroupSpec(1)],
             [("rel", 1, capi.OP_LT, 100)]),
            ("C5t", w.plan.groups, w.plan.filter)):
        plan = executor.AggPlan(filter=flt, groups=groups, metrics=[3])
        ms = []
        for _ in range(4):
            r = t.query_agg(plan)
            ms.append(r.scan_kernel_ms)
        k = sorted(ms)[len(ms) // 2]
        print(json.dumps({"case": label, "path": r.path, "fast": r.fast, "groups": r.ngroups, "passed": r.passed_recs, "kernel_ms": round(k, 3),
                          "rows_per_s_G": round(rows / k / 1e6, 1), "retries": r.retries}), flush=True)
    t.close()


if __name__ == "__main__":
    main()

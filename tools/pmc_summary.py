"""Summarise rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE collected SEPARATELY, as MI355X_MICROARCH.md prescribes)
into profiles/<round>/<workload>_1gpu_pmc_hbm.json, the file bench.py reads `roofline.traffic` from.

usage: python tools/pmc_summary.py <fetch_dir> <write_dir> <out.json> --rows N --bref BYTES [--kernel SUBSTR]"""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict


def collect(d, counter):
    """rocprofv3 writes either CSV (--output-format csv) or a rocpd SQLite database (the default of ROCm 7.x)."""
    per = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                per[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    for f in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
        import sqlite3
        for name, value in sqlite3.connect(f).execute(
                "select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
            per[name].append(float(value))
    # median over the dispatches: a first attempt that overflowed its table (and was re-planned) must not set the figure
    return {k: {"dispatches": len(v), "mean_KiB": sorted(v)[len(v) // 2]} for k, v in per.items() if k.startswith("void scan") or "kernel" in k or "viya_jit" in k}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("out")
    ap.add_argument("--rows", type=int, required=True)
    ap.add_argument("--bref", type=float, required=True)
    ap.add_argument("--kernel", default="scan_agg_fast_kernel<2", help="substring of the kernel symbol; 'a + b' sums the kernels of one query (e.g. bench.py's roofline.kernel)")
    ap.add_argument("--command", default="")
    ap.add_argument("--head", default="", help="git revision the pass was taken at (bench.py reports it as traffic_head)")
    ap.add_argument("--packed", type=int, default=0, help="1: the query gathered its payload from a projection (bench.py only uses a pass taken the same way)")
    ap.add_argument("--narrow", type=int, default=0, help="1: predicate columns were streamed from narrow copies")
    ap.add_argument("--sources", default="", help="bench.kernel_sources_hash() of the tree the pass was taken with (bench.py refuses a summary whose stamp differs)")
    a = ap.parse_args()
    fetch, write = collect(a.fetch_dir, "FETCH_SIZE"), collect(a.write_dir, "WRITE_SIZE")
    names = []
    for part in a.kernel.split(" + "):
        names.append(max((k for k in fetch if part.strip() in k), key=lambda k: fetch[k]["mean_KiB"]))
    kname = " + ".join(names)
    fb = sum(fetch[k]["mean_KiB"] for k in names) * 1024.0
    wb = sum(write.get(k, {"mean_KiB": 0.0})["mean_KiB"] for k in names) * 1024.0
    out = {
        "command": a.command or "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --steps 3 --warmup 1 --no-cpu",
        "kernel": kname, "head": a.head, "sources": a.sources, "packed": bool(a.packed), "narrow": bool(a.narrow),
        "raw": {"FETCH_SIZE": fetch, "WRITE_SIZE": write},
        "fetch_bytes_raw": fb, "fetch_bytes_corrected_x2": fb * 2, "write_bytes_raw": wb,
        "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming "
                      "reads -> doubled (the narrow gathers fetch whole 128 B lines too: the sum matches the line-touch model of the "
                      "query); WRITE_SIZE uncalibrated, reported raw",
        "rows": a.rows, "B_ref": a.bref, "B_meas_per_launch": fb * 2,
    }
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("kernel", "fetch_bytes_corrected_x2", "write_bytes_raw")}))


def kernel_stats_csv(db, out):
    """`rocprofv3 --kernel-trace --stats` summary (top_kernels view of the rocpd database) as the usual CSV."""
    import sqlite3
    c = sqlite3.connect(db)
    with open(out, "w") as f:
        f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
        for name, calls, total, avg, pct in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
            mn, mx = c.execute("select min(duration), max(duration) from kernels where name = ?", (name,)).fetchone()
            f.write('"%s",%d,%d,%.3f,%.2f,%d,%d\n' % (name, calls, round(total * 1000), avg * 1000, pct, mn, mx))


if __name__ == "__main__":
    import sys
    if len(sys.argv) == 4 and sys.argv[1] == "--kernel-stats":
        kernel_stats_csv(sys.argv[2], sys.argv[3])
    else:
        main()

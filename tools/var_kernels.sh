#!/bin/bash
# Per-kernel averages of the C3 query in N separate processes (rocprofv3 kernel trace): which kernel carries the process-to-process spread?
N=${1:-4}; REPO=$PWD; OUT=gpurun_out/var_kernels; mkdir -p $OUT
for i in $(seq $N); do
  rm -rf $OUT/kt
  (cd /tmp && TMPDIR=/tmp timeout 240 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/kt -o v -- python $REPO/bench.py --no-cpu --no-check --steps 10 > $REPO/$OUT/kt.log 2>&1)
  DB=$(find $OUT/kt -name "*_results.db" | head -1)
  [ -n "$DB" ] || { echo "run $i: no db"; continue; }
  timeout 120 python tools/pmc_summary.py --kernel-stats "$DB" $OUT/k$i.csv > /dev/null
  python - $OUT/k$i.csv $i <<'PY'
import csv, sys
rows = {r["Name"]: r for r in csv.DictReader(open(sys.argv[1]))}
def avg(sub):
    for k, r in rows.items():
        if sub in k: return float(r["AverageNs"]) / 1e6
    return 0.0
print("run", sys.argv[2], "phase1 %.3f" % avg("scan_agg_fast_kernel<4"), "phase2 %.3f" % avg("part_agg_kernel"), "merge %.3f" % avg("dense_merge"), "emit %.3f" % avg("emit_groups"))
PY
done
rm -rf $OUT/kt

#!/bin/bash
REPO=$PWD; OUT=gpurun_out/phase2_scale; mkdir -p $OUT
for TH in 1 50 447 1000; do
  rm -rf $OUT/kt
  (cd /tmp && TMPDIR=/tmp timeout 240 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/kt -o v -- python $REPO/tools/phase2_probe.py $TH > $REPO/$OUT/kt.log 2>&1)
  grep threshold $OUT/kt.log
  DB=$(find $OUT/kt -name "*_results.db" | head -1)
  [ -n "$DB" ] || { echo "no db"; continue; }
  timeout 120 python tools/pmc_summary.py --kernel-stats "$DB" $OUT/k.csv > /dev/null
  grep -E "part_agg|scan_agg_fast_kernel<4|dense_merge" $OUT/k.csv | cut -d, -f1,2,4 
done
rm -rf $OUT/kt

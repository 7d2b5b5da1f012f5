#!/bin/bash
# rocprofv3 per-kernel averages of an arbitrary probe script. usage: tools/phase_split.sh <script.py> [args...]
REPO=$PWD; OUT=gpurun_out/phase_split; mkdir -p $OUT; rm -rf $OUT/kt
(cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/kt -o v -- python $REPO/"$@" > $REPO/$OUT/kt.log 2>&1)
grep "^{" $OUT/kt.log | cut -c1-200
DB=$(find $OUT/kt -name "*_results.db" | head -1)
[ -n "$DB" ] || { echo "no db"; tail -3 $OUT/kt.log; exit 1; }
timeout 120 python tools/pmc_summary.py --kernel-stats "$DB" $OUT/k.csv > /dev/null
grep -E "scan_agg|part_|dense_merge|emit" $OUT/k.csv | cut -c1-200
rm -rf $OUT/kt

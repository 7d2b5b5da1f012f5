#!/bin/bash
REPO=$PWD; OUT=gpurun_out/phase2_ablate; mkdir -p $OUT
for V in "" ab16 ab32; do
  rm -rf $OUT/kt
  LIB=""; [ -n "$V" ] && LIB="VIYA_HIP_LIB=$REPO/viyadb_amd/build/variants/$V/libviya_hip.so"
  (cd /tmp && env TMPDIR=/tmp $LIB timeout 240 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/kt -o v -- python $REPO/tools/phase2_probe.py 447 > $REPO/$OUT/kt.log 2>&1)
  DB=$(find $OUT/kt -name "*_results.db" | head -1)
  [ -n "$DB" ] || { echo "$V: no db"; tail -3 $OUT/kt.log; continue; }
  timeout 120 python tools/pmc_summary.py --kernel-stats "$DB" $OUT/k.csv > /dev/null
  echo "variant [$V]"; grep -E "part_agg" $OUT/k.csv
done
rm -rf $OUT/kt

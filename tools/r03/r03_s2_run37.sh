#!/bin/bash
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for et in 0 256 4096; do
  D=gpurun_out/r03/kt_et$et; rm -rf $D
  VH_EXT_TUPLES=$et timeout 200 rocprofv3 --kernel-trace --stats -d $D -o c3 -- python bench.py --steps 8 --warmup 2 --no-cpu --no-check --no-reference-layout > $D.log 2>&1
  echo "== VH_EXT_TUPLES=$et"; python tools/pmc_summary.py --kernel-stats $(find $D -name "*_results.db" | head -1) $D.csv > /dev/null 2>&1; grep -E "viya_jit|part_agg" $D.csv | cut -d, -f1-4
done

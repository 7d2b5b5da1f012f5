#!/bin/bash
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for a in 0 1 2 3 8 11 4 12; do
  D=gpurun_out/r03/kt_a$a; rm -rf $D
  VH_HP_ABLATE=$a timeout 200 rocprofv3 --kernel-trace -d $D -o c5 -- python bench.py --workload C5 --segments 125 --steps 3 --warmup 2 --no-cpu --no-check --no-reference-layout > $D.log 2>&1
  echo "== VH_HP_ABLATE=$a"; python tools/last_query_kernels.py $D viya_jit | grep -E "hp_aggregate" | head -1
done

#!/bin/bash
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
run() { # tag, env...
  TAG=$1; shift
  D=gpurun_out/r03/kt_$TAG; rm -rf $D
  env "$@" timeout 120 rocprofv3 --kernel-trace -d $D -o c5 -- python bench.py --workload C5 --segments 125 --steps 3 --warmup 2 --no-cpu --no-check --no-reference-layout > $D.log 2>&1
  echo "== $TAG $@"; python tools/last_query_kernels.py $D viya_jit | grep -E "hp_aggregate" | head -6
}
for B in 2 4 8 16 32 64 128; do run b$B VH_HP_BPP=$B; done
run l9 VH_HP_BPP=16 VH_HP_LOAD_G=0.95 VH_HP_LOAD_S=0.95

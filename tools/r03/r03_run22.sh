#!/bin/bash
# C5: where the time of hp_aggregate_kernel and of the scan kernel goes (ablations; results wrong, timing only)
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
run() { # tag, env...
  TAG=$1; shift
  D=gpurun_out/r03/kt_$TAG; rm -rf $D
  env "$@" timeout 120 rocprofv3 --kernel-trace -d $D -o c5 -- python bench.py --workload C5 --segments 125 --steps 3 --warmup 2 --no-cpu --no-check --no-reference-layout > $D.log 2>&1
  echo "== $TAG $@"; python tools/last_query_kernels.py $D viya_jit | grep -E "viya_jit|hp_aggregate|hp_scatter" | head -6
}
run a0 VH_HP_ABLATE=0
run a1 VH_HP_ABLATE=1
run a2 VH_HP_ABLATE=2
run a3 VH_HP_ABLATE=3
run a4 VH_HP_ABLATE=4
run a12 VH_HP_ABLATE=12
run j8 VH_JIT_FLAGS=-DVJ_ABL=8
run j4 VH_JIT_FLAGS=-DVJ_ABL=4
run j12 VH_JIT_FLAGS=-DVJ_ABL=12

#!/bin/bash
# session 2, run 1: C5 scan ablations (timing only) + C3 launch-geometry sweep
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
run() { # tag, env...
  TAG=$1; shift
  D=gpurun_out/r03/kt_$TAG; rm -rf $D
  env "$@" VH_JIT_VERBOSE=1 timeout 120 rocprofv3 --kernel-trace -d $D -o c5 -- python bench.py --workload C5 --segments 125 --steps 3 --warmup 2 --no-cpu --no-check --no-reference-layout > $D.log 2>&1
  echo "== $TAG $@"; grep "vh jit" $D.log | head -2; python tools/last_query_kernels.py $D viya_jit | grep -E "viya_jit|hp_aggregate|hp_scatter|emit|copy" | head -9
}
run a0 VH_HP_ABLATE=0
run j8 VH_JIT_FLAGS=-DVJ_ABL=8
run j4 VH_JIT_FLAGS=-DVJ_ABL=4
run j12 VH_JIT_FLAGS=-DVJ_ABL=12
run b6 VH_BLOCKS_PER_CU=6
run b3 VH_BLOCKS_PER_CU=3
c3() { TAG=$1; shift
  env "$@" python bench.py --steps 10 --warmup 3 --no-cpu --no-check --no-reference-layout > gpurun_out/r03/c3_$TAG.json 2> gpurun_out/r03/c3_$TAG.err
  echo "== c3 $TAG $@"; python - <<P
import json
d=json.loads(open('gpurun_out/r03/c3_$TAG.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel'])
P
}
c3 base VH_JIT_VERBOSE=1; grep "vh jit" gpurun_out/r03/c3_base.err | head -3
for b in 4 5 6 7 8; do c3 bpc$b VH_BLOCKS_PER_CU=$b; done
c3 unit16k VH_UNIT_ROWS=16384
c3 unit64k VH_UNIT_ROWS=65536
c3 ntg VH_JIT_FLAGS=-DVJ_NT_GATHER=1

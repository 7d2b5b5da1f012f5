#!/bin/bash
# C3 with the tuple pool placed by measurement: where the scan's time goes (ablations: timing only), blocks per CU again, per-query fixed cost
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
c3() { TAG=$1; shift
  env "$@" VH_TRACE_ALLOC=1 VH_TIMES=1 python bench.py --steps 10 --warmup 3 --no-cpu --no-check --no-reference-layout > gpurun_out/r03/c3_$TAG.json 2> gpurun_out/r03/c3_$TAG.err
  python - <<P
import json
d=json.loads(open('gpurun_out/r03/c3_$TAG.json').read().strip().splitlines()[-1])
print("c3 $TAG $@", round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), d['config']['table_path'])
P
  grep "scratch trial" gpurun_out/r03/c3_$TAG.err | tail -1; grep "vh times" gpurun_out/r03/c3_$TAG.err | tail -1
}
c3 base
c3 abl1 VH_JIT_ABLATE=1
c3 abl2 VH_JIT_ABLATE=2
c3 abl8 VH_JIT_FLAGS=-DVJ_ABL=8
c3 nophase2 VH_ABLATE_NO_PHASE2=1
for b in 2 3 4 5; do c3 bpc$b VH_BLOCKS_PER_CU=$b; done
c3 base2
D=gpurun_out/r03/kt_c3; rm -rf $D
timeout 200 rocprofv3 --kernel-trace -d $D -o c3 -- python bench.py --steps 5 --warmup 2 --no-cpu --no-check --no-reference-layout > $D.log 2>&1
python tools/last_query_kernels.py $D viya_jit | head -12

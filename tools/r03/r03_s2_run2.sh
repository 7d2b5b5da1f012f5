#!/bin/bash
# session 2, run 2: blocks per CU of the compiled scan kernels, interleaved repeats on one box (C3 whole table, C3 an eighth, C5)
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
c3() { TAG=$1; SEG=$2; shift 2
  env "$@" python bench.py --segments $SEG --steps 10 --warmup 3 --no-cpu --no-check --no-reference-layout > gpurun_out/r03/c3_$TAG.json 2> gpurun_out/r03/c3_$TAG.err
  python - <<P
import json
d=json.loads(open('gpurun_out/r03/c3_$TAG.json').read().strip().splitlines()[-1])
print("c3 $TAG seg=$SEG $@", round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), d['config']['table_path'])
P
}
for rep in 1 2; do for b in 0 3 4 5 6; do c3 bpc${b}_r$rep 1000 VH_BLOCKS_PER_CU=$b; done; done
for rep in 1 2; do for b in 0 2 3 4 6; do c3 s125_bpc${b}_r$rep 125 VH_BLOCKS_PER_CU=$b; done; done
c5() { TAG=$1; shift
  env "$@" VH_TIMES=1 python bench.py --workload C5 --segments 125 --steps 5 --warmup 2 --no-cpu --no-check --no-reference-layout > gpurun_out/r03/c5_$TAG.json 2> gpurun_out/r03/c5_$TAG.err
  python - <<P
import json
d=json.loads(open('gpurun_out/r03/c5_$TAG.json').read().strip().splitlines()[-1])
print("c5 $TAG $@", round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))
P
  grep "vh times" gpurun_out/r03/c5_$TAG.err | tail -1
}
for rep in 1 2; do for b in 0 2 3 4 6; do c5 bpc${b}_r$rep VH_BLOCKS_PER_CU=$b; done; done

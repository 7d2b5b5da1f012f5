#!/bin/bash
# C5 phase 1: what is it made of? (ablations: 3 = scan + compaction only, 2 = + gathers, default = everything)
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for AB in 0 3 2; do
  D=gpurun_out/r03/kt_c5_ab$AB; rm -rf $D
  VH_JIT_ABLATE=$AB rocprofv3 --kernel-trace --stats -d $D -o c5 -- python bench.py --workload C5 --segments 125 --steps 3 --warmup 3 --no-cpu --no-check --no-reference-layout > $D.log 2>&1
  echo "== VH_JIT_ABLATE=$AB"; python tools/last_query_kernels.py $D viya_jit | head -14
done
D=gpurun_out/r03/pmc_c5; rm -rf $D
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --kernel-trace -d $D -o c5 -- python bench.py --workload C5 --segments 125 --steps 2 --warmup 1 --no-cpu --no-check --no-reference-layout > $D.log 2>&1
for K in viya_jit hp_scatter hp_aggregate; do python tools/pmc_kernel.py $D $K; done
D=gpurun_out/r03/pmc_c5b; rm -rf $D
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace -d $D -o c5 -- python bench.py --workload C5 --segments 125 --steps 2 --warmup 1 --no-cpu --no-check --no-reference-layout > $D.log 2>&1
for K in viya_jit hp_scatter hp_aggregate; do python tools/pmc_kernel.py $D $K; done

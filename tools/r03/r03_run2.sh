#!/bin/bash
mkdir -p gpurun_out/r03
printf '%s\n' - VH_BLOCKS_PER_CU=4 VH_BLOCKS_PER_CU=6 VH_BLOCKS_PER_CU=8 VH_JIT_ABLATE=3 VH_JIT_ABLATE=2 "VH_JIT_FLAGS=-DVH_ABLATE=4" VH_PLACEMENT_TRIALS=1 VH_JIT=off - | bash tools/r03_exp.sh c3a --steps 20
export TMPDIR=/tmp; REPO=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/r03/kt_c3 -o c3 -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu --no-check > $REPO/gpurun_out/r03/kt_c3.log 2>&1)
python tools/pmc_summary.py --kernel-stats $(find gpurun_out/r03/kt_c3 -name "*_results.db" | head -1) gpurun_out/r03/c3_kernel_stats.csv; head -8 gpurun_out/r03/c3_kernel_stats.csv | cut -c1-150
for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  D=gpurun_out/r03/pmc_d; rm -rf $D
  (cd /tmp && timeout 240 rocprofv3 --pmc $SET -d $REPO/$D -o p -- python $REPO/bench.py --no-cpu --no-check --steps 3 --warmup 1 > $REPO/$D.log 2>&1)
  timeout 60 python tools/pmc_kernel.py $D "viya_jit" | grep -v "^void"
done
rm -rf gpurun_out/r03/pmc_d gpurun_out/r03/kt_c3

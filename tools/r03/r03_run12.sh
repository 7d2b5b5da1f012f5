#!/bin/bash
mkdir -p gpurun_out/r03
( time timeout 1200 python -m pytest tests/test_gpu_jit.py tests/test_gpu_typed.py -q -m gpu -x -k "hashed" ) > gpurun_out/r03/hp_tests.log 2>&1
grep -E "passed|failed" gpurun_out/r03/hp_tests.log | tail -2
export TMPDIR=/tmp; REPO=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/r03/kt_c5 -o c5 -- python $REPO/bench.py --workload C5 --segments 125 --steps 5 --warmup 3 --no-cpu --no-check > $REPO/gpurun_out/r03/kt_c5.log 2>&1)
python tools/pmc_summary.py --kernel-stats $(find gpurun_out/r03/kt_c5 -name "*_results.db" | head -1) gpurun_out/r03/c5_kernel_stats.csv; head -12 gpurun_out/r03/c5_kernel_stats.csv | cut -c1-150
rm -rf gpurun_out/r03/kt_c5
printf '%s\n' "-" | bash tools/r03_exp.sh c5th --steps 5 --warmup 3 --workload C5t --segments 125

#!/bin/bash
mkdir -p gpurun_out/r03
printf '%s\n' "VH_TIMES=1" "VH_TIMES=1 VIYA_X=1" | bash tools/r03_exp.sh c5h --steps 5 --warmup 3 --workload C5 --segments 125
grep "vh times" gpurun_out/r03/c5h/1.err | tail -3
printf '%s\n' "VH_TIMES=1" | bash tools/r03_exp.sh c5th --steps 5 --warmup 3 --workload C5t --segments 125
grep "vh times" gpurun_out/r03/c5th/1.err | tail -3
timeout 900 python bench.py --workload C5 --segments 125 --steps 5 --warmup 3 --no-cpu 2> gpurun_out/r03/c5_full.err | cut -c1-900
tail -3 gpurun_out/r03/c5_full.err
export TMPDIR=/tmp; REPO=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/r03/kt_c5 -o c5 -- python $REPO/bench.py --workload C5 --segments 125 --steps 5 --warmup 3 --no-cpu --no-check > $REPO/gpurun_out/r03/kt_c5.log 2>&1)
python tools/pmc_summary.py --kernel-stats $(find gpurun_out/r03/kt_c5 -name "*_results.db" | head -1) gpurun_out/r03/c5_kernel_stats.csv; head -12 gpurun_out/r03/c5_kernel_stats.csv | cut -c1-150
rm -rf gpurun_out/r03/kt_c5

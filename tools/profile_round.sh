#!/bin/bash
# One gpurun call that refreshes profiles/<round>/ for the bench workload (C3, 1 GPU):
#   bench line, rocprofv3 kernel-trace stats of the same command, PMC passes (FETCH_SIZE / WRITE_SIZE separately).
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh r01
R=${1:-r01}
OUT=gpurun_out/$R
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench_c3_1gpu.json 2> $OUT/bench_c3.err
tail -c 600 $OUT/bench_c3_1gpu.json
REPO=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -d $REPO/$OUT/kt -o c3 -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu > $REPO/$OUT/kt.log 2>&1)
(cd /tmp && rocprofv3 --pmc FETCH_SIZE -d $REPO/$OUT/pmc_fetch -o c3 -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu > $REPO/$OUT/pmc_fetch.log 2>&1)
(cd /tmp && rocprofv3 --pmc WRITE_SIZE -d $REPO/$OUT/pmc_write -o c3 -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu > $REPO/$OUT/pmc_write.log 2>&1)
python tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/c3_1gpu_pmc_hbm.json --rows 1000000000 --bref 32e9
python tools/pmc_summary.py --kernel-stats $(find $OUT/kt -name "*_results.db" | head -1) $OUT/c3_1gpu_kernel_stats.csv; head -4 $OUT/c3_1gpu_kernel_stats.csv
python bench.py --workload C2 --no-cpu > $OUT/bench_c2_1gpu.json 2>> $OUT/bench_c3.err; tail -c 400 $OUT/bench_c2_1gpu.json
(cd /tmp && rocprofv3 --kernel-trace --stats -d $REPO/$OUT/kt_c2 -o c2 -- python $REPO/bench.py --workload C2 --steps 20 --warmup 2 --no-cpu > $REPO/$OUT/kt_c2.log 2>&1)
python tools/pmc_summary.py --kernel-stats $(find $OUT/kt_c2 -name "*_results.db" | head -1) $OUT/c2_1gpu_kernel_stats.csv; head -3 $OUT/c2_1gpu_kernel_stats.csv
(cd /tmp && rocprofv3 --pmc FETCH_SIZE -d $REPO/$OUT/pmc_fetch_c2 -o c2 -- python $REPO/bench.py --workload C2 --steps 3 --warmup 1 --no-cpu > $REPO/$OUT/pmc_fetch_c2.log 2>&1)
(cd /tmp && rocprofv3 --pmc WRITE_SIZE -d $REPO/$OUT/pmc_write_c2 -o c2 -- python $REPO/bench.py --workload C2 --steps 3 --warmup 1 --no-cpu > $REPO/$OUT/pmc_write_c2.log 2>&1)
python tools/pmc_summary.py $OUT/pmc_fetch_c2 $OUT/pmc_write_c2 $OUT/c2_1gpu_pmc_hbm.json --rows 100000000 --bref 2e9 --kernel scan_agg_lanes_kernel --command 'rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --workload C2 --steps 3 --warmup 1 --no-cpu'

mkdir -p gpurun_out/split
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( timeout 900 python -m pytest tests/test_gpu_typed.py -q -x -k "two_level" ) > gpurun_out/split/typed.log 2>&1; tail -5 gpurun_out/split/typed.log
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_jit.py -q -x ) > gpurun_out/split/parity.log 2>&1; tail -3 gpurun_out/split/parity.log
REPO=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/split/kt -o p2 -- python $REPO/tools/part2_probe.py 1000 1000,120 > $REPO/gpurun_out/split/part2.log 2>&1)
python tools/pmc_summary.py --kernel-stats $(find gpurun_out/split/kt -name "*_results.db" | head -1) gpurun_out/split/part2.csv; grep "split\|pagg" gpurun_out/split/part2.csv | cut -c1-150
grep "^{" gpurun_out/split/part2.log | grep -v direct | cut -c1-330
rm -rf gpurun_out/split/kt

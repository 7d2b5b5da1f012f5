mkdir -p gpurun_out/fan1
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( timeout 900 python -m pytest tests/test_gpu_hpart.py -q -x ) > gpurun_out/fan1/hpart.log 2>&1; tail -5 gpurun_out/fan1/hpart.log
( timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k c5 ) > gpurun_out/fan1/full.log 2>&1; tail -5 gpurun_out/fan1/full.log
Q="--no-cpu --no-check --no-reference-layout --no-cpu-parallel"
REPO=$PWD
for V in fan stream; do
  if [ $V = stream ]; then export VH_TEST_HOOKS=1 VH_NO_HP_FAN=1; fi
  python bench.py --workload C5 --segments 125 --steps 5 --warmup 1 $Q > gpurun_out/fan1/bench_$V.json 2> gpurun_out/fan1/bench_$V.err
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/fan1/kt_$V -o c5 -- python $REPO/bench.py --workload C5 --segments 125 --steps 10 --warmup 2 $Q > $REPO/gpurun_out/fan1/kt_$V.log 2>&1)
  python tools/pmc_summary.py --kernel-stats $(find gpurun_out/fan1/kt_$V -name "*_results.db" | head -1) gpurun_out/fan1/c5_${V}_kernel_stats.csv; head -8 gpurun_out/fan1/c5_${V}_kernel_stats.csv | cut -c1-200
  rm -rf gpurun_out/fan1/kt_$V
done
python -c "
import json
for v in ('fan','stream'):
    d=json.load(open('gpurun_out/fan1/bench_%s.json'%v)); print(v, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel'][:200])"

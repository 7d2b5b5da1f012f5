mkdir -p gpurun_out/fan2
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( timeout 900 python -m pytest tests/test_gpu_hpart.py -q -x ) > gpurun_out/fan2/hpart.log 2>&1; tail -5 gpurun_out/fan2/hpart.log
( timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k c5 ) > gpurun_out/fan2/full.log 2>&1; tail -5 gpurun_out/fan2/full.log
Q="--no-cpu --no-check --no-reference-layout --no-cpu-parallel"
REPO=$PWD
export VH_TEST_HOOKS=1
for V in ring ring2 ring4; do
  unset VH_TEST_HP_RING_NB
  if [ $V = ring2 ]; then export VH_TEST_HP_RING_NB=2; fi
  if [ $V = ring4 ]; then export VH_TEST_HP_RING_NB=4; fi
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/fan2/kt_$V -o c5 -- python $REPO/bench.py --workload C5 --segments 125 --steps 10 --warmup 2 $Q > $REPO/gpurun_out/fan2/kt_$V.log 2>&1)
  python tools/pmc_summary.py --kernel-stats $(find gpurun_out/fan2/kt_$V -name "*_results.db" | head -1) gpurun_out/fan2/c5_${V}_kernel_stats.csv; head -8 gpurun_out/fan2/c5_${V}_kernel_stats.csv | cut -c1-200
  rm -rf gpurun_out/fan2/kt_$V
done
unset VH_TEST_HP_RING_NB
python bench.py --workload C5 --segments 125 --steps 5 --warmup 1 $Q > gpurun_out/fan2/bench_ring.json 2> gpurun_out/fan2/bench_ring.err
python bench.py --workload C5t --segments 125 --steps 5 --warmup 1 $Q > gpurun_out/fan2/bench_c5t.json 2> gpurun_out/fan2/bench_c5t.err
python -c "
import json
for v in ('ring','c5t'):
    d=json.load(open('gpurun_out/fan2/bench_%s.json'%v)); print(v, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel'][:200])"

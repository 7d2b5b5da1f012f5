mkdir -p gpurun_out/fan3
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( timeout 900 python -m pytest tests/test_gpu_hpart.py -q -x ) > gpurun_out/fan3/hpart.log 2>&1; tail -3 gpurun_out/fan3/hpart.log
( timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k c5 ) > gpurun_out/fan3/full.log 2>&1; tail -3 gpurun_out/fan3/full.log
Q="--no-cpu --no-check --no-reference-layout --no-cpu-parallel"
REPO=$PWD
for V in uni old uni35 old35 uni50; do
  unset VH_JIT_FLAGS VH_HP_LOAD_G VH_HP_LOAD_S
  case $V in old*) export VH_JIT_FLAGS=-DHP_PROBE_UNIFORM=0;; esac
  case $V in *35) export VH_HP_LOAD_G=0.35 VH_HP_LOAD_S=0.35;; esac
  case $V in *50) export VH_HP_LOAD_G=0.5 VH_HP_LOAD_S=0.5;; esac
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/fan3/kt_$V -o c5 -- python $REPO/bench.py --workload C5 --segments 125 --steps 10 --warmup 2 $Q > $REPO/gpurun_out/fan3/kt_$V.log 2>&1)
  python tools/pmc_summary.py --kernel-stats $(find gpurun_out/fan3/kt_$V -name "*_results.db" | head -1) gpurun_out/fan3/c5_${V}_kernel_stats.csv; echo $V; grep "hpagg\|viya_jit_scan_[0-9a-f]*\"\|ring" gpurun_out/fan3/c5_${V}_kernel_stats.csv | cut -c1-120
  rm -rf gpurun_out/fan3/kt_$V
done

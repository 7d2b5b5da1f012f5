#!/bin/bash
# One gpurun call that refreshes profiles/<round>/ for the bench workloads on 1 GPU: the bench line, rocprofv3 kernel-trace stats of
# the same command, PMC passes (FETCH_SIZE / WRITE_SIZE in SEPARATE runs, as MI355X_MICROARCH.md prescribes) summed over the kernels
# the bench line names (roofline.kernel comes from the library: vh_result_kernel).
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh r02 <git head>
R=${1:-r03}
HEAD=${2:-unknown}
OUT=gpurun_out/$R
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
SRC=$(python -c "import bench; print(bench.kernel_sources_hash())")
one() {   # name (workload[_variant]), rows, bref, bench args...
  local W=$1 ROWS=$2 BREF=$3; shift 3
  python bench.py "$@" ${FIRST_EXTRA---no-reference-layout} > $OUT/bench_${W}_1gpu.json 2> $OUT/bench_${W}.err
  tail -c 300 $OUT/bench_${W}_1gpu.json; echo
  local K=$(python -c "import json; print(json.load(open('$OUT/bench_${W}_1gpu.json'))['roofline']['kernel'])")
  local PK=$(python -c "import json; print(int(json.load(open('$OUT/bench_${W}_1gpu.json'))['config']['payload_projection']))")
  local NR=$(python -c "import json; print(int(json.load(open('$OUT/bench_${W}_1gpu.json'))['config'].get('narrow_predicates', False)))")
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $REPO/$OUT/kt_$W -o $W -- python $REPO/bench.py "$@" --steps 10 --warmup 2 --no-cpu --no-check --no-reference-layout > $REPO/$OUT/kt_$W.log 2>&1)
  python tools/pmc_summary.py --kernel-stats $(find $OUT/kt_$W -name "*_results.db" | head -1) $OUT/${W}_1gpu_kernel_stats.csv; head -4 $OUT/${W}_1gpu_kernel_stats.csv | cut -c1-160
  for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && rocprofv3 --pmc $C -d $REPO/$OUT/pmc_${C}_$W -o $W -- python $REPO/bench.py "$@" --steps 3 --warmup 1 --no-cpu --no-check --no-reference-layout > $REPO/$OUT/pmc_${C}_$W.log 2>&1)
  done
  local J=$OUT/${W}_1gpu_pmc_hbm.json
  case $W in *_*) J=$OUT/${W%%_*}_1gpu_pmc_hbm_${W#*_}.json;; esac
  python tools/pmc_summary.py $OUT/pmc_FETCH_SIZE_$W $OUT/pmc_WRITE_SIZE_$W $J --rows $ROWS --bref $BREF --kernel "$K" --head $HEAD --sources $SRC --packed $PK --narrow $NR \
    --command "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py $* --steps 3 --warmup 1 --no-cpu --no-check"
  case $W in c3|c5)      # instruction mix of the kernels the line names (its own pass: SQ counters)
    (cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM -d $REPO/$OUT/pmc_insts_$W -o $W -- python $REPO/bench.py "$@" --steps 3 --warmup 1 --no-cpu --no-check --no-reference-layout > $REPO/$OUT/pmc_insts_$W.log 2>&1)
    { echo "# rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM -- python bench.py $* --steps 3 --warmup 1 --no-cpu --no-check (head $HEAD, sources $SRC): wave-level instructions per launch"
      for KK in viya_jit part_agg hp_scatter scan_agg; do python tools/pmc_kernel.py $OUT/pmc_insts_$W $KK; done; } > $OUT/${W}_1gpu_pmc_insts.txt;;
  esac
}
one c3_arena 1000000000 32e9 --no-pack --no-cpu              # the reference layout: column arenas only (bench.py's reference_layout leg reads this pass)
mkdir -p profiles/$R; cp $OUT/c3_1gpu_pmc_hbm_arena.json profiles/$R/ 2>/dev/null    # (the headline run below looks for it under profiles/)
FIRST_EXTRA="" one c3 1000000000 32e9
one c3_direct 1000000000 32e9 --flags 16 --no-cpu             # the same query forced onto direct atomics (what a slower box or a smaller shard runs)
one c2 100000000 2e9 --workload C2 --no-cpu
one c5 125000000 3.5e9 --workload C5 --segments 125 --no-cpu --steps 5 --warmup 1
one c5t 125000000 1.5e9 --workload C5t --segments 125 --no-cpu --steps 5 --warmup 1
bash tools/fetch_calib.sh $OUT/fetch_calibration.json > $OUT/fetch_calibration.log 2>&1
rm -rf $OUT/kt_* $OUT/pmc_FETCH_SIZE_* $OUT/pmc_WRITE_SIZE_* $OUT/pmc_insts_*
ls $OUT

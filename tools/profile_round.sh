#!/bin/bash
# One gpurun call that refreshes profiles/<round>/ for the bench workloads on 1 GPU. Per workload, in THIS order (the
# committed bench lines must carry their own traffic): a short run for the kernel names and layout flags; rocprofv3 kernel-trace stats of
# the same command; PMC passes (FETCH_SIZE / WRITE_SIZE in SEPARATE runs, as MI355X_MICROARCH.md prescribes) summed over the kernels the
# line names (roofline.kernel comes from the library: vh_result_kernel), written to profiles/<round>/ at once; THEN the bench line, which
# finds that summary (same kernel, same layout, same sources) and reports roofline.traffic / frac from it.
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh r04 <git head>
R=${1:-r06}
HEAD=${2:-unknown}
OUT=gpurun_out/$R
mkdir -p $OUT profiles/$R
export TMPDIR=/tmp
REPO=$PWD
SRC=$(python -c "import bench; print(bench.kernel_sources_hash())")
Q="--no-cpu --no-check --no-reference-layout --no-cpu-parallel --no-unprepared"
one() {   # name (workload[_variant]), rows, bref, bench args...
  local W=$1 ROWS=$2 BREF=$3; shift 3
  python bench.py "$@" $Q --steps 3 --warmup 2 > $OUT/pre_$W.json 2> $OUT/pre_$W.err
  local K=$(python -c "import json; print(json.load(open('$OUT/pre_$W.json'))['roofline']['kernel'])")
  local PK=$(python -c "import json; print(int(json.load(open('$OUT/pre_$W.json'))['config']['payload_projection']))")
  local NR=$(python -c "import json; print(int(json.load(open('$OUT/pre_$W.json'))['config'].get('narrow_predicates', False)))")
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $REPO/$OUT/kt_$W -o $W -- python $REPO/bench.py "$@" $Q --steps 10 --warmup 2 > $REPO/$OUT/kt_$W.log 2>&1)
  python tools/pmc_summary.py --kernel-stats $(find $OUT/kt_$W -name "*_results.db" | head -1) $OUT/${W}_1gpu_kernel_stats.csv; head -5 $OUT/${W}_1gpu_kernel_stats.csv | cut -c1-160
  for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && rocprofv3 --pmc $C -d $REPO/$OUT/pmc_${C}_$W -o $W -- python $REPO/bench.py "$@" $Q --steps 3 --warmup 1 > $REPO/$OUT/pmc_${C}_$W.log 2>&1)
  done
  local J=$OUT/${W}_1gpu_pmc_hbm.json
  case $W in *_*) J=$OUT/${W%%_*}_1gpu_pmc_hbm_${W#*_}.json;; esac
  python tools/pmc_summary.py $OUT/pmc_FETCH_SIZE_$W $OUT/pmc_WRITE_SIZE_$W $J --rows $ROWS --bref $BREF --kernel "$K" --head $HEAD --sources $SRC --packed $PK --narrow $NR \
    --command "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py $* $Q --steps 3 --warmup 1"
  cp $J profiles/$R/                          # (the bench line below looks for it under profiles/)
  case $W in c3|c5)      # instruction mix of the kernels the line names (its own pass: SQ counters)
    (cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM -d $REPO/$OUT/pmc_insts_$W -o $W -- python $REPO/bench.py "$@" $Q --steps 3 --warmup 1 > $REPO/$OUT/pmc_insts_$W.log 2>&1)
    { echo "# rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM -- python bench.py $* $Q --steps 3 --warmup 1 (head $HEAD, sources $SRC): wave-level instructions per launch"
      for KK in viya_jit part_agg hp_scatter hp_ring_scatter scan_agg; do python tools/pmc_kernel.py $OUT/pmc_insts_$W $KK; done; } > $OUT/${W}_1gpu_pmc_insts.txt;;
  esac
  # the line itself, LAST: it now carries the traffic of the pass above
  python bench.py "$@" ${FINAL_EXTRA---no-cpu --no-reference-layout --no-cpu-parallel --no-unprepared} > $OUT/bench_${W}_1gpu.json 2> $OUT/bench_${W}.err
  python -c "import json; d=json.load(open('$OUT/bench_${W}_1gpu.json')); r=d['roofline']; print('$W', round(d['ms_per_step'],3), 'ms/step, kernels', round(r['kernel_ms'],3), 'frac', round(r['frac'],3), 'traffic', r['traffic'])"
}
one c3_arena 1000000000 32e9 --no-pack              # the reference layout: column arenas only (bench.py's reference_layout leg reads this pass)
FINAL_EXTRA="" one c3 1000000000 32e9               # the headline line: parity gate, reference_layout leg, both CPU baselines
one c3_direct 1000000000 32e9 --flags 16            # the same query forced onto direct atomics (what a slower box or a smaller shard runs)
one c2 100000000 2e9 --workload C2
# (C5 / C5t: with the CPU twin's baseline on a 5-segment sample — VERDICT r05 #2 — and the parity gate)
FINAL_EXTRA="--no-reference-layout --no-cpu-parallel --no-unprepared" one c5 125000000 3.5e9 --workload C5 --segments 125 --steps 5 --warmup 1
FINAL_EXTRA="--no-reference-layout --no-cpu-parallel --no-unprepared" one c5t 125000000 1.5e9 --workload C5t --segments 125 --steps 5 --warmup 1
python bench.py --workload C1 --no-cpu-parallel --no-reference-layout > $OUT/bench_c1_1gpu.json 2> $OUT/bench_c1.err     # the plumbing case: a bench line only (launch-bound)
python tools/scale_proxy.py 1 2 4 8 2>/dev/null | grep '^{' > $OUT/scale_proxy.txt                                         # one rank's step of the N-GPU run, on one GPU
bash tools/fetch_calib.sh $OUT/fetch_calibration.json > $OUT/fetch_calibration.log 2>&1
# the mirror under ingest (batched sync, derived layouts by row range, the shim's clean pass) and the layout A/Bs of the round, each inside ONE process
g++ -std=c++17 -O2 tools/ingest_bench.cc -Iinclude -Lviyadb_amd -lviya_host -lviya_hip -Wl,-rpath,$PWD/viyadb_amd -o /tmp/ingest_bench && /tmp/ingest_bench > $OUT/ingest.json 2> $OUT/ingest.err
python tools/pred_ab.py 1000 7 2>/dev/null | grep '^{' > $OUT/pred_ab.txt
python tools/pred_ab.py 125 7 2>/dev/null | grep '^{' >> $OUT/pred_ab.txt
python tools/qpay_probe.py 2>/dev/null | grep '^{' > $OUT/qpay_probe.txt
python tools/part2_probe.py 1000 1000,120 2>/dev/null | grep '^{' > $OUT/part2_probe.txt      # 4 M groups (two partition levels) at 100 % / 12 % of 1 B rows
python tools/hisel_probe.py 1000 2>/dev/null | grep '^{' > $OUT/hisel_probe.txt              # 100 K groups from the arenas at 100 / 50 / 25 %
python tools/skew_probe.py 1000 125 15 2>/dev/null | grep '^{' > $OUT/skew.json              # Zipf-like keys, a table loaded in predicate order, a hot composite key: next to their uniform twins
VH_TIMES=1 python tools/host_share_probe.py C1,C2,C3 50 2> $OUT/host_share.err | grep '^{' > $OUT/host_share.json; grep 'vh plan steps' $OUT/host_share.err | awk 'NR%50==0' >> $OUT/host_share.json
# how stable the headline is from process to process: ten fresh processes as a caller that prepares its query shape (vh_table_prepare) and
# ten as one that does not (an ordinary first query: plain hipMalloc for the tuple pool)
{ for i in 1 2 3 4 5 6 7 8 9 10; do python bench.py $Q --steps 10 --warmup 3 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prepared', $i, round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"; done
  for i in 1 2 3 4 5 6 7 8 9 10; do python bench.py $Q --no-warm --steps 10 --warmup 3 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('unprepared', $i, round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"; done; } > $OUT/c3_ten_processes.txt 2>/dev/null
cat $OUT/c3_ten_processes.txt
rm -rf $OUT/kt_* $OUT/pmc_FETCH_SIZE_* $OUT/pmc_WRITE_SIZE_* $OUT/pmc_insts_* $OUT/pre_*
ls $OUT

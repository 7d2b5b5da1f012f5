#!/bin/bash
# Instruction counters of one kernel of the C3 query for several library variants. usage: tools/pmc_variant.sh <kernel substr> [variant ...] ("" = release build)
K="$1"; shift
REPO=$PWD; OUT=gpurun_out/pmc_variant; mkdir -p $OUT
for V in "$@"; do
  LIB=""; [ "$V" != "release" ] && LIB="VIYA_HIP_LIB=$REPO/viyadb_amd/build/variants/$V/libviya_hip.so"
  for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
    D=$OUT/d
    rm -rf $D
    (cd /tmp && env TMPDIR=/tmp $LIB timeout 240 rocprofv3 --pmc $SET -d $REPO/$D -o p -- python $REPO/bench.py $BENCH_ARGS --no-cpu --no-check --steps 3 --warmup 1 > $REPO/$D.log 2>&1)
    echo "== $V"; timeout 60 python tools/pmc_kernel.py $D "$K" | grep -v "^void"
  done
done
rm -rf $OUT/d

#!/bin/bash
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
kt() { TAG=$1; shift
  D=gpurun_out/r03/kt_$TAG; rm -rf $D
  env "$@" timeout 200 rocprofv3 --kernel-trace --stats -d $D -o c3 -- python bench.py --steps 8 --warmup 2 --no-cpu --no-check --no-reference-layout > $D.log 2>&1
  echo "== $TAG $@"; python tools/pmc_summary.py --kernel-stats $(find $D -name "*_results.db" | head -1) $D.csv > /dev/null 2>&1; grep -E "viya_jit|part_agg" $D.csv | cut -d, -f1-4
}
kt base
kt p2s6 VIYA_HIP_LIB=$PWD/viyadb_amd/build/variants/p2s6/libviya_hip.so
kt p2s8 VIYA_HIP_LIB=$PWD/viyadb_amd/build/variants/p2s8/libviya_hip.so
kt base_b
kt p2s8_b VIYA_HIP_LIB=$PWD/viyadb_amd/build/variants/p2s8/libviya_hip.so

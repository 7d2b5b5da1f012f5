#!/bin/bash
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for v in base agg256 agg1024 base agg256 agg1024; do
  L=""; [ $v != base ] && L="VIYA_HIP_LIB=$PWD/viyadb_amd/build/variants/$v/libviya_hip.so"
  D=gpurun_out/r03/kt_$v; rm -rf $D
  env $L timeout 200 rocprofv3 --kernel-trace -d $D -o c5 -- python bench.py --workload C5 --segments 125 --steps 3 --warmup 2 --no-cpu --no-check --no-reference-layout > $D.log 2>&1
  echo "== $v"; python tools/last_query_kernels.py $D viya_jit | grep -E "hp_aggregate" | head -2
done

#!/bin/bash
mkdir -p gpurun_out/r03
export TMPDIR=/tmp; REPO=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/r03/kt_c5 -o c5 -- python $REPO/bench.py --workload C5 --segments 125 --steps 3 --warmup 3 --no-cpu --no-check > $REPO/gpurun_out/r03/kt_c5.log 2>&1)
python tools/last_query_kernels.py gpurun_out/r03/kt_c5 viya_jit_scan
rm -rf gpurun_out/r03/kt_c5

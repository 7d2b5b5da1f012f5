#!/bin/bash
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( timeout 900 python -m pytest tests/test_gpu_hpart.py tests/test_gpu_fullsize.py -q -m gpu -x ) > gpurun_out/r03/hp_tests4.log 2>&1; tail -3 gpurun_out/r03/hp_tests4.log
for W in C5 C5t; do
D=gpurun_out/r03/kt_$W; rm -rf $D
timeout 200 rocprofv3 --kernel-trace -d $D -o c5 -- python bench.py --workload $W --segments 125 --steps 3 --warmup 2 --no-cpu --no-reference-layout > $D.log 2>&1
echo "== $W"; python tools/last_query_kernels.py $D viya_jit | grep -E "viya_jit|hp_" | head -8; grep -o '"parity_checked": [a-z]*' $D.log | head -1
done

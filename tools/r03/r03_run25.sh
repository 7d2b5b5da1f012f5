#!/bin/bash
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
D=gpurun_out/r03/kt_c3; rm -rf $D
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $D -o c3 -- python bench.py --steps 10 --warmup 3 --no-cpu --no-check --no-reference-layout > $D.log 2>&1
python tools/last_query_kernels.py $D viya_jit | head -12
printf '%s\n' - - - | bash tools/r03_exp.sh c3i --steps 20 --warmup 5 --no-reference-layout
( time timeout 2400 python -m pytest tests -q -m gpu -x ) > gpurun_out/r03/gpu_tests.log 2>&1
grep -E "passed|failed" gpurun_out/r03/gpu_tests.log | tail -3

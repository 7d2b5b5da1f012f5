#!/bin/bash
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( timeout 900 python -m pytest tests/test_gpu_jit.py tests/test_gpu_typed.py tests/test_gpu_fullsize.py -q -m gpu -x -k "hpart or hashed or c5 or bitset or distinct" ) > gpurun_out/r03/hp_tests.log 2>&1; tail -3 gpurun_out/r03/hp_tests.log
run() { # tag, env...
  TAG=$1; shift
  D=gpurun_out/r03/kt_$TAG; rm -rf $D
  env "$@" timeout 120 rocprofv3 --kernel-trace -d $D -o c5 -- python bench.py --workload C5 --segments 125 --steps 3 --warmup 2 --no-cpu --no-reference-layout > $D.log 2>&1
  echo "== $TAG $@"; python tools/last_query_kernels.py $D viya_jit | grep -E "viya_jit|hp_aggregate" | head -6; grep -o '"parity": [a-z"]*' $D.log | head -1
}
run n0 VH_HP_ABLATE=0
run n1 VH_HP_ABLATE=1

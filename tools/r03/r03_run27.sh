#!/bin/bash
mkdir -p gpurun_out/r03
( timeout 1200 python -m pytest tests/test_gpu_pack.py tests/test_gpu_jit.py -q -m gpu -x ) > gpurun_out/r03/pack_tests.log 2>&1; tail -15 gpurun_out/r03/pack_tests.log
printf '%s\n' - - VH_PACK_PLAIN=1 | bash tools/r03_exp.sh c3k --steps 20 --warmup 5 --no-reference-layout
grep -o '"derived_layout": {[^}]*}' gpurun_out/r03/c3k/1.json

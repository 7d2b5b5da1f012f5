#!/bin/bash
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_jit.py tests/test_gpu_pack.py tests/test_gpu_fullsize.py -q -m gpu -x ) > gpurun_out/r03/part_tests.log 2>&1; tail -3 gpurun_out/r03/part_tests.log
( timeout 600 python -m pytest tests/test_gpu_typed.py -q -m gpu -x -k "partition or two_level or part" ) > gpurun_out/r03/part_tests2.log 2>&1; tail -3 gpurun_out/r03/part_tests2.log
for pad in 8 0 8 0; do echo "== VH_EXT_PAD=$pad"; VH_EXT_PAD=$pad python tools/scratch_probe.py 16 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['context'], sorted(d['kernel_ms'])[2])
" | tr '\n' ' '; echo; done
for j in 1 0; do echo "== C5 VJ_BS_MERGE=$j"; VH_JIT_FLAGS=-DVJ_BS_MERGE=$j VH_TIMES=1 python bench.py --workload C5 --segments 125 --steps 5 --warmup 2 --no-cpu --no-check --no-reference-layout 2>&1 | grep "vh times" | tail -2; done

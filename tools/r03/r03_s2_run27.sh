#!/bin/bash
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( timeout 1500 python -m pytest tests/test_gpu_hpart.py tests/test_gpu_fullsize.py tests/test_gpu_distributed.py tests/test_gpu_cluster_merge.py -q -m gpu -x ) > gpurun_out/r03/direct_tests.log 2>&1; tail -5 gpurun_out/r03/direct_tests.log
( VH_FUZZ_SEEDS=24 timeout 900 python -m pytest tests/test_gpu_typed.py -q -m gpu -x -k random_plans ) > gpurun_out/r03/direct_fuzz.log 2>&1; tail -2 gpurun_out/r03/direct_fuzz.log
for W in C5 C5t; do for L in 0 1; do echo "== $W VH_HP_LIST=$L"; VH_HP_LIST=$L VH_TIMES=1 python bench.py --workload $W --segments 125 --steps 5 --warmup 2 --no-cpu --no-reference-layout > gpurun_out/r03/d_$W$L.json 2> gpurun_out/r03/d_$W$L.err; grep "vh times" gpurun_out/r03/d_$W$L.err | tail -1; python -c "
import json; d=json.loads(open('gpurun_out/r03/d_$W$L.json').read().strip().splitlines()[-1]); print(d['parity_checked'], round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"; done; done

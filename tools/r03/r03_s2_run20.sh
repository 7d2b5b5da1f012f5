#!/bin/bash
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_typed.py tests/test_gpu_jit.py tests/test_gpu_pack.py tests/test_gpu_narrow.py tests/test_gpu_fullsize.py tests/test_gpu_distributed.py tests/test_gpu_hpart.py -q -m gpu -x ) > gpurun_out/r03/part_tests3.log 2>&1; tail -3 gpurun_out/r03/part_tests3.log
( VH_POISON=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_jit.py -q -m gpu -x ) > gpurun_out/r03/poison_tests.log 2>&1; tail -2 gpurun_out/r03/poison_tests.log
python bench.py > gpurun_out/r03/bench_full.json 2> gpurun_out/r03/bench_full.err; python - <<P
import json
d=json.loads(open('gpurun_out/r03/bench_full.json').read().strip().splitlines()[-1])
print(round(d['value']/1e9,1), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), d['parity'])
P
tail -2 gpurun_out/r03/bench_full.err
echo "== overhead"; python tools/overhead.py 2>&1 | tail -4

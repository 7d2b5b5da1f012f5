#!/bin/bash
# first GPU pass of round 3: the per-query compiled kernels against the parity suites, then C3 with and without them
mkdir -p gpurun_out/r03
export VH_JIT_VERBOSE=1
( time timeout 1500 env VH_JIT=force python -m pytest tests/test_gpu_parity.py tests/test_gpu_pack.py tests/test_gpu_narrow.py tests/test_gpu_typed.py -q -m gpu ) > gpurun_out/r03/jit_parity.log 2>&1
tail -5 gpurun_out/r03/jit_parity.log
for i in 1 2; do
  timeout 600 python bench.py --no-cpu --steps 20 > gpurun_out/r03/bench_c3_jit_$i.json 2> gpurun_out/r03/bench_c3_jit_$i.err
  timeout 600 env VH_JIT=off python bench.py --no-cpu --steps 20 > gpurun_out/r03/bench_c3_nojit_$i.json 2> gpurun_out/r03/bench_c3_nojit_$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03/bench_c3_*.json')):
    try:
        d=json.load(open(f)); print(f, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel'])
    except Exception as e: print(f, 'ERR', e)
PY

#!/bin/bash
# One gpurun call: the whole -m gpu suite as the driver runs it, then again with the per-query compiled kernels switched off
# (VH_JIT=off; the suites that ask for a compiled kernel by flag are left out of that pass), then a short bench line.
mkdir -p gpurun_out/verify
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( time timeout 1500 python -m pytest tests -q -m gpu -x ) > gpurun_out/verify/gpu_default.log 2>&1; tail -4 gpurun_out/verify/gpu_default.log
( time VH_JIT=off timeout 1200 python -m pytest tests -q -m gpu --deselect tests/test_gpu_jit.py ) > gpurun_out/verify/gpu_jit_off.log 2>&1; tail -4 gpurun_out/verify/gpu_jit_off.log
python bench.py > gpurun_out/verify/bench_c3.json 2> gpurun_out/verify/bench_c3.err; tail -c 600 gpurun_out/verify/bench_c3.json

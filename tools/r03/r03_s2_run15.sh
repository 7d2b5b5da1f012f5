#!/bin/bash
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( timeout 900 python -m pytest tests/test_gpu_hpart.py -q -m gpu -x ) > gpurun_out/r03/hpart_tests.log 2>&1; tail -25 gpurun_out/r03/hpart_tests.log
echo "== scale proxy"; timeout 600 python tools/scale_proxy.py 1 2 4 8 > gpurun_out/r03/scale_proxy.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r03/scale_proxy.log | tail -12

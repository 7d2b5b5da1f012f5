#!/bin/bash
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( timeout 900 python -m pytest tests/test_gpu_hpart.py -q -m gpu -x ) > gpurun_out/r03/hpart_tests5.log 2>&1; tail -15 gpurun_out/r03/hpart_tests5.log
( timeout 1500 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_cluster_merge.py -q -m gpu -x ) > gpurun_out/r03/dist_tests.log 2>&1; tail -15 gpurun_out/r03/dist_tests.log

#!/bin/bash
mkdir -p gpurun_out/r03
( timeout 1200 python -m pytest tests/test_gpu_pack.py -q -m gpu ) > gpurun_out/r03/pack_tests.log 2>&1; tail -8 gpurun_out/r03/pack_tests.log | cut -c1-300

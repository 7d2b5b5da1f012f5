#!/bin/bash
mkdir -p gpurun_out/r03
timeout 600 python -m pytest tests/test_gpu_jit.py -q -m gpu -x -k random_plans > gpurun_out/r03/rnd.log 2>&1
tail -3 gpurun_out/r03/rnd.log | cut -c1-600

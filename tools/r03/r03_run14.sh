#!/bin/bash
mkdir -p gpurun_out/r03
( time timeout 1200 python -m pytest tests/test_gpu_jit.py -q -m gpu -x -k "hashed" ) > gpurun_out/r03/hp_tests.log 2>&1
grep -E "passed|failed|Error" gpurun_out/r03/hp_tests.log | tail -3
bash tools/r03_run13.sh

#!/bin/bash
mkdir -p gpurun_out/r03
( time timeout 1500 python -m pytest tests/test_gpu_jit.py -q -m gpu -x ) > gpurun_out/r03/jit_tests.log 2>&1
tail -5 gpurun_out/r03/jit_tests.log
printf '%s\n' - "VH_NO_STAGE=1" "VH_ABLATE_NO_PHASE2=1" "VH_ABLATE_NO_PHASE2=1 VH_NO_STAGE=1" "VH_BLOCKS_PER_CU=4" "VH_BLOCKS_PER_CU=6" - "VH_NO_STAGE=1" | bash tools/r03_exp.sh c3e --steps 20
python bench.py --no-cpu --steps 20 | cut -c1-600

#!/bin/bash
mkdir -p gpurun_out/r04
O=gpurun_out/r04/ab_probe.txt; : > $O
python tools/c5_probe.py > gpurun_out/r04/c5_probe_a.txt 2>&1; tail -4 gpurun_out/r04/c5_probe_a.txt
VH_TEST_EXT_CURSOR=1 python tools/c5_probe.py > gpurun_out/r04/c5_probe_b.txt 2>&1; tail -4 gpurun_out/r04/c5_probe_b.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hpart.py tests/test_gpu_typed.py tests/test_gpu_distributed.py -x -q 2>&1 | tail -3

#!/bin/bash
mkdir -p gpurun_out/r03
( time timeout 2400 python -m pytest tests -q -m gpu -x ) > gpurun_out/r03/gpu_tests.log 2>&1
grep -E "passed|failed" gpurun_out/r03/gpu_tests.log | tail -3
timeout 600 python tools/c3_sweep.py > gpurun_out/r03/c3_sweep.jsonl 2> gpurun_out/r03/c3_sweep.err; cat gpurun_out/r03/c3_sweep.jsonl | cut -c1-200

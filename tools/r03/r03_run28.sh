#!/bin/bash
mkdir -p gpurun_out/r03
( timeout 2400 python -m pytest tests -q -m gpu ) > gpurun_out/r03/gpu_tests.log 2>&1; tail -15 gpurun_out/r03/gpu_tests.log | cut -c1-300

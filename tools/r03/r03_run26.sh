#!/bin/bash
mkdir -p gpurun_out/r03
( time timeout 2400 python -m pytest tests -q -m gpu -x ) > gpurun_out/r03/gpu_tests.log 2>&1
grep -E "passed|failed" gpurun_out/r03/gpu_tests.log | tail -3
printf '%s\n' - - | bash tools/r03_exp.sh c3j --steps 20 --warmup 5 --no-reference-layout
printf '%s\n' - | bash tools/r03_exp.sh c5j --steps 5 --warmup 3 --workload C5 --segments 125 --no-reference-layout
printf '%s\n' - | bash tools/r03_exp.sh c2j --steps 20 --warmup 3 --workload C2 --no-reference-layout

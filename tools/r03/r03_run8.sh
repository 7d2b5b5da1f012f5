#!/bin/bash
mkdir -p gpurun_out/r03
( time timeout 2400 python -m pytest tests -q -m gpu -x ) > gpurun_out/r03/gpu_tests.log 2>&1
tail -5 gpurun_out/r03/gpu_tests.log
printf '%s\n' - - | bash tools/r03_exp.sh c3f --steps 20
printf '%s\n' - | bash tools/r03_exp.sh c2f --steps 20 --workload C2
printf '%s\n' - | bash tools/r03_exp.sh c5f --steps 5 --warmup 1 --workload C5 --segments 125
printf '%s\n' - | bash tools/r03_exp.sh c5tf --steps 5 --warmup 1 --workload C5t --segments 125

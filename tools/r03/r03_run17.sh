#!/bin/bash
mkdir -p gpurun_out/r03
( time timeout 2400 python -m pytest tests -q -m gpu -x ) > gpurun_out/r03/gpu_tests.log 2>&1
grep -E "passed|failed" gpurun_out/r03/gpu_tests.log | tail -3
printf '%s\n' - | bash tools/r03_exp.sh c3g --steps 20
printf '%s\n' - | bash tools/r03_exp.sh c5g --steps 5 --warmup 3 --workload C5 --segments 125
printf '%s\n' - | bash tools/r03_exp.sh c5tg --steps 5 --warmup 3 --workload C5t --segments 125
printf '%s\n' - | bash tools/r03_exp.sh c2g --steps 20 --workload C2
bash tools/fetch_calib.sh gpurun_out/r03/fetch_calibration.json

#!/bin/bash
mkdir -p gpurun_out/r03
echo "== placement trials, C3, 4 processes each"
printf '%s\n' "VH_PLACEMENT_TRIALS=1" "VH_PLACEMENT_TRIALS=3" "VH_PLACEMENT_TRIALS=1" "VH_PLACEMENT_TRIALS=3" "VH_PLACEMENT_TRIALS=1" "VH_PLACEMENT_TRIALS=3" "VH_PLACEMENT_TRIALS=1" "VH_PLACEMENT_TRIALS=3" | bash tools/r03_exp.sh c3h --steps 20 --warmup 5
echo "== an eighth / a quarter of C3 (what one rank of 8 / 4 scans): default, forced partitioned, forced direct"
printf '%s\n' - - | bash tools/r03_exp.sh c3s8 --steps 30 --warmup 5 --segments 125
printf '%s\n' - | bash tools/r03_exp.sh c3s8p --steps 30 --warmup 5 --segments 125 --flags 64
printf '%s\n' - | bash tools/r03_exp.sh c3s8d --steps 30 --warmup 5 --segments 125 --flags 16
printf '%s\n' - | bash tools/r03_exp.sh c3s4 --steps 30 --warmup 5 --segments 250
printf '%s\n' - | bash tools/r03_exp.sh c3s4p --steps 30 --warmup 5 --segments 250 --flags 64
printf '%s\n' - | bash tools/r03_exp.sh c3s4d --steps 30 --warmup 5 --segments 250 --flags 16
printf '%s\n' - | bash tools/r03_exp.sh c5t --steps 5 --warmup 3 --workload C5t --segments 125
bash tools/fetch_calib.sh gpurun_out/r03/fetch_calibration.json | head -8

#!/bin/bash
mkdir -p gpurun_out/r03
printf '%s\n' "VH_TIMES=1" "VH_TIMES=1 VH_JIT=off" | bash tools/r03_exp.sh c5tg --steps 3 --warmup 2 --workload C5t --segments 125
grep "vh times" gpurun_out/r03/c5tg/1.err | tail -4
grep "vh times" gpurun_out/r03/c5tg/2.err | tail -4
printf '%s\n' "VH_TIMES=1" | bash tools/r03_exp.sh c5g --steps 3 --warmup 2 --workload C5 --segments 125
grep "vh times" gpurun_out/r03/c5g/1.err | tail -4

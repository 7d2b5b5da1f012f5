#!/bin/bash
mkdir -p gpurun_out/r03 gpurun_out/verify
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( time timeout 1500 python -m pytest tests -q -m gpu -x ) > gpurun_out/verify/gpu_default.log 2>&1; tail -4 gpurun_out/verify/gpu_default.log
for f in 0 1048576; do echo "== C5t flags=$f"; VH_TIMES=1 python bench.py --workload C5t --segments 125 --steps 5 --warmup 2 --no-cpu --no-reference-layout --flags $f 2>&1 | grep -E "vh times|parity_checked" | tail -2 | cut -c1-200; done
for v in "0.7 0.7" "0.5 0.5" "0.35 0.5" "0.5 0.35" "0.35 0.35"; do set -- $v; echo "== C5 load_g=$1 load_s=$2"; VH_HP_LOAD_G=$1 VH_HP_LOAD_S=$2 VH_TIMES=1 python bench.py --workload C5 --segments 125 --steps 5 --warmup 2 --no-cpu --no-check --no-reference-layout 2>&1 | grep "vh times" | tail -1; done
for b in 1 2 4 8; do echo "== C5 bpp=$b"; VH_HP_BPP=$b VH_TIMES=1 python bench.py --workload C5 --segments 125 --steps 5 --warmup 2 --no-cpu --no-check --no-reference-layout 2>&1 | grep "vh times" | tail -1; done

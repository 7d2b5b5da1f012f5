#!/bin/bash
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for f in 0 524288; do echo "== C5t flags=$f"; VH_TIMES=1 python bench.py --workload C5t --segments 125 --steps 5 --warmup 2 --no-cpu --no-reference-layout --flags $f > gpurun_out/r03/c5t_$f.json 2> gpurun_out/r03/c5t_$f.err; grep "vh times" gpurun_out/r03/c5t_$f.err | tail -1; python -c "
import json; d=json.loads(open('gpurun_out/r03/c5t_$f.json').read().strip().splitlines()[-1]); print(d['parity_checked'], round(d['ms_per_step'],3), d['roofline']['kernel'], round(d['roofline']['kernel_ms'],3))"; done
for v in "0.85 0.85" "0.7 0.85" "0.85 0.7" "1.0 1.0"; do set -- $v; echo "== C5 load_g=$1 load_s=$2"; VH_HP_LOAD_G=$1 VH_HP_LOAD_S=$2 VH_TIMES=1 python bench.py --workload C5 --segments 125 --steps 5 --warmup 2 --no-cpu --no-check --no-reference-layout 2>&1 | grep "vh times" | tail -1; done
( timeout 900 python -m pytest tests/test_gpu_jit.py tests/test_gpu_typed.py tests/test_gpu_fullsize.py tests/test_gpu_distributed.py -q -m gpu -x ) > gpurun_out/r03/hp_tests2.log 2>&1; tail -3 gpurun_out/r03/hp_tests2.log
for b in 0 6; do echo "== hisel bpc=$b"; VH_BLOCKS_PER_CU=$b python tools/hisel_probe.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if d.get('variant') in ('default', 'no_lanes'): print(d['case'], d['variant'], d.get('kernel_ms'), d.get('jit'))
"; done

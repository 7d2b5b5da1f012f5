#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_jit.py tests/test_gpu_pack.py tests/test_gpu_narrow.py -x -q -m gpu 2>&1 | tail -5
export VH_PLACE_TRIALS=1
for i in 1 2; do python bench.py --steps 10 --warmup 3 --no-cpu --no-check --no-reference-layout 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 one-word tuples run $i', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), d['roofline']['kernel'])"; done
for i in 1 2; do VH_NO_NARROW_TUPLES=1 python bench.py --steps 10 --warmup 3 --no-cpu --no-check --no-reference-layout 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 two-word tuples run $i', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"; done
python tools/hisel_probe.py 2>&1 | tail -8

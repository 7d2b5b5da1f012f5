#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for i in 1 2 3 4 5 6; do VH_PLACE_TRIALS=1 python bench.py --steps 10 --warmup 3 --no-cpu --no-check --no-reference-layout 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('trials=1 run $i', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"; done
for i in 1 2 3 4; do python bench.py --steps 10 --warmup 3 --no-cpu --no-check --no-reference-layout 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('search run $i', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"; done

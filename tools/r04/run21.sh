#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for b in 2 3 4 5 6 8; do for i in 1 2; do VH_BLOCKS_PER_CU=$b python bench.py --steps 10 --warmup 3 --no-cpu --no-check --no-reference-layout --no-cpu-parallel --no-warm 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('blocks/CU $b run $i', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"; done; done
for u in 16384 32768; do VH_UNIT_ROWS=$u python bench.py --steps 10 --warmup 3 --no-cpu --no-check --no-reference-layout --no-cpu-parallel --no-warm 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('unit rows $u', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"; done

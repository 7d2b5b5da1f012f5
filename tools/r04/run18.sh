#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -4
for i in 1 2 3; do python bench.py --steps 10 --warmup 3 --no-cpu --no-check --no-reference-layout 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prepared run $i', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), d['config']['pool_placed_by_measurement'], d['config']['pack_seconds'])"; done
python bench.py --steps 10 --warmup 3 --no-cpu --no-check --no-reference-layout --no-warm 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no-warm', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), d['config']['pool_placed_by_measurement'])"

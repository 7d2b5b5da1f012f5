#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1200 python -m pytest tests/test_gpu_pack.py tests/test_gpu_jit.py -x -q -m gpu 2>&1 | tail -6
for i in 1 2 3; do python bench.py --steps 10 --warmup 3 --no-cpu --no-check --no-reference-layout --no-cpu-parallel --no-warm 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bit records run $i', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), d['config']['device_bytes'])"; done
for i in 1 2 3; do VH_NO_PACK_BITS=1 python bench.py --steps 10 --warmup 3 --no-cpu --no-check --no-reference-layout --no-cpu-parallel --no-warm 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('byte records run $i', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), d['config']['device_bytes'])"; done

#!/bin/bash
one() { env "$@" python bench.py --workload $W --segments $S --no-cpu --no-check --no-reference-layout --no-cpu-parallel --steps 50 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W segs', $S, '$*', 'ms_per_step', round(d['ms_per_step'], 4), 'kernel_ms', round(d['roofline']['kernel_ms'], 4), d['roofline']['kernel'][:50])"; }
W=C1; for S in 2 10 50; do one VH_X=1; done
W=C2; S=100; one VH_X=1
python tools/env_ab_probe.py 1000 - 2>/dev/null | grep '^{' | cut -c1-140
python tools/env_ab_probe.py 125 - 2>/dev/null | grep '^{' | cut -c1-140
python tools/c5_probe.py 2>&1 | tail -1 | cut -c1-120
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_typed.py tests/test_gpu_jit.py tests/test_gpu_hpart.py -x -q 2>&1 | tail -2

#!/bin/bash
for S in 25 50 100 1000; do
for E in VH_X=1 VH_TEST_BLOCKS_PER_CU=1 VH_TEST_NO_JIT_LANES=1; do
  env $E python bench.py --workload C2 --segments $S --no-cpu --no-check --no-reference-layout --no-cpu-parallel --steps 30 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', $S, '$E', 'ms_per_step', round(d['ms_per_step'], 4), 'kernel_ms', round(d['roofline']['kernel_ms'], 4), d['roofline']['kernel'][:50])"
done; done

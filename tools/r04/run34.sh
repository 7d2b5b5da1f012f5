#!/bin/bash
# The tail of a small dense query as one launch (small_tail_kernel) against merge + emission + header (VH_TEST_NO_SMALL_TAIL=1): C1 and C2.
one() { env "$@" python bench.py --workload $W --no-cpu --no-reference-layout --no-cpu-parallel --steps 50 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W', '$*', 'ms_per_step', round(d['ms_per_step'], 4), 'kernel_ms', round(d['roofline']['kernel_ms'], 4))"; }
for W in C1 C2; do one VH_X=1; one VH_TEST_NO_SMALL_TAIL=1; done

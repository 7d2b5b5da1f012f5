#!/bin/bash
# C1 (10 M rows, SUM WHERE k = K): the scan kernel's time against the number of blocks — what every wave's counters at the end of the kernel cost
# (profiles/r04/NOTES.md, "The other fixed cost"). Before vh_scan_block_end: 72.6 us default, 33.4 with one block per CU, 25.5 with 64 K-row units.
one() { env "$@" python bench.py --workload C1 --segments $S --no-cpu --no-check --no-reference-layout --no-cpu-parallel --steps 50 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C1 segs', $S, '$*', 'ms_per_step', round(d['ms_per_step'], 4), 'kernel_ms', round(d['roofline']['kernel_ms'], 4), d['roofline']['kernel'][:50])"; }
for S in 2 10 50 200; do one VH_X=1; done
S=10; one VH_JIT=off; one VH_TEST_BLOCKS_PER_CU=1; one VH_TEST_UNIT_ROWS=65536; one VH_TEST_UNIT_ROWS=262144

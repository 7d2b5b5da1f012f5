#!/bin/bash
# Where do the fixed ~0.12 ms of the scan of a small shard go? kernel_ms (scan + phase 2) over shard sizes, unit sizes and blocks per CU.
mkdir -p gpurun_out/r04
O=gpurun_out/r04/shard_fixed.txt; : > $O
run() { echo "== $*" >> $O; env "$@" python tools/scale_proxy.py $SIZES 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['segments'], d['query_only_ms'], d['kernel_ms'])" >> $O; }
for U in 32768 65536 131072 262144 524288; do SIZES="1" run VH_UNIT_ROWS=$U; done
for U in 32768 65536; do SIZES="8 4 2" run VH_UNIT_ROWS=$U; done
SIZES="4 2" run VH_UNIT_ROWS=16384
cat $O

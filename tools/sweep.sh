#!/bin/bash
# knob sweep on one GPU: blocks per CU x unit rows, C3 at 1000 segments (bench scale)
SEG=${1:-1000}
for bpc in 3 4 5 8; do for unit in 8192 16384 65536; do
  echo "bpc=$bpc unit=$unit: $(VH_BLOCKS_PER_CU=$bpc VH_UNIT_ROWS=$unit python tools/ubench.py --segments $SEG --workloads C3 --iters 7 --variants default 2>&1 | grep '"default"' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["kernel_ms"], "ms", round(d["bref_GBs"]), "GB/s")')"
done; done

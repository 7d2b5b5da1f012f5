#!/bin/bash
# Process-to-process spread of the C3 kernel time under different environments: tools/var_bench.sh rounds "ENV1=.." "ENV2=.." ...
N=$1; shift
for i in $(seq $N); do
  for E in "$@"; do
    env $E timeout 300 python bench.py --no-cpu --no-check --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['roofline']['kernel_ms'],3), d['config']['table_path'])" "$E"
  done
done

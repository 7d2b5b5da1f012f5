#!/bin/bash
# A/B on ONE box: alternate bench.py runs with two argument sets (boxes of the pool differ by more than most changes are worth).
# usage: tools/ab_bench.sh "<args A>" "<args B>" [rounds]
A="$1"; B="$2"; N=${3:-3}
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), d['config']['table_path'], d['roofline']['kernel'])" "$1"; }
for i in $(seq $N); do
  timeout 300 python bench.py --no-cpu --no-check --steps 30 $A 2>/dev/null | show "A[$A]"
  timeout 300 python bench.py --no-cpu --no-check --steps 30 $B 2>/dev/null | show "B[$B]"
done

#!/bin/bash
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
c3() { TAG=$1; shift
  env "$@" VH_TRACE_ALLOC=1 python bench.py --steps 10 --warmup 3 --no-cpu --no-check --no-reference-layout > gpurun_out/r03/c3_$TAG.json 2> gpurun_out/r03/c3_$TAG.err
  python - <<P
import json
d=json.loads(open('gpurun_out/r03/c3_$TAG.json').read().strip().splitlines()[-1])
print("c3 $TAG $@", round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), d['config']['table_path'])
P
  grep "candidate" gpurun_out/r03/c3_$TAG.err | awk '{printf "%s ", $(NF-1)} END {print ""}'; grep "scratch trial" gpurun_out/r03/c3_$TAG.err | tail -1
}
for rep in 1 2 3 4 5 6 7 8 9 10; do c3 ad_r$rep; done

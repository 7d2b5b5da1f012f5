#!/bin/bash
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for i in 1 2 3 4 5 6 7 8 9 10; do
  VH_TRACE_ALLOC=1 python bench.py --steps 10 --warmup 3 --no-cpu --no-check --no-reference-layout > gpurun_out/r03/pl_$i.json 2> gpurun_out/r03/pl_$i.err
  python - <<P
import json
d=json.loads(open('gpurun_out/r03/pl_$i.json').read().strip().splitlines()[-1])
print("run $i", round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))
P
  grep "derived layouts" gpurun_out/r03/pl_$i.err | tail -1 | cut -c1-160; grep "scratch trial" gpurun_out/r03/pl_$i.err | awk '{print $(NF-3), $(NF-2), $(NF-1), $NF}' | tr '\n' ';'; echo
done

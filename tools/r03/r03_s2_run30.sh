#!/bin/bash
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for seg in 125 1000; do for b in 0 12 8 4 2; do
  VH_PART_BPP=$b VH_TIMES=1 python bench.py --segments $seg --steps 20 --warmup 3 --no-cpu --no-check --no-reference-layout > gpurun_out/r03/pb_$seg_$b.json 2> gpurun_out/r03/pb_$seg_$b.err
  python - <<P
import json
d=json.loads(open('gpurun_out/r03/pb_$seg_$b.json').read().strip().splitlines()[-1])
print("seg=$seg bpp=$b", round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4))
P
  grep "vh times" gpurun_out/r03/pb_$seg_$b.err | tail -1
done; done

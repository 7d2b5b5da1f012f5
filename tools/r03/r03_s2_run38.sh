#!/bin/bash
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for seg in 1000 125; do for et in 0 256 512 0 256 512; do
  VH_EXT_TUPLES=$et python bench.py --segments $seg --steps 20 --warmup 3 --no-cpu --no-check --no-reference-layout 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"seg=$seg et=$et\", round(d[\"ms_per_step\"], 4), round(d[\"roofline\"][\"kernel_ms\"], 4))"
done; done

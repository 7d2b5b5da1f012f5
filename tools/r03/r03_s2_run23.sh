#!/bin/bash
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( timeout 1200 python -m pytest tests/test_gpu_jit.py tests/test_gpu_typed.py tests/test_gpu_parity.py tests/test_gpu_pack.py -q -m gpu -x ) > gpurun_out/r03/stage64_tests.log 2>&1; tail -3 gpurun_out/r03/stage64_tests.log
for ns in 0 1; do
  E=""; [ $ns = 1 ] && E="VH_NO_STAGE=1"
  echo "== one level, 38 partitions (VH_PART_TABLE_KB=32) $E"; env $E VH_PART_TABLE_KB=32 python bench.py --steps 10 --warmup 3 --no-cpu --no-check --no-reference-layout 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'], 3), round(d['roofline']['kernel_ms'], 3), d['roofline']['kernel'])"
  echo "== two levels (4 M groups) $E"; env $E timeout 600 python tools/part2_probe.py 1000 50,120,1000 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if d['variant'] != 'direct': print(d['d0_lt'], d['sel'], d['variant'], d['kernel_ms'], d['lanes'], d['kernel'][:30])
"
done
echo "== two levels, compiled compacting kernel at 100 % (PLAN_NO_LANES)"; timeout 300 python - <<P
import json, os, sys
sys.path.insert(0, os.getcwd() if os.path.exists('bench.py') else '/root/repo')
from viyadb_amd import capi, executor, synth
from viyadb_amd.executor import AggPlan, GroupSpec
executor.init(0)
w = synth.c3(); t = synth.create_device_table(w, 1000)
plan = AggPlan(filter=[], groups=[GroupSpec(5), GroupSpec(2)], metrics=w.plan.metrics, flags=capi.PLAN_FORCE_PART | capi.PLAN_NO_LANES | capi.PLAN_NO_PACK, groups_hint=4 << 20)
ms = [t.query_agg(plan).scan_kernel_ms for _ in range(5)]
r = t.query_agg(plan); print(round(sorted(ms[1:])[1], 3), r.kernel, r.lanes)
P

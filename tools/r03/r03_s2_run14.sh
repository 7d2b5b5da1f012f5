#!/bin/bash
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
echo "== scale proxy"; timeout 600 python tools/scale_proxy.py 1 2 4 8 2>&1 | tail -5
echo "== two-level probe (4 M groups)"; timeout 600 python tools/part2_probe.py 1000 50,120,1000 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['d0_lt'], d['sel'], d['variant'], d['kernel_ms'], d['kernel'][:70], d['lanes'], d['retries'])
"
( timeout 900 python -m pytest tests/test_gpu_typed.py tests/test_gpu_parity.py -q -m gpu -x ) > gpurun_out/r03/typed_tests.log 2>&1; tail -3 gpurun_out/r03/typed_tests.log

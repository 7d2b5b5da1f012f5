#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_jit.py tests/test_gpu_typed.py -x -q -m gpu 2>&1 | tail -4
python tools/part2_probe.py 1000 50,120,1000 2>/dev/null | grep "two-level\|default" | cut -c1-250

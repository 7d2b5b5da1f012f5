#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1200 python -m pytest tests/test_gpu_hpart.py tests/test_gpu_parity.py tests/test_gpu_distributed.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -4
python tools/c5_probe.py C5 125 4 2>&1 | tail -1 | cut -c1-120
VH_NO_OFF32=1 python tools/c5_probe.py C5 125 4 2>&1 | tail -1 | cut -c1-120

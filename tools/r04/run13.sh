#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python tools/c5_probe.py C5 125 4 2>&1 | tail -1 | cut -c1-120
C5_CARD64=1 python tools/c5_probe.py C5 125 4 2>&1 | tail -1 | cut -c1-120
VH_HP_BPP=16 python tools/c5_probe.py C5 125 4 2>&1 | tail -1 | cut -c1-120
python tools/c5_probe.py C5t 125 4 2>&1 | tail -1 | cut -c1-120
timeout 900 python -m pytest tests/test_gpu_hpart.py tests/test_gpu_fullsize.py tests/test_gpu_distributed.py tests/test_gpu_cluster_merge.py -x -q -m gpu 2>&1 | tail -3

#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests/test_gpu_hpart.py -x -q -m gpu 2>&1 | tail -8
VH_TIMES=1 python tools/c5_probe.py C5 125 4 2>&1 | grep "vh times\|kernel_ms" | tail -3 | cut -c1-260
VH_TIMES=1 VH_HP_STREAM=0 python tools/c5_probe.py C5 125 4 2>&1 | grep "vh times\|kernel_ms" | tail -2 | cut -c1-200
VH_HP_BPP=16 python tools/c5_probe.py C5 125 4 2>&1 | tail -1 | cut -c1-200
python tools/c5_probe.py C5t 125 4 2>&1 | tail -1 | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k c5 2>&1 | tail -3

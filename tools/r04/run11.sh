#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for b in 64 32 128 0; do echo "deliver blocks $b"; VH_DELIVER_BLOCKS=$b python tools/c5_probe.py C5 125 4 2>&1 | tail -1 | cut -c1-120; done
echo "no stream"; VH_HP_STREAM=0 python tools/c5_probe.py C5 125 4 2>&1 | tail -1 | cut -c1-120
python tools/c5_probe.py C5t 125 4 2>&1 | tail -1 | cut -c1-120
timeout 900 python -m pytest tests/test_gpu_hpart.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3

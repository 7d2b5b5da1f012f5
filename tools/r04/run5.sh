#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for a in 0 1 2 3 4; do VH_HP_ABLATE=$a python tools/c5_probe.py C5 125 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate $a', d['kernel_ms'])"; done
for b in 2 8 16 32; do VH_HP_BPP=$b python tools/c5_probe.py C5 125 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bpp $b', d['kernel_ms'])"; done
for l in 0.5 0.85 1.0; do VH_HP_LOAD_G=$l VH_HP_LOAD_S=$l python tools/c5_probe.py C5 125 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('load $l', d['kernel_ms'])"; done
VH_TIMES=1 python tools/c5_probe.py C5 125 3 2>&1 | grep "vh times" | tail -2

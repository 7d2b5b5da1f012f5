#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for i in 1 2 3; do VH_TIMES=1 python tools/c5_probe.py C5 125 4 2>&1 | grep "vh times\|kernel_ms" | tail -2 | cut -c1-200; done
VH_TIMES=1 VH_HP_STREAM=0 python tools/c5_probe.py C5 125 4 2>&1 | grep "vh times\|kernel_ms" | tail -2 | cut -c1-200
python tools/c5_probe.py C5t 125 4 2>&1 | tail -1 | cut -c1-200

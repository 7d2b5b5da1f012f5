#!/bin/bash
mkdir -p gpurun_out/r04
O=gpurun_out/r04/ab_probe.txt; : > $O
python tools/env_ab_probe.py 1000 - VH_TEST_UNIT_ROWS=16384 2>/dev/null | grep '^{' >> $O
python tools/env_ab_probe.py 125 - VH_TEST_UNIT_ROWS=4096 2>/dev/null | grep '^{' >> $O
python tools/env_ab_probe.py 250 - VH_TEST_UNIT_ROWS=4096 2>/dev/null | grep '^{' >> $O
python tools/env_ab_probe.py 500 - VH_TEST_UNIT_ROWS=8192 2>/dev/null | grep '^{' >> $O
cat $O
python tools/c5_probe.py 2>&1 | tail -1 | cut -c1-200
VH_TEST_UNIT_ROWS=4096 python tools/c5_probe.py 2>&1 | tail -1 | cut -c1-200

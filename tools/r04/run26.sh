#!/bin/bash
mkdir -p gpurun_out/r04
O=gpurun_out/r04/ab_probe.txt; : > $O
python tools/env_ab_probe.py 1000 - VH_TEST_BLOCKS_PER_CU=2 VH_TEST_BLOCKS_PER_CU=4 VH_TEST_BLOCKS_PER_CU=5 VH_TEST_BLOCKS_PER_CU=6 VH_TEST_BLOCKS_PER_CU=8 2>/dev/null | grep '^{' >> $O
python tools/env_ab_probe.py 125 - VH_TEST_BLOCKS_PER_CU=2 VH_TEST_BLOCKS_PER_CU=4 VH_TEST_BLOCKS_PER_CU=6 2>/dev/null | grep '^{' >> $O
cat $O

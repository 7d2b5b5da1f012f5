#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
echo "taskset node0"; taskset -c 0-63 python tools/c5_probe.py C5 125 4 2>&1 | tail -1 | cut -c1-120
echo "taskset node1"; taskset -c 64-127 python tools/c5_probe.py C5 125 4 2>&1 | tail -1 | cut -c1-120
echo "HSA_ENABLE_SDMA=0"; HSA_ENABLE_SDMA=0 python tools/c5_probe.py C5 125 4 2>&1 | tail -1 | cut -c1-120
echo "stream=0 node0"; VH_HP_STREAM=0 taskset -c 0-63 python tools/c5_probe.py C5 125 4 2>&1 | tail -1 | cut -c1-120
python - <<'P'
import os
print("affinity of a fresh python:", len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:4], "...")
P
cat /proc/self/status | grep -i "cpus_allowed_list\|mems_allowed_list"
numactl -H 2>/dev/null | head -8; grep -i "MemFree\|MemTotal" /sys/devices/system/node/node*/meminfo

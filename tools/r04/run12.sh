#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for k in 8 4 2; do for pr in 0 1; do echo "chunks $k prio $pr"; if [ $pr = 1 ]; then export VH_COPY_PRIO=1; else unset VH_COPY_PRIO; fi; VH_HP_STREAM=$k python tools/c5_probe.py C5 125 4 2>&1 | tail -1 | cut -c1-120; done; done
unset VH_COPY_PRIO
(cd /tmp && VH_HP_STREAM=4 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r04/kt_c5 -o c5 -- python /root/repo/tools/c5_probe.py C5 125 3 > /root/repo/gpurun_out/r04/kt_c5.log 2>&1)
python tools/pmc_summary.py --kernel-stats $(find gpurun_out/r04/kt_c5 -name "*_results.db" | head -1) gpurun_out/r04/c5_stream_kernel_stats.csv; head -8 gpurun_out/r04/c5_stream_kernel_stats.csv | cut -c1-150
rm -rf gpurun_out/r04/kt_c5

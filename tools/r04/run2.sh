#!/bin/bash
# packed 16-byte tuples for C5: parity first, then what they buy
O=gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests/test_gpu_hpart.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -15
python tools/c5_probe.py C5 > $O/c5_pk.json 2> $O/c5_pk.err; tail -c 500 $O/c5_pk.json
VH_NO_HP_PACK=1 python tools/c5_probe.py C5 2>/dev/null | tail -c 300
(cd /tmp && rocprofv3 --kernel-trace --stats -d /root/repo/$O/kt_c5 -o c5 -- python /root/repo/tools/c5_probe.py C5 125 3 > /root/repo/$O/kt_c5.log 2>&1)
python tools/pmc_summary.py --kernel-stats $(find $O/kt_c5 -name "*_results.db" | head -1) $O/c5_pk_kernel_stats.csv; head -8 $O/c5_pk_kernel_stats.csv | cut -c1-150
rm -rf $O/kt_c5

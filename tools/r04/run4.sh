#!/bin/bash
# the ranges' aggregation compiled per plan shape: parity, then kernel times + instruction mix
O=gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests/test_gpu_hpart.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -8
python tools/c5_probe.py C5 2>/dev/null | tail -c 400
python tools/c5_probe.py C5t 2>/dev/null | tail -c 400
(cd /tmp && rocprofv3 --kernel-trace --stats -d /root/repo/$O/kt_c5 -o c5 -- python /root/repo/tools/c5_probe.py C5 125 3 > /root/repo/$O/kt_c5.log 2>&1)
python tools/pmc_summary.py --kernel-stats $(find $O/kt_c5 -name "*_results.db" | head -1) $O/c5_jitagg_kernel_stats.csv; head -6 $O/c5_jitagg_kernel_stats.csv | cut -c1-150
rm -rf $O/kt_c5 $O/pmc_tmp
(cd /tmp && rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d /root/repo/$O/pmc_tmp -o c5 -- python /root/repo/tools/c5_probe.py C5 125 2 > /root/repo/$O/pmc_tmp.log 2>&1)
python tools/pmc_kernel.py $O/pmc_tmp hpagg
rm -rf $O/pmc_tmp

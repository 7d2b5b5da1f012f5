#!/bin/bash
# where hp_aggregate_kernel's cycles go (SQ counters, their own passes), and the existing ablations on the packed form
O=gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM"; do
  rm -rf $O/pmc_tmp
  (cd /tmp && rocprofv3 --pmc $set -d /root/repo/$O/pmc_tmp -o c5 -- python /root/repo/tools/c5_probe.py C5 125 2 > /root/repo/$O/pmc_tmp.log 2>&1)
  python tools/pmc_kernel.py $O/pmc_tmp hp_aggregate
  python tools/pmc_kernel.py $O/pmc_tmp hp_scatter
done
rm -rf $O/pmc_tmp
for a in 8 12 13 15; do VH_HP_ABLATE=$a python tools/c5_probe.py C5 125 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate $a', d['kernel_ms'])"; done

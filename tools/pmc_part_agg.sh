#!/bin/bash
# SQ counters of part_agg_kernel on the C3 query (separate rocprofv3 passes, no trace domains). usage: tools/pmc_part_agg.sh
REPO=$PWD; OUT=gpurun_out/pmc_part_agg; mkdir -p $OUT
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"; do
  D=$OUT/$(echo $SET | tr ' ' '_' | cut -c1-40)
  (cd /tmp && TMPDIR=/tmp timeout 240 rocprofv3 --pmc $SET -d $REPO/$D -o p -- python $REPO/bench.py --no-cpu --no-check --steps 3 --warmup 1 > $REPO/$D.log 2>&1)
  timeout 60 python tools/pmc_kernel.py $D part_agg
done
rm -rf $OUT/SQ_*/ 

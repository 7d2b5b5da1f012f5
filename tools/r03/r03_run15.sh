#!/bin/bash
mkdir -p gpurun_out/r03
export TMPDIR=/tmp; REPO=$PWD
for G in 128 192; do
(cd /tmp && VH_HP_GRID_A=$G rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/r03/kt_c5 -o c5 -- python $REPO/bench.py --workload C5 --segments 125 --steps 3 --warmup 3 --no-cpu --no-check > $REPO/gpurun_out/r03/kt_c5.log 2>&1)
echo "grid A $G"; python tools/last_query_kernels.py gpurun_out/r03/kt_c5 viya_jit_scan | grep -E "scatter|aggregate"
rm -rf gpurun_out/r03/kt_c5
done
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT"; do
  D=gpurun_out/r03/pmc_d; rm -rf $D
  (cd /tmp && timeout 240 rocprofv3 --pmc $SET -d $REPO/$D -o p -- python $REPO/bench.py --workload C5 --segments 125 --no-cpu --no-check --steps 2 --warmup 3 > $REPO/$D.log 2>&1)
  python - "$D" <<'PY'
import glob, os, sqlite3, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*_results.db"), recursive=True):
    for k, c, v in sqlite3.connect(f).execute("select kernel_name, counter_name, value from counters_collection"):
        if "hp_" in k or "viya_jit" in k: acc[k[:40]][c].append(float(v))
for k, cs in acc.items():
    for c, v in sorted(cs.items()):
        print("%-42s %-22s %s" % (k, c, " ".join("%.4g" % x for x in v[-4:])))
PY
done
rm -rf gpurun_out/r03/pmc_d

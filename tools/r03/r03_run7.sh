#!/bin/bash
mkdir -p gpurun_out/r03
export TMPDIR=/tmp; REPO=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/r03/kt_c3 -o c3 -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu --no-check > $REPO/gpurun_out/r03/kt_c3.log 2>&1)
python tools/pmc_summary.py --kernel-stats $(find gpurun_out/r03/kt_c3 -name "*_results.db" | head -1) gpurun_out/r03/c3_kernel_stats.csv; head -14 gpurun_out/r03/c3_kernel_stats.csv | cut -c1-160
python - <<'PY'
import sqlite3, glob
db = glob.glob('gpurun_out/r03/kt_c3/**/*_results.db', recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
# last query: find last 'part_agg' and print the kernels around it
idx = [i for i, r in enumerate(rows) if 'viya_jit_scan' in r[0]]
i0 = idx[-2]; i1 = idx[-1]
t0 = rows[i0][1]
for r in rows[i0 - 3:i1 + 1]:
    print("%9.1f us  +%8.1f us  %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[0][:60]))
PY
for C in FETCH_SIZE WRITE_SIZE; do
  D=gpurun_out/r03/pmc_d; rm -rf $D
  (cd /tmp && timeout 240 rocprofv3 --pmc $C -d $REPO/$D -o p -- python $REPO/bench.py --no-cpu --no-check --steps 3 --warmup 1 > $REPO/$D.log 2>&1)
  timeout 60 python tools/pmc_kernel.py $D "viya_jit" | grep -v "^void"
  timeout 60 python tools/pmc_kernel.py $D "part_agg" | grep -v "^void"
done
rm -rf gpurun_out/r03/pmc_d gpurun_out/r03/kt_c3

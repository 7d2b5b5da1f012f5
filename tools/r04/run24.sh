#!/bin/bash
# The eighth-shard query (one rank of the 8-GPU run, one-rank RCCL communicator) kernel by kernel.
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python tools/scale_proxy.py 1 8 > gpurun_out/r04/proxy_1_8.txt 2>&1; cat gpurun_out/r04/proxy_1_8.txt
rocprofv3 --kernel-trace --stats -d gpurun_out/r04/p8 -o p8 -- python tools/scale_proxy.py 8 > gpurun_out/r04/p8.log 2>&1
python - <<'PY'
import glob, sqlite3
db = glob.glob('gpurun_out/r04/p8/**/*_results.db', recursive=True)[0]
c = sqlite3.connect(db)
for row in c.execute("select name, total_calls, total_duration, average from top_kernels limit 14"):
    print('%-90s %6d %10.1f us avg' % (row[0][:90], row[1], row[3] / 1e3))
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print(cols)
rows = list(c.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if 'emit_groups' in r[0]]
a, b = idx[-2] + 1, min(len(rows), idx[-1] + 3)
t0 = rows[a][1]
for r in rows[a:b]:
    print('%9.1f %9.1f %8.1f us  %s' % ((r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[0][:90]))
PY
rm -rf gpurun_out/r04/p8

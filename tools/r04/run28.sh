#!/bin/bash
# One query's timeline (rocprofv3 kernel trace): C2 and C3 through bench.py, the last step's dispatches with start / end / duration.
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for W in ${WL:-C2 C3}; do
  rm -rf gpurun_out/r04/tl
  rocprofv3 --kernel-trace --stats -d gpurun_out/r04/tl -o tl -- python bench.py --workload $W --no-cpu --no-check --no-reference-layout --no-cpu-parallel --steps 10 --warmup 2 > gpurun_out/r04/tl_$W.json 2> gpurun_out/r04/tl_$W.err
  python - $W <<'PY'
import glob, sqlite3, sys, json
db = glob.glob('gpurun_out/r04/tl/**/*_results.db', recursive=True)[0]
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if 'publish_header' in r[0]]
a, b = idx[-3] + 1, idx[-1] + 1
t0 = rows[a][1]
print('==', sys.argv[1], open('gpurun_out/r04/tl_%s.json' % sys.argv[1]).read()[:0])
for r in rows[a:b]:
    print('%9.1f %9.1f %8.1f us  %s' % ((r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[0][:80]))
d = json.loads(open('gpurun_out/r04/tl_%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])
PY
done
rm -rf gpurun_out/r04/tl

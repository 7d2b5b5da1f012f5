#!/bin/bash
# The scan kernel's duration over tiny to small shards (rocprofv3 kernel trace), with and without the tuple appends.
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r04/scan_fixed.txt; : > $O
one() {  # label segs env...
  local L=$1 S=$2; shift 2
  rm -rf gpurun_out/r04/pf
  env "$@" PREPARE=0 rocprofv3 --kernel-trace --stats -d gpurun_out/r04/pf -o pf -- python tools/env_ab_probe.py $S - > gpurun_out/r04/pf.log 2>&1
  python - "$L" "$S" >> $O <<'PY'
import glob, sqlite3, sys
db = glob.glob('gpurun_out/r04/pf/**/*_results.db', recursive=True)[0]
c = sqlite3.connect(db)
out = []
for name, calls, avg in c.execute("select name, total_calls, average from top_kernels"):
    if any(k in name for k in ('viya_jit_scan', 'part_agg', 'dense_merge', 'emit_groups')):
        mn = c.execute("select min(duration) from kernels where name = ?", (name,)).fetchone()[0]
        out.append('%s avg %.1f min %.1f us' % (name.split('(')[0][-28:], avg / 1e3, mn / 1e3))
print(sys.argv[1], sys.argv[2], ' | '.join(out))
PY
}
for S in 62 125 250 1000; do one default $S VH_X=0; one noappend $S VH_JIT_FLAGS=-DVJ_ABL=8; done
rm -rf gpurun_out/r04/pf
cat $O

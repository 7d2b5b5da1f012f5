"""Print per-kernel means of the counters in a rocprofv3 --pmc run (rocpd database). usage: pmc_kernel.py <dir> [substr]"""
import glob, os, sqlite3, sys
from collections import defaultdict
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
    for k, c, v in sqlite3.connect(f).execute("select kernel_name, counter_name, value from counters_collection"):
        if sub in k:
            acc[k][c].append(float(v))
for k, cs in acc.items():
    print(k[:70])
    for c, v in sorted(cs.items()):
        print("   %-34s mean %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))

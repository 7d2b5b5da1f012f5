"""Kernels of the last query of a rocprofv3 --kernel-trace run, in launch order. usage: last_query_kernels.py <dir> <first kernel substring>"""
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + '/**/*_results.db', recursive=True)[0]
rows = sqlite3.connect(db).execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r[0]]
i0 = idx[-1]
t0 = rows[i0][1]
for r in rows[i0 - 2:]:
    print("%9.1f us  +%8.1f us  %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[0][:70]))

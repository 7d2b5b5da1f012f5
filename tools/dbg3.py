import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from viyadb_amd import executor, synth
from viyadb_amd.executor import AggPlan
executor.init(0)
w = synth.c3(segment_rows=100_003)
def run(t, flags):
    return t.query_agg(AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, flags=flags))
def key(r):
    o = np.lexsort([r.keys[1], r.keys[0]])
    return o, r.keys[0][o].astype(np.int64) * 1000 + r.keys[1][o]
for trial in range(6):
    t = synth.create_device_table(w, 4, 99_991)
    seq = [320, 320, 0, 320, 192, 320] if trial % 2 == 0 else [0, 320, 320, 192, 320, 320]
    res = [run(t, f) for f in seq]
    ref = run(t, 0)
    oref, kref = key(ref)
    out = []
    for f, r in zip(seq, res):
        o, k = key(r)
        if len(k) != len(kref):
            miss = sorted(set(kref.tolist()) - set(k.tolist()))
            i = int(np.nonzero(kref == miss[0])[0][0])
            out.append((f, "MISSING", miss[:3], int(ref.states[1][oref][i]), "seen", r.algorithmic_bytes >> 20, "oor", r.algorithmic_bytes & 0xFFFFF, "passed", r.passed_recs, "oorw0", hex(r.scanned_segments)))
        else:
            d = (r.states[0][o].astype(np.int64) - ref.states[0][oref].astype(np.int64))
            out.append((f, "ok" if not d.any() else "DIFF", (r.algorithmic_bytes >> 20) if f == 320 else 0))
    print(trial, out)
    t.close()

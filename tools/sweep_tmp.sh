python - <<'PY'
import json, sys
sys.path.insert(0, '.')
from viyadb_amd import executor, synth
executor.init(0)
w = synth.c3()
t = synth.create_device_table(w, 1000)
for flags, label in ((0, "default (direct atomics)"), (64, "FORCE_PART staged"), (64 | 256, "FORCE_PART lanes tiles")):
    plan = executor.AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, flags=flags, groups_hint=w.plan.groups_hint)
    ms = []
    for _ in range(7):
        r = t.query_agg(plan); ms.append(r.scan_kernel_ms)
    print(json.dumps({"variant": label, "path": r.path, "lanes": r.lanes, "kernel_ms": sorted(ms)[3]}))
PY

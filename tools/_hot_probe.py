import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viyadb_amd import capi, executor, synth
from viyadb_amd.executor import AggPlan
executor.init(0)
for name in ("C5", "C5h"):
    w = synth.WORKLOADS[name]()
    t = synth.create_device_table(w, 125)
    for label, metrics in (("users+count", [3, 4]), ("count", [4]), ("users", [3])):
        plan = AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=metrics, flags=capi.PLAN_CARD32 | capi.PLAN_NO_HPART, groups_hint=0)
        ms = []
        for _ in range(4):
            r = t.query_agg(plan, copy=False); ms.append(r.scan_kernel_ms)
        print(json.dumps({"table": name, "metrics": label, "kernel_ms": [round(x, 2) for x in ms], "kernel": r.kernel, "retries": r.retries, "groups": r.ngroups}), flush=True)
    t.close()

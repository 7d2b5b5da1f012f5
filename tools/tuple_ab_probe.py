import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["VH_TEST_HOOKS"] = "1"
from viyadb_amd import capi, executor, synth
from viyadb_amd.executor import AggPlan
executor.init(0)
w = synth.c3()
t = synth.create_device_table(w, 1000)
plan = AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, flags=capi.PLAN_CARD32, groups_hint=w.plan.groups_hint)
t.pack(t.gather_columns(plan)); t.predpack(t.filter_columns(plan)); t.warm(plan)
for rnd in range(2):
    for label, env in (("default", {}), ("no_balance", {"VH_NO_PART_BALANCE": "1"}), ("no_tuple4", {"VH_NO_TUPLE4": "1"}), ("neither", {"VH_NO_PART_BALANCE": "1", "VH_NO_TUPLE4": "1"})):
        for k in ("VH_NO_PART_BALANCE", "VH_NO_TUPLE4"):
            os.environ.pop(k, None)
        os.environ.update(env)
        ks, ws = [], []
        for i in range(25):
            q0 = time.perf_counter(); r = t.query_agg(plan, copy=False); ws.append((time.perf_counter() - q0) * 1e3); ks.append(r.scan_kernel_ms)
        ks, ws = sorted(ks[5:]), sorted(ws[5:])
        print(json.dumps({"variant": label, "kernel_ms": round(ks[len(ks) // 2], 4), "wall_ms": round(ws[len(ws) // 2], 4), "kernel": r.kernel}), flush=True)

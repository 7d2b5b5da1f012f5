"""vh_query_agg_sharded on real device memory. This box has one GPU, so: two ranks share cuda:0 and talk through the callback
transport over gloo (every line of the sharded protocol except the RCCL calls themselves), and one rank goes through RCCL
proper (world 1 with the protocol forced on). Every scenario must equal the oracle on the union of the ranks' rows."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, json
    sys.path.insert(0, {root!r})
    import numpy as np, torch, torch.distributed as dist
    from viyadb_amd import capi, distributed, executor, synth
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    executor.init(0)
    comm = distributed.Comm.rccl(dist) if {backend!r} == "rccl" else distributed.Comm.gloo(dist)
    spec = {spec!r}
    root = spec.get("root", 0)
    if spec["kind"] == "synth":
        w = synth.WORKLOADS[spec["wl"]](segment_rows=50000)
        lo, hi = distributed.shard_segments(9, rank, world)
        t = synth.create_device_table(w, hi - lo, 50000, row_base=lo * 50000)
        nk = len(w.plan.groups)
        having = [("rel", nk + len(w.plan.metrics) - 1, 4, 1)] if spec.get("having") else []    # last metric (a count) > 1, on MERGED groups
        plan = executor.AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, flags=spec.get("flags", 0), having=having)
    else:
        from oracle import viya_oracle as vo
        from tests import shard_scenarios
        from tests.planner import mirror_table, plan_from_query
        tab, q, flags, hint = shard_scenarios.build(spec["name"], [rank], world)
        t = mirror_table(tab)
        plan = plan_from_query(tab, vo.parse_query(tab, q), now=1496570140, flags=flags | spec.get("flags", 0), groups_hint=hint)
    retries = 0
    for it in range(2):
        if it == 1 and spec.get("grow") and rank == 1:      # one rank's table changes between two runs of the same plan
            tab2 = shard_scenarios.build(spec["name"], [rank], world, grown=True)[0]
            seg = tab2.segments[-1]
            t.sync_segment(len(tab2.segments) - 1, [a[:seg["size"]] for a in seg["d"]] + [a[:seg["size"]] for a in seg["m"]], seg["size"])
        res = distributed.sharded_query(t, plan, comm, root=root)
        retries = max(retries, res.retries)      # (the second run already knows how many groups to expect)
    torch.cuda.synchronize()
    if root >= 0:
        assert (res.returned == 0) == (rank != root or res.ngroups == 0), (rank, res.returned)
    np.savez({out!r} + ".%d.npz" % rank, *(res.keys + res.states), ngroups=res.ngroups, nk=len(res.keys), returned=res.returned, path=res.path,
             scanned=res.scanned_recs, passed=res.passed_recs, retries=retries,
             calls=json.dumps(comm.transport.calls if comm.transport else {{}}))
    dist.barrier()
    t.close()
    comm.close()
    dist.destroy_process_group()
''')


def _launch(tmp_path, spec, backend, nproc, env=None):
    out = str(tmp_path / "res")
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=out, spec=spec, backend=backend))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
                        "127.0.0.1", "--master-port", str(port), str(script)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return [np.load(out + ".%d.npz" % k) for k in range(nproc)]


def _rows(got):
    nk = int(got["nk"])
    arrs = [got["arr_%d" % i] for i in range(len([k for k in got.files if k.startswith("arr_")]))]
    return arrs[:nk], arrs[nk:]


def _check(gots, st, root=0, having_keep=None):
    from tests.parity import sort_rows
    if having_keep is not None:
        st.keys = [k[having_keep] for k in st.keys]
        st.states = [x[having_keep] for x in st.states]
    if root >= 0:
        keys, states = _rows(gots[root])
        for k, g in enumerate(gots):
            if k != root:
                assert int(g["returned"]) == 0
    else:                                             # results left with their owners: the union must be the answer, owners disjoint
        parts = [_rows(g) for g in gots]
        keys = [np.concatenate([p[0][i] for p in parts]) for i in range(len(parts[0][0]))]
        states = [np.concatenate([p[1][j] for p in parts]) for j in range(len(parts[0][1]))]
        assert all(int(g["returned"]) > 0 for g in gots)
    assert len(states[0]) == len(st.states[0]), (len(states[0]), len(st.states[0]))
    pg, po = sort_rows(keys, states), sort_rows(st.keys, st.states)
    for a, b in zip(keys + states, st.keys + st.states):
        if a.dtype.kind == "f":
            np.testing.assert_allclose(a[pg], b[po], rtol=1e-9)
        else:
            assert np.array_equal(a[pg], b[po].astype(a.dtype))
    for g in gots:                                    # counters are global on every rank
        assert int(g["scanned"]) == st.scanned_recs and int(g["passed"]) == st.passed_recs


def _synth_oracle(wl):
    from oracle import viya_oracle as vo
    from tests.parity import build_oracle_table
    from viyadb_amd import synth
    w = synth.WORKLOADS[wl](segment_rows=50000)
    return vo.scan_aggregate(vo.parse_query(build_oracle_table(w, 9, 50000), w.query), now=getattr(w, "now", None))


HPART = 1 | (1 << 18) | (1 << 20)      # hash organisation, compiled scan, hashed partitioning (vh_hpart.h): the ranks' (group, id) pairs then come out of the tuple pool


@pytest.mark.parametrize("wl,flags", [("C3", 0), ("C3", 64), ("C2", 0), ("C2", 2), ("C1", 0), ("C3", 1), ("C5t", 0), ("C5", 0),
                                      ("C5t", 2048), ("C5", 2048), ("C3", 1 | 2048), ("C3", 16 | 32),      # 2048: the hash table as records
                                      ("C5", HPART), ("C5t", HPART), ("C3", HPART)])
def test_two_ranks_one_gpu(tmp_path, wl, flags):
    gots = _launch(tmp_path, {"kind": "synth", "wl": wl, "flags": flags}, "gloo", 2)
    st = _synth_oracle(wl)
    assert int(gots[0]["ngroups"]) == st.ngroups
    _check(gots, st)


def test_one_rank_through_rccl(tmp_path):
    """The whole protocol through RCCL itself — ncclAllGather, ncclAllReduce, ncclReduce with the states' own types, grouped
    ncclSend / ncclRecv — with the single rank this box allows."""
    for wl in ("C3", "C5t", "C5", "C2"):
        gots = _launch(tmp_path, {"kind": "synth", "wl": wl}, "rccl", 1, env={"VH_TEST_SHARDED_WORLD1": "1"})
        _check(gots, _synth_oracle(wl))


@pytest.mark.parametrize("wl,flags", [("C3", 0), ("C3", 1), ("C5t", 0), ("C5", 0), ("C5", 1 | (1 << 18) | (1 << 20))])
def test_two_ranks_having_on_merged_groups(tmp_path, wl, flags):
    gots = _launch(tmp_path, {"kind": "synth", "wl": wl, "flags": flags, "having": True}, "gloo", 2)
    st = _synth_oracle(wl)
    assert int(gots[0]["ngroups"]) == st.ngroups
    _check(gots, st, having_keep=st.states[-1] > 1)


@pytest.mark.parametrize("wl", ["C5t", "C5", "C3"])
def test_sparse_results_stay_with_their_owners(tmp_path, wl):
    gots = _launch(tmp_path, {"kind": "synth", "wl": wl, "flags": 1, "root": -1}, "gloo", 2)
    st = _synth_oracle(wl)
    assert all(int(g["ngroups"]) == st.ngroups for g in gots)
    _check(gots, st, root=-1)


def _scenario_oracle(name):
    from oracle import viya_oracle as vo
    from tests import shard_scenarios
    tab, q, _, _ = shard_scenarios.build(name, [0, 1])
    return vo.scan_aggregate(vo.parse_query(tab, q), now=1496570140)


@pytest.mark.parametrize("name,flags,path", [("disjoint_ranges", 0, None), ("disjoint_ranges", 64, "dense_part"), ("disjoint_ranges", 2, "dense_global"),
                                             ("part_vs_global", 0, None), ("part_vs_global", 8, None), ("one_rank_overflows", 0, "hash"),
                                             ("empty_shard", 0, None), ("empty_shard", 1, "hash"), ("uniform", 16 | 32, None)])
def test_ranks_that_see_different_data_agree_on_one_plan(tmp_path, name, flags, path):
    """Shards whose group columns span disjoint ranges, whose filters pass 100 % vs 1 % of the rows, of which one overflows
    its hash table or holds no rows at all: left to themselves the ranks would plan different dense ranges and different
    organisations (silently wrong sums, or mismatched collectives). Unsigned MIN / MAX metrics ride along."""
    gots = _launch(tmp_path, {"kind": "scenario", "name": name, "flags": flags}, "gloo", 2)
    st = _scenario_oracle(name)
    assert int(gots[0]["ngroups"]) == st.ngroups
    _check(gots, st)
    assert str(gots[0]["path"]) == str(gots[1]["path"])
    if path:
        assert str(gots[0]["path"]) == path
    if name == "one_rank_overflows":
        assert int(gots[0]["retries"]) >= 1 and int(gots[1]["retries"]) == int(gots[0]["retries"])   # one rank's overflow re-plans both


def test_cached_agreement_is_dropped_when_one_rank_changes(tmp_path):
    """The second run of a plan takes the agreement of the first from the communicator's cache — unless a rank's table changed in
    between (here: a new segment whose group values lie outside every range agreed before): that rank says so in the verdict,
    all ranks agree afresh, and the answer covers the new rows."""
    from oracle import viya_oracle as vo
    from tests import shard_scenarios
    for flags in (0, 2):
        gots = _launch(tmp_path, {"kind": "scenario", "name": "disjoint_ranges", "flags": flags, "grow": True}, "gloo", 2)
        tab, q, _, _ = shard_scenarios.build("disjoint_ranges", [0, 1], grown=True)
        st = vo.scan_aggregate(vo.parse_query(tab, q), now=1496570140)
        assert int(gots[0]["ngroups"]) == st.ngroups
        _check(gots, st)
        calls = __import__("json").loads(str(gots[0]["calls"]))
        assert calls["allgather"] == 2, calls          # first run + the re-agreement; a steady second run would have made it 1


def test_steady_state_needs_no_allgather(tmp_path):
    """Same plan, unchanged tables: the second run's only host-visible collective is the verdict all-reduce."""
    gots = _launch(tmp_path, {"kind": "synth", "wl": "C3"}, "gloo", 2)
    calls = __import__("json").loads(str(gots[0]["calls"]))
    assert calls["allgather"] == 1 and calls["reduce"] >= 4, calls   # 2 runs x (verdict + one state buffer)


def test_bench_multi_rank_flow_on_one_gpu(tmp_path):
    """bench.py's N>1 path end to end (sharding, vh_query_agg_sharded, barrier + max-over-ranks timing, one JSON line from
    rank 0), with two ranks sharing this box's single GPU over the callback transport (VH_BENCH_BACKEND=gloo); the real run uses RCCL."""
    import json
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, VH_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--segments", "40"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "strong" and d["value"] > 1e9
    assert d["config"]["rows"] == 40_000_000 and d["config"]["groups"] == 100_000
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1


def test_bench_falls_back_when_rccl_refuses(tmp_path):
    """Two ranks on this box's ONE GPU with the default (RCCL) data plane: ncclCommInitRank refuses a duplicate device, every rank
    learns about it and the run continues over the callback transport — and says so in its output line."""
    import json
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k != "VH_BENCH_BACKEND"}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--segments", "20"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and "RCCL communicator failed" in d["config"]["parallelism"], d["config"]["parallelism"]


HOST_WORKER = textwrap.dedent('''
    import os, sys, json
    sys.path.insert(0, {root!r})
    import numpy as np, torch, torch.distributed as dist
    from viyadb_amd import distributed, executor, hostdb
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    executor.init(0)
    comm = distributed.Comm.gloo(dist)
    spec = json.load(open({spec!r}))
    db = hostdb.Database({{}}, device=0)
    db.create_table(spec["table"])
    db.load("events", spec["rows"][rank], now=spec["now"])
    db.join_node(comm)
    out = []
    for q in spec["queries"]:
        rows, stats = db.query(q, now=spec["now"])
        out.append({{"rows": rows, "stats": {{k: stats[k] for k in ("scanned_recs", "scanned_segments", "aggregated_recs", "output_recs")}}}})
    json.dump(out, open({out!r} + ".%d.json" % rank, "w"))
    dist.barrier()
    db.close()
    comm.close()
    dist.destroy_process_group()
''')


def test_cxx_database_queries_all_ranks_rows(tmp_path):
    """The C++ host shim end to end over two ranks (Database::JoinNode -> GpuAggregate -> vh_query_agg_sharded): every rank loaded
    its own rows, the same vdb_query runs on both, rank 0 returns what ONE database holding all rows returns — string and time
    dimensions, AVG over the hidden count, count-distinct, HAVING, sort + limit."""
    import json
    import random
    from viyadb_amd import executor, hostdb
    rnd = random.Random(5)
    now = 1496570140
    table = {"name": "events", "segment_size": 5000,
             "dimensions": [{"name": "country"}, {"name": "event", "cardinality": 100}, {"name": "t", "type": "time"}, {"name": "n", "type": "uint"}],
             "metrics": [{"name": "count", "type": "count"}, {"name": "revenue", "type": "double_sum"}, {"name": "best", "type": "int_max"},
                         {"name": "avg_len", "type": "long_avg"}, {"name": "users", "type": "bitset"}]}
    countries, events = ["US", "RU", "IL", "KZ", "CH", "AZ"], ["open", "purchase", "refund", "donate"]

    def make_rows(n, seed):
        r = random.Random(seed)
        # every rank first sees every string once, in the same order: the same dictionary codes everywhere
        rows = [[c, e, str(now - 1), "0", "0.5", "1", "7", "1"] for c in countries for e in events]
        for _ in range(n):
            rows.append([r.choice(countries), r.choice(events), str(now - r.randrange(0, 40 * 86400)), str(r.randrange(0, 50)),
                         str(r.randrange(0, 4000) / 8.0), str(r.randrange(-1000, 1000)), str(r.randrange(1, 500)), str(r.randrange(0, 300))])
        return rows
    rows = [make_rows(7000, 11), make_rows(9000, 12)]
    queries = [
        {"type": "aggregate", "table": "events", "dimensions": ["country", "event"], "metrics": ["count", "revenue", "best", "avg_len"],
         "filter": {"op": "ne", "column": "country", "value": "RU"}, "sort": [{"column": "revenue"}, {"column": "country", "ascending": True}]},
        {"type": "aggregate", "table": "events", "select": [{"column": "t", "granularity": "day"}, {"column": "country"}, {"column": "users"}, {"column": "count"}],
         "filter": {"op": "lt", "column": "n", "value": "25"}, "having": {"op": "ge", "column": "count", "value": "3"},
         "sort": [{"column": "count"}, {"column": "country", "ascending": True}, {"column": "t", "ascending": True}], "limit": 40},
        {"type": "aggregate", "table": "events", "dimensions": ["n"], "metrics": ["count", "users"], "filter": {"op": "gt", "column": "count", "value": "0"},
         "sort": [{"column": "n", "ascending": True}]},
    ]
    spec = str(tmp_path / "spec.json")
    json.dump({"table": table, "rows": rows, "queries": queries, "now": now}, open(spec, "w"))
    out = str(tmp_path / "out")
    script = tmp_path / "worker.py"
    script.write_text(HOST_WORKER.format(root=ROOT, spec=spec, out=out))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    got = [json.load(open(out + ".%d.json" % k)) for k in range(2)]
    executor.init(0)
    db = hostdb.Database({}, device=0)
    try:
        db.create_table(table)
        db.load("events", rows[0], now=now)
        db.load("events", rows[1], now=now)
        for k, q in enumerate(queries):
            want, st = db.query(q, now=now)
            assert got[0][k]["rows"] == want, (k, got[0][k]["rows"][:3], want[:3])
            assert got[1][k]["rows"] == []
            assert got[0][k]["stats"]["output_recs"] == len(want) and len(want) > 0
            assert got[0][k]["stats"]["aggregated_recs"] == st["aggregated_recs"]
    finally:
        db.close()


def test_bench_from_a_plain_shell_two_ranks_one_gpu():
    """`python bench.py --gpus 2 …` with nothing in the environment: the launcher re-executes itself under
    torch.distributed.run; with one GPU here the two ranks share it over the callback transport, and the line must SAY so —
    "rccl".transport = "gloo-fallback", "degraded" true — with the parity gate run on both shards."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["VH_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--segments", "4", "--segment-rows", "20000",
                        "--steps", "3", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["parity_checked"] and d["config"]["rows"] == 80000
    assert d["rccl"]["nranks"] == 2 and d["rccl"]["transport"] == "gloo-fallback" and d["degraded"] is True
    assert len(d["rccl"]["devices"]) == 2 and d["rccl"]["distinct_devices"] == 1


def test_comm_info_reads_the_rccl_communicator_back(tmp_path):
    """vh_comm_info on a real RCCL communicator (world 1): transport, rank count and device come from RCCL itself."""
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent('''
        import os, sys, json
        sys.path.insert(0, %r)
        import torch, torch.distributed as dist
        from viyadb_amd import distributed, executor
        torch.cuda.set_device(0)
        dist.init_process_group("gloo")
        executor.init(0)
        c = distributed.Comm.rccl(dist)
        i = c.info()
        assert i["transport"] == "rccl" and i["nranks"] == 1 and i["rank"] == 0 and i["device"] == 0 and ":" in i["pci_bus_id"], i
        g = distributed.Comm.gloo(dist)
        j = g.info()
        assert j["transport"] == "callbacks" and j["nranks"] == 1 and j["pci_bus_id"] == i["pci_bus_id"], j
        g.close(); c.close()
        dist.destroy_process_group()
    ''' % ROOT))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])

"""The N>1 flow on real device memory: two ranks (both on cuda:0, gloo backend, because this box has one
GPU) each mirror half of the segments, run vh_query_launch, reduce the library-owned dense partial tables
in place through zero-copy views, and rank 0 finalises. Must equal the oracle on the whole table."""
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
    import os, sys
    sys.path.insert(0, {root!r})
    import numpy as np, torch, torch.distributed as dist
    from viyadb_amd import distributed, executor, synth
    torch.cuda.set_device(0)
    dist.init_process_group({backend!r}, **({{"device_id": torch.device("cuda", 0)}} if {backend!r} == "nccl" else {{}}))
    rank, world = dist.get_rank(), dist.get_world_size()
    executor.init(0, stream=torch.cuda.current_stream().cuda_stream)
    w = synth.WORKLOADS[{wl!r}](segment_rows=50000)
    total = 9
    lo, hi = distributed.shard_segments(total, rank, world)
    t = synth.create_device_table(w, hi - lo, 50000, row_base=lo * 50000)
    nk = len(w.plan.groups)
    having = [("rel", nk + len(w.plan.metrics) - 1, 4, 1)] if {having} else []    # last metric (a count) > 1, on MERGED groups
    plan = executor.AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, flags={flags}, having=having)
    for _ in range(2):
        res = distributed.sharded_query(torch, dist, t, plan, world, force_collectives=True)
    torch.cuda.synchronize()
    assert (res is None) == (rank != 0)
    if rank == 0:
        np.savez({out!r}, *(res.keys + res.states), ngroups=res.ngroups, nk=len(res.keys), returned=res.returned)
    dist.barrier()
    t.close()
    dist.destroy_process_group()
''')


def test_one_rank_rccl(tmp_path):
    """The same flow through the nccl (= RCCL) backend with a single rank: dtype views, in-place reduce on the
    library's buffers and the all-to-all of the hash path go through RCCL itself (this box has one GPU)."""
    for wl, flags in (("C3", 0), ("C5t", 0), ("C5", 0)):
        _run(tmp_path, wl, flags, "nccl", 1)


@pytest.mark.parametrize("wl,flags", [("C3", 0), ("C3", 64), ("C2", 0), ("C2", 2), ("C1", 0), ("C3", 1), ("C5t", 0), ("C5", 0),
                                      ("C5t", 2048), ("C5", 2048), ("C3", 1 | 2048)])      # 2048: the hash table as records
def test_two_ranks_one_gpu(tmp_path, wl, flags):
    _run(tmp_path, wl, flags, "gloo", 2)


@pytest.mark.parametrize("wl,flags", [("C3", 0), ("C3", 1), ("C5t", 0)])
def test_two_ranks_having_on_merged_groups(tmp_path, wl, flags):
    _run(tmp_path, wl, flags, "gloo", 2, having=True)


def _run(tmp_path, wl, flags, backend, nproc, having=False):
    out = str(tmp_path / "res.npz")
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=out, wl=wl, flags=flags, backend=backend, having=having))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
                        "127.0.0.1", "--master-port", str(port), str(script)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    got = np.load(out)
    from oracle import viya_oracle as vo
    from tests.parity import build_oracle_table, sort_rows
    from viyadb_amd import synth
    w = synth.WORKLOADS[wl](segment_rows=50000)
    st = vo.scan_aggregate(vo.parse_query(build_oracle_table(w, 9, 50000), w.query), now=getattr(w, "now", None))
    nk = int(got["nk"])
    arrs = [got["arr_%d" % i] for i in range(nk + len(st.states))]
    keys, states = arrs[:nk], arrs[nk:]
    assert int(got["ngroups"]) == st.ngroups
    if having:
        keep = st.states[-1] > 1
        st.keys = [k[keep] for k in st.keys]
        st.states = [x[keep] for x in st.states]
        assert int(got["returned"]) == int(keep.sum())
    pg, po = sort_rows(keys, states), sort_rows(st.keys, st.states)
    for a, b in zip(keys + states, st.keys + st.states):
        assert np.array_equal(a[pg], b[po])


def test_dense_partials_are_one_collective_for_c3():
    """C3/C4: SUM(int64) + COUNT carried as SOP_ADD32P, both 64-bit integer sums laid out back to back -> the
    library hands the caller ONE reduce buffer (no presence bytes, no second call): the collective is latency-bound."""
    from viyadb_amd import capi, executor, synth
    executor.init(0)
    w = synth.WORKLOADS["C3"](segment_rows=50000)
    t = synth.create_device_table(w, 4, 50000)
    try:
        plan = executor.AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics)
        res = t.query_launch(plan)
        bufs = t.device_buffers(res)
        assert len(bufs) == 1 and bufs[0][2] == capi.U64 and bufs[0][3] == 0 and bufs[0][1] >= 2 * 100_000
        out = t.finalize(res, plan)
        assert out.ngroups > 0
        # without the presence carrier the layout is [presence | states...]: still correct, more buffers
        res = t.query_launch(executor.AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=w.plan.metrics, flags=capi.PLAN_NO_CARRIER))
        assert len(t.device_buffers(res)) >= 2
        t.discard(res)
    finally:
        t.close()


def test_bench_multi_rank_flow_on_one_gpu(tmp_path):
    """bench.py's N>1 path end to end (sharding, collective, barrier + max-over-ranks timing, one JSON line from
    rank 0), with two gloo ranks sharing this box's single GPU (VH_BENCH_BACKEND=gloo); the real run uses RCCL."""
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

"""N>1 path on CPU: two gloo ranks shard a table's segments, build dense partial aggregate tables
(oracle-side, numpy), reduce them with the product's reduction glue and must reproduce the
single-process result bit for bit (integer wrap-around included)."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_segments_partition():
    from viyadb_amd.distributed import shard_segments
    for total in (0, 1, 7, 8, 125, 1000):
        for world in (1, 2, 3, 4, 8):
            parts = [shard_segments(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, {root!r})
    import numpy as np, torch, torch.distributed as dist
    from viyadb_amd import capi, distributed, synth
    from tests.parity import build_oracle_table
    from oracle import viya_oracle as vo
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    w = synth.c3(segment_rows=20000)
    total = 7
    lo, hi = distributed.shard_segments(total, rank, world)
    # this rank's shard, same global row ids as the unsharded table
    ot = build_oracle_table(w, hi - lo, 20000, row_base=lo * 20000)
    st = vo.scan_aggregate(vo.parse_query(ot, w.query))
    G = 1000 * 100
    gid = st.keys[0].astype(np.int64) * 100 + st.keys[1].astype(np.int64)
    present = np.zeros(G, dtype=np.uint8); present[gid] = 1
    s0 = np.zeros(G, dtype=np.uint64); s0[gid] = st.states[0].astype(np.int64).view(np.uint64) + np.uint64(2**63 + 12345)  # force wrap-around
    s1 = np.zeros(G, dtype=np.uint32); s1[gid] = st.states[1] + np.uint32(4294960000)
    distributed.reduce_host_partials(torch, dist, [(present, capi.U8, distributed.RED_MAX), (s0, capi.U64, distributed.RED_SUM),
                                                   (s1, capi.U32, distributed.RED_SUM)])
    if rank == 0:
        np.savez({out!r}, present=present, s0=s0, s1=s1)
    dist.destroy_process_group()
''')


def test_two_rank_reduce_matches_single_process(tmp_path):
    out = str(tmp_path / "reduced.npz")
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=out))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(out)
    # single-process reference: whole table, each rank's offset counted once per rank that had the group
    from oracle import viya_oracle as vo
    from tests.parity import build_oracle_table
    from viyadb_amd import distributed, synth
    w = synth.c3(segment_rows=20000)
    G = 100000
    want_p = np.zeros(G, dtype=np.uint8)
    want0 = np.zeros(G, dtype=np.uint64)
    want1 = np.zeros(G, dtype=np.uint32)
    with np.errstate(over="ignore"):
        for rank in range(2):
            lo, hi = distributed.shard_segments(7, rank, 2)
            ot = build_oracle_table(w, hi - lo, 20000, row_base=lo * 20000)
            st = vo.scan_aggregate(vo.parse_query(ot, w.query))
            gid = st.keys[0].astype(np.int64) * 100 + st.keys[1].astype(np.int64)
            want_p[gid] = 1
            want0[gid] += st.states[0].astype(np.int64).view(np.uint64) + np.uint64(2 ** 63 + 12345)
            want1[gid] += st.states[1] + np.uint32(4294960000)
    assert np.array_equal(got["present"], want_p)
    assert np.array_equal(got["s0"], want0)
    assert np.array_equal(got["s1"], want1)
    # and the merged table equals the unsharded aggregate (without the artificial offsets)
    ot = build_oracle_table(w, 7, 20000)
    st = vo.scan_aggregate(vo.parse_query(ot, w.query))
    gid = st.keys[0].astype(np.int64) * 100 + st.keys[1].astype(np.int64)
    assert np.array_equal(np.nonzero(want_p)[0], np.sort(gid))

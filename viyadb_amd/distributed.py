"""Multi-GPU glue for the aggregate path: one process per GPU, segments sharded as
contiguous blocks, per-GPU partial aggregates reduced with RCCL over xGMI.

The reference's own cross-worker pattern is "partial aggregate per shard, then re-aggregate"
(src/cluster/query/agg_runner.cc:83-140, over HTTP + a temp table). Here every rank holds the same
dense, identically indexed partial tables (presence bytes + one state array per metric), so the
merge is one collective per array with the op the library reports (SUM / MIN / MAX)."""
from __future__ import annotations

from typing import List, Tuple

from . import capi

# vh_elem -> (numpy typestr for the reduce view, needs_order_fix)
# two's-complement adds are the same bits signed or unsigned, so u32/u64 SUMs are reduced as i32/i64
_SUM_VIEW = {capi.U8: "|u1", capi.U32: "<i4", capi.U64: "<i8", capi.I32: "<i4", capi.I64: "<i8", capi.F32: "<f4", capi.F64: "<f8"}
RED_SUM, RED_MIN, RED_MAX = 0, 1, 2


def shard_segments(total_segments: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of segments owned by `rank`."""
    return total_segments * rank // world, total_segments * (rank + 1) // world


class _DevArray:
    """Zero-copy __cuda_array_interface__ view of a library-owned device buffer."""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def reduce_view_typestr(elem: int, reduce: int) -> str:
    if reduce == RED_SUM or elem in (capi.U8, capi.I32, capi.I64, capi.F32, capi.F64):
        return _SUM_VIEW[elem]
    raise NotImplementedError("unsigned 32/64-bit MIN/MAX partials need an order-preserving view")


_VIEW_CACHE = {}   # (ptr, count, typestr) -> torch view; the library's scratch is grow-only, so views repeat query after query


def _device_view(torch, ptr, count, typestr):
    key = (int(ptr), int(count), typestr)
    t = _VIEW_CACHE.get(key)
    if t is None:
        if len(_VIEW_CACHE) > 64:
            _VIEW_CACHE.clear()
        t = _VIEW_CACHE[key] = torch.as_tensor(_DevArray(ptr, count, typestr), device="cuda")
    return t


def reduce_partials(torch, dist, buffers: List[tuple], dst: int = 0):
    """buffers: [(ptr, count, elem, reduce)] from DeviceTable.device_buffers(). In-place reduce to `dst`."""
    ops = {RED_SUM: dist.ReduceOp.SUM, RED_MIN: dist.ReduceOp.MIN, RED_MAX: dist.ReduceOp.MAX}
    gloo = dist.get_backend() == "gloo"   # test rigs without RCCL: gloo has no device-side reduce-to-root
    for ptr, count, elem, reduce in buffers:
        t = _device_view(torch, ptr, count, reduce_view_typestr(elem, reduce))
        if gloo:
            dist.all_reduce(t, op=ops[reduce])
        else:
            dist.reduce(t, dst=dst, op=ops[reduce])


def reduce_host_partials(torch, dist, arrays: List[tuple], dst: int = 0):
    """CPU (gloo) twin of reduce_partials for tests: arrays = [(numpy array, elem, reduce)], reduced in place."""
    import numpy as np
    ops = {RED_SUM: dist.ReduceOp.SUM, RED_MIN: dist.ReduceOp.MIN, RED_MAX: dist.ReduceOp.MAX}
    for arr, elem, reduce in arrays:
        view = arr.view(np.dtype(reduce_view_typestr(elem, reduce)))
        t = torch.from_numpy(view)
        dist.reduce(t, dst=dst, op=ops[reduce])


def _merge_kind(kind: int) -> int:
    """Aggregation that merges two partial states of a metric of this kind (the reference retypes `count`
    as `long_sum` in its cluster merge table for the same reason: src/cluster/query/agg_runner.cc:66-76)."""
    if kind == capi.METRIC_MAX:
        return capi.METRIC_MAX
    if kind == capi.METRIC_MIN:
        return capi.METRIC_MIN
    if kind == capi.METRIC_BITSET:
        raise NotImplementedError("bitset (count-distinct) partials are cardinalities, not sets: they merge through their "
                                  "(group, id) pairs (exchange_hash_partials), not through host partials")
    return capi.METRIC_SUM


def merge_partials_by_reaggregation(table, plan, partials):
    """Merge per-rank partial aggregates that are NOT identically indexed (hash path: sparse keys): load them
    as segments of a temporary device table [group columns..., metric states...] and run the aggregate kernel
    over it with no filter — "partial aggregates are merged by re-aggregation", the reference's own cluster
    algebra (src/cluster/query/agg_runner.cc:93-140), on the GPU instead of over HTTP + upsert.
    partials: [(keys [np arrays], states [np arrays], hidden_count or None), ...]"""
    from .executor import AggPlan, DeviceTable, GroupSpec
    nk, nm = len(plan.groups), len(plan.metrics)
    has_hidden = any(p[2] is not None for p in partials)
    cols = [(capi.DIM_NUMERIC, table.cols[g.col][1]) for g in plan.groups]
    cols += [(_merge_kind(table.cols[m][0]), table.cols[m][1]) for m in plan.metrics]
    if has_hidden:
        cols.append((capi.METRIC_SUM, capi.U64))
    rows = max([len(p[1][0]) if p[1] else (len(p[0][0]) if p[0] else 0) for p in partials] + [1])
    tmp = DeviceTable(cols, segment_rows=rows, reserve_segments=len(partials))
    try:
        for s, (keys, states, hidden) in enumerate(partials):
            arrs = list(keys) + list(states) + ([hidden] if has_hidden else [])
            n = len(arrs[0]) if arrs else 0
            tmp.sync_segment(s, arrs, n)
        mplan = AggPlan(filter=[], groups=[GroupSpec(i) for i in range(nk)], metrics=list(range(nk, nk + nm + (1 if has_hidden else 0))),
                        groups_hint=sum(len(p[0][0]) if p[0] else 1 for p in partials))
        res = tmp.query_agg(mplan)
    finally:
        tmp.close()
    if has_hidden:
        res.hidden_count = res.states[-1]
        res.states = res.states[:-1]
    return res


def _merge_table(table, plan, has_hidden, rows, nseg, metrics=None):
    """Temporary device table [group columns..., metric states..., hidden count] + the plan that re-aggregates it.
    `metrics`: indices into plan.metrics to carry (default: all)."""
    from .executor import AggPlan, DeviceTable, GroupSpec
    nk = len(plan.groups)
    metrics = list(range(len(plan.metrics))) if metrics is None else list(metrics)
    cols = [(capi.DIM_NUMERIC, table.cols[g.col][1]) for g in plan.groups]
    cols += [(_merge_kind(table.cols[plan.metrics[j]][0]), table.cols[plan.metrics[j]][1]) for j in metrics]
    if has_hidden:
        cols.append((capi.METRIC_SUM, capi.U64))
    tmp = DeviceTable(cols, segment_rows=max(int(rows), 1), reserve_segments=nseg)
    mplan = AggPlan(filter=[], groups=[GroupSpec(i) for i in range(nk)],
                    metrics=list(range(nk, nk + len(metrics) + (1 if has_hidden else 0))))
    return tmp, mplan


def _all_to_all_columns(torch, dist, bufs, offs, world):
    """Ship every column buffer (rows of owner p at [offs[p], offs[p+1])) with one all-to-all; -> (received uint8
    device tensors, rows received)."""
    import numpy as np
    send = np.diff(offs.astype(np.int64))
    gloo = dist.get_backend() == "gloo"      # CPU test rig: gloo moves host tensors only
    send_t = torch.from_numpy(send.copy())
    send_t = send_t if gloo else send_t.cuda()
    recv_t = torch.empty_like(send_t)
    dist.all_to_all_single(recv_t, send_t)
    recv = recv_t.cpu().numpy().astype(np.int64)
    nrecv = int(recv.sum())
    received = []
    for ptr, count, elem, _red in bufs:
        es = capi.ELEM_SIZE[elem]
        src = (torch.as_tensor(_DevArray(ptr, count * es, "|u1"), device="cuda") if count
               else torch.empty(0, dtype=torch.uint8, device="cuda"))
        out = torch.empty(nrecv * es, dtype=torch.uint8, device="cuda")
        ins, outs = (send * es).tolist(), (recv * es).tolist()
        if gloo:
            host_out = torch.empty(nrecv * es, dtype=torch.uint8)
            dist.all_to_all_single(host_out, src.cpu(), outs, ins)
            out.copy_(host_out)
        else:
            dist.all_to_all_single(out, src, outs, ins)
        received.append(out)
    torch.cuda.current_stream().synchronize()
    return received, nrecv


def _row_keys(keys):
    """Group-key columns -> one sortable record per row (bit patterns, so float keys compare exactly)."""
    import numpy as np
    if not keys:
        return np.zeros(0, dtype=[("k0", np.uint64)])
    n = len(keys[0])
    rec = np.zeros(n, dtype=[("k%d" % i, np.uint64) for i in range(len(keys))])
    for i, k in enumerate(keys):
        rec["k%d" % i] = k.view(np.dtype("u%d" % k.dtype.itemsize)).astype(np.uint64)
    return rec


def exchange_hash_partials(torch, dist, table, plan, handle, world: int, having=None):
    """SURVEY 8(e), hash path: this rank's finalised groups are regrouped by owner = mix(key) % world in HBM
    (vh_result_partition), every column is shipped with one all-to-all (RCCL grouped send/recv over xGMI), and
    the owner merges what it received by re-aggregation on its own GPU, straight from the receive buffers
    (vh_segment_sync takes device addresses). Count-distinct metrics travel as their distinct (group, id) pairs
    (vh_result_partition_pairs) to the same owner, which counts them again. Returns this rank's OWNED groups."""
    import numpy as np
    from .executor import DeviceTable
    nk, nm = len(plan.groups), len(plan.metrics)
    bitset_js = [j for j, m in enumerate(plan.metrics) if m != capi.COL_ROWID and table.cols[m][0] == capi.METRIC_BITSET]
    plain_js = [j for j in range(nm) if j not in bitset_js]
    if bitset_js and having:
        raise NotImplementedError("HAVING over merged count-distinct values")
    offs, bufs = table.partition(handle, world)
    has_hidden = len(bufs) > nk + nm
    keep = list(range(nk)) + [nk + j for j in plain_js] + ([nk + nm] if has_hidden else [])
    received, nrecv = _all_to_all_columns(torch, dist, [bufs[i] for i in keep], offs, world)
    tmp, mplan = _merge_table(table, plan, has_hidden, nrecv, 1, plain_js)
    try:
        if nrecv:
            tmp.sync_segment_device(0, [t.data_ptr() for t in received], nrecv)
        mplan.groups_hint = nrecv
        if having:
            mplan.having = list(having)   # result-column indices are the same in the merge table
        res = tmp.query_agg(mplan)
    finally:
        tmp.close()
    if has_hidden:
        res.hidden_count = res.states[-1]
        res.states = res.states[:-1]
    if not bitset_js:
        return res
    # count-distinct: the owner re-counts the pairs it received
    plain_states = res.states
    states = [None] * nm
    for j, s_ in zip(plain_js, plain_states):
        states[j] = s_
    mine = _row_keys(res.keys)
    for j in bitset_js:
        poffs, pbufs = table.partition_pairs(handle, j, world)
        cols_recv, npairs = _all_to_all_columns(torch, dist, pbufs, poffs, world)
        wide = pbufs[-1][2] == capi.U64
        pcols = [(capi.DIM_NUMERIC, table.cols[g.col][1]) for g in plan.groups] + \
                [(capi.METRIC_BITSET, capi.BITSET64 if wide else capi.BITSET32)]
        ptab = DeviceTable(pcols, segment_rows=max(npairs, 1), reserve_segments=1)
        try:
            card = np.zeros(len(mine), dtype=np.uint64)
            if npairs:
                ptab.sync_segment_device(0, [t.data_ptr() for t in cols_recv[:nk]] + [None], npairs)
                ptab.sync_ids_device(0, nk, npairs, cols_recv[nk].data_ptr())
                from .executor import AggPlan, GroupSpec
                pres = ptab.query_agg(AggPlan(filter=[], groups=[GroupSpec(i) for i in range(nk)], metrics=[nk], groups_hint=len(mine)))
                theirs = _row_keys(pres.keys)
                _, inv = np.unique(np.concatenate([mine, theirs]), return_inverse=True)   # join on the key columns
                lookup = np.full(int(inv.max()) + 1, -1, dtype=np.int64)
                lookup[inv[:len(mine)]] = np.arange(len(mine))
                pos = lookup[inv[len(mine):]]
                if (pos < 0).any():
                    raise RuntimeError("count-distinct pairs arrived for a group this rank does not own")
                card[pos] = pres.states[0]
            states[j] = card
        finally:
            ptab.close()
    res.states = states
    return res


def sharded_query(torch, dist, table, plan, world: int, copy: bool = True, gather: bool = True,
                  force_collectives: bool = False):
    """One query over a table sharded across `world` ranks; the merged result lands on rank 0, the other
    ranks return None (they only contribute their partial tables to the collective). Hash-path queries with
    gather=False leave the result sharded: every rank returns the groups it owns."""
    import numpy as np
    if world == 1 and not force_collectives:
        return table.query_agg(plan, copy=copy)
    # The library must run on torch's current stream (executor.init(..., stream=...)): the collective is
    # then ordered after the scan kernels and the finalisation after the collective by stream order alone.
    res = table.query_launch(plan)
    try:
        bufs = table.device_buffers(res)
    except capi.VhError:
        bufs = None   # hash path: keys are sparse, partial tables are not identically indexed
    if bufs is not None:
        reduce_partials(torch, dist, bufs)
        if dist.get_backend() == "gloo":
            torch.cuda.current_stream().synchronize()
        if dist.get_rank() != 0:
            table.discard(res)
            return None
        return table.finalize(res, plan, copy=copy)
    # hash path: key-partitioned all-to-all, owner-computes merge
    pplan = plan
    if plan.having:                 # HAVING is a predicate on MERGED groups: partials run without it
        import dataclasses
        pplan = dataclasses.replace(plan, having=[])
        table.discard(res)
        res = table.query_agg_keep(pplan)
    else:
        try:
            table.finalize_keep(res)
        except capi.VhError:        # the partial table overflowed: vh_query_agg owns the re-plan loop
            table.discard(res)
            res = table.query_agg_keep(pplan)
    try:
        mine = table.collect(res, pplan, copy=False)
        local = (mine.scanned_recs, mine.scanned_segments, mine.passed_recs, mine.scan_kernel_ms, mine.algorithmic_bytes)
        owned = exchange_hash_partials(torch, dist, table, pplan, res, world, having=plan.having)
    finally:
        table.discard(res)
    stats = torch.tensor(local[:3], dtype=torch.int64)
    if dist.get_backend() != "gloo":
        stats = stats.cuda()
    dist.all_reduce(stats)
    owned.scanned_recs, owned.scanned_segments, owned.passed_recs = (int(x) for x in stats.tolist())
    owned.path = "hash+all_to_all"
    owned.scan_kernel_ms, owned.algorithmic_bytes = local[3], local[4]
    if not gather:
        return owned
    part = (owned.keys, owned.states, owned.hidden_count, owned.ngroups)
    gathered = [None] * world if dist.get_rank() == 0 else None
    dist.gather_object(part, gathered, dst=0)
    if dist.get_rank() != 0:
        return None
    nk, nm = len(plan.groups), len(plan.metrics)
    owned.keys = [np.concatenate([g[0][i] for g in gathered]) for i in range(nk)]
    owned.states = [np.concatenate([g[1][j] for g in gathered]) for j in range(nm)]
    if owned.hidden_count is not None:
        owned.hidden_count = np.concatenate([g[2] for g in gathered])
    owned.returned = len(owned.keys[0]) if nk else len(owned.states[0])
    owned.ngroups = sum(g[3] for g in gathered)     # groups before HAVING, over all owners
    return owned

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


def reduce_partials(torch, dist, buffers: List[tuple], dst: int = 0):
    """buffers: [(ptr, count, elem, reduce)] from DeviceTable.device_buffers(). In-place reduce to `dst`."""
    ops = {RED_SUM: dist.ReduceOp.SUM, RED_MIN: dist.ReduceOp.MIN, RED_MAX: dist.ReduceOp.MAX}
    gloo = dist.get_backend() == "gloo"   # test rigs without RCCL: gloo has no device-side reduce-to-root
    for ptr, count, elem, reduce in buffers:
        t = torch.as_tensor(_DevArray(ptr, count, reduce_view_typestr(elem, reduce)), device="cuda")
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


def sharded_query(torch, dist, table, plan, world: int, copy: bool = True):
    """One query over a table sharded across `world` ranks; the merged result lands on rank 0, the other
    ranks return None (they only contribute their partial tables to the collective)."""
    if world == 1:
        return table.query_agg(plan, copy=copy)
    # The library must run on torch's current stream (executor.init(..., stream=...)): the collective is
    # then ordered after the scan kernels and the finalisation after the collective by stream order alone.
    res = table.query_launch(plan)
    reduce_partials(torch, dist, table.device_buffers(res))
    if dist.get_backend() == "gloo":
        torch.cuda.current_stream().synchronize()
    if dist.get_rank() != 0:
        table.discard(res)
        return None
    return table.finalize(res, plan, copy=copy)

"""Multi-GPU glue for the aggregate path: one process per GPU, segments sharded as contiguous blocks.

Everything that matters happens behind the C boundary (include/viya_hip.h, vh_query_agg_sharded): plan agreement, the
verdict all-reduce, the RCCL reduce of dense partial tables, the key-partitioned exchange + merge of sparse ones. This
module only (a) splits segments between ranks, (b) creates the communicator — RCCL, with the unique id carried by
torch.distributed's store, or a callback transport over gloo so that two ranks can share ONE GPU in tests — and
(c) calls the C entry point.

The reference's own cross-worker pattern is "partial aggregate per shard, then re-aggregate"
(src/cluster/query/agg_runner.cc:83-140, over HTTP + a temp table)."""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import numpy as np

from . import capi

RED_SUM, RED_MIN, RED_MAX = 0, 1, 2


def shard_segments(total_segments: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of segments owned by `rank`."""
    return total_segments * rank // world, total_segments * (rank + 1) // world


# ---------------------------------------------------------------------------------------------- raw memory access
class _Mem:
    """Read / write `nbytes` at an address that may be host or device memory (the callback transport moves bytes through
    the host). Without a GPU (CPU-only tests of the transport itself) addresses are plain host pointers."""

    def __init__(self):
        import torch
        self.hip = None
        if torch.cuda.is_available():
            for name in ("libamdhip64.so.7", "libamdhip64.so"):
                try:
                    self.hip = C.CDLL(name)
                    break
                except OSError:
                    continue
            if self.hip is None:
                raise RuntimeError("libamdhip64 not found")
            self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            self.hip.hipStreamSynchronize.argtypes = [C.c_void_p]

    def sync(self, stream):
        if self.hip is not None and self.hip.hipStreamSynchronize(C.c_void_p(stream or 0)) != 0:
            raise RuntimeError("hipStreamSynchronize failed")

    def read(self, ptr, nbytes) -> np.ndarray:
        out = np.empty(int(nbytes), dtype=np.uint8)
        if nbytes:
            if self.hip is not None:
                if self.hip.hipMemcpy(out.ctypes.data, C.c_void_p(ptr), int(nbytes), 4) != 0:   # hipMemcpyDefault
                    raise RuntimeError("hipMemcpy failed")
            else:
                C.memmove(out.ctypes.data, ptr, int(nbytes))
        return out

    def write(self, ptr, arr: np.ndarray):
        arr = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
        if arr.size:
            if self.hip is not None:
                if self.hip.hipMemcpy(C.c_void_p(ptr), arr.ctypes.data, arr.size, 4) != 0:
                    raise RuntimeError("hipMemcpy failed")
            else:
                C.memmove(ptr, arr.ctypes.data, arr.size)


class GlooTransport:
    """vh_comm_ops over torch.distributed (gloo): every collective goes through host memory. For tests — two ranks on one
    GPU, or no GPU at all — not for speed. Unsigned 32/64-bit MIN / MAX are reduced through the order-preserving
    sign-bit flip (torch has no unsigned reductions; RCCL has, and the real transport uses them)."""

    def __init__(self, dist):
        import torch
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.mem = _Mem()
        self.errors = []
        self.calls = {"allgather": 0, "reduce": 0, "alltoallv": 0}
        self.ops = capi.CommOps(None, capi.CommOps.ALLGATHER(self._allgather), capi.CommOps.REDUCE(self._reduce),
                                capi.CommOps.ALLTOALLV(self._alltoallv))

    def _guard(self, fn, *a):
        try:
            fn(*a)
            return 0
        except Exception as e:   # noqa: BLE001 - a Python exception must not unwind through C
            self.errors.append(repr(e))
            return 1

    # -- host all-gather
    def _allgather(self, ctx, send, recv, nbytes):
        return self._guard(self.allgather, send, recv, nbytes)

    def allgather(self, send, recv, nbytes):
        torch = self.torch
        self.calls["allgather"] += 1
        mine = torch.from_numpy(np.frombuffer((C.c_char * nbytes).from_address(send), dtype=np.uint8).copy())
        outs = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(self.world)]
        self.dist.all_gather(outs, mine)
        gathered = torch.cat(outs).numpy()            # (keep it referenced while memmove reads it)
        C.memmove(recv, gathered.ctypes.data, nbytes * self.world)

    # -- in-place reduce of a "device" buffer
    def _reduce(self, ctx, buf, count, elem, op, root, stream):
        return self._guard(self.reduce, buf, count, elem, op, root, stream)

    def reduce(self, buf, count, elem, op, root, stream):
        torch = self.torch
        self.calls["reduce"] += 1
        self.mem.sync(stream)
        dt = np.dtype(capi.ELEM_NP[elem])
        a = self.mem.read(buf, count * dt.itemsize).view(dt)
        flip = None
        if dt.kind == "u" and dt.itemsize > 1:
            if op == RED_SUM:
                a = a.view(np.dtype("i%d" % dt.itemsize))          # two's complement: the same bits either way
            else:
                flip = np.array(1 << (8 * dt.itemsize - 1), dtype=dt)
                a = (a ^ flip).view(np.dtype("i%d" % dt.itemsize))  # unsigned order -> signed order
        t = torch.from_numpy(a.copy())
        self.dist.all_reduce(t, op={RED_SUM: self.dist.ReduceOp.SUM, RED_MIN: self.dist.ReduceOp.MIN, RED_MAX: self.dist.ReduceOp.MAX}[op])
        if root < 0 or root == self.rank:
            r = t.numpy()
            if flip is not None:
                r = r.view(dt) ^ flip
            self.mem.write(buf, r)

    # -- all-to-all-v over several columns
    def _alltoallv(self, ctx, ncols, send, recv, esize, send_off, recv_off, stream):
        return self._guard(self.alltoallv, ncols, send, recv, esize, send_off, recv_off, stream)

    def alltoallv(self, ncols, send, recv, esize, send_off, recv_off, stream):
        torch = self.torch
        self.calls["alltoallv"] += 1
        self.mem.sync(stream)
        W = self.world
        so = [int(send_off[p]) for p in range(W + 1)]
        ro = [int(recv_off[p]) for p in range(W + 1)]
        for c in range(ncols):
            es = int(esize[c])
            src = self.mem.read(send[c], so[W] * es) if so[W] else np.empty(0, dtype=np.uint8)
            ins = [torch.from_numpy(src[so[p] * es:so[p + 1] * es].copy()) for p in range(W)]
            outs = [torch.empty((ro[p + 1] - ro[p]) * es, dtype=torch.uint8) for p in range(W)]
            self.dist.all_to_all(outs, ins) if self.dist.get_backend() != "gloo" else self._a2a_gloo(outs, ins)
            if ro[W]:
                self.mem.write(recv[c], torch.cat(outs).numpy())

    def _a2a_gloo(self, outs, ins):
        """gloo has no all_to_all on every build: pairwise isend / irecv."""
        dist, reqs = self.dist, []
        outs[self.rank].copy_(ins[self.rank])
        for p in range(self.world):
            if p == self.rank:
                continue
            if ins[p].numel():
                reqs.append(dist.isend(ins[p], p))
            if outs[p].numel():
                reqs.append(dist.irecv(outs[p], p))
        for r in reqs:
            r.wait()


class Comm:
    """vh_comm handle: RCCL (`Comm.rccl(dist)`) or the gloo callback transport (`Comm.gloo(dist)`)."""

    def __init__(self, handle, rank, world, transport=None):
        self.handle, self.rank, self.world, self.transport = handle, rank, world, transport

    @classmethod
    def rccl(cls, dist):
        """One RCCL communicator inside the library; the unique id travels through torch.distributed (any backend)."""
        lib = capi.load()
        rank, world = dist.get_rank(), dist.get_world_size()
        ident = (C.c_char * capi.COMM_ID_BYTES)()
        box = [None]
        if rank == 0:
            try:
                capi.check(lib.vh_comm_unique_id(ident))
                box = [bytes(ident)]
            except capi.VhError as e:          # (no RCCL here): tell the others instead of leaving them in the broadcast
                box = [e]
        dist.broadcast_object_list(box, src=0)
        if isinstance(box[0], Exception):
            raise box[0]
        ident = (C.c_char * capi.COMM_ID_BYTES).from_buffer_copy(box[0])
        h = C.c_void_p()
        capi.check(lib.vh_comm_init(ident, rank, world, C.byref(h)))
        return cls(h, rank, world)

    @classmethod
    def gloo(cls, dist):
        lib = capi.load()
        tr = GlooTransport(dist)
        h = C.c_void_p()
        capi.check(lib.vh_comm_init_custom(C.byref(tr.ops), tr.rank, tr.world, C.byref(h)))
        return cls(h, tr.rank, tr.world, tr)

    def info(self) -> dict:
        """The communicator as the transport reports it (vh_comm_info): rank count, this rank, device, PCI bus id."""
        ci = capi.CommInfo()
        capi.check(capi.load().vh_comm_info(self.handle, C.byref(ci)))
        return {"transport": "rccl" if ci.transport == capi.COMM_RCCL else "callbacks", "nranks": int(ci.nranks), "rank": int(ci.rank),
                "device": int(ci.device), "pci_bus_id": ci.pci_bus_id.decode("ascii", "replace")}

    def close(self):
        if self.handle:
            capi.load().vh_comm_destroy(self.handle)
            self.handle = None


def sharded_query(table, plan, comm: Comm, root: int = 0, copy: bool = True):
    """One query over a table sharded across comm.world ranks (vh_query_agg_sharded). The merged groups land on `root`
    (the other ranks get an AggResult without rows, with global counters); root = -1 leaves sparse results with their owners."""
    p, keep = table._build_plan(plan)
    res = C.c_void_p()
    rc = table.lib.vh_query_agg_sharded(table.handle, C.byref(p), comm.handle, int(root), C.byref(res))
    if rc != 0:
        extra = "; transport: %s" % comm.transport.errors[-1] if comm.transport is not None and comm.transport.errors else ""
        raise capi.VhError(rc, table.lib.vh_last_error().decode("utf-8", "replace") + extra)
    try:
        return table._collect(res, plan, copy)
    finally:
        table.lib.vh_result_free(res)

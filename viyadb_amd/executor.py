"""Thin Python handle over the C-ABI (viyadb_amd.capi): a device-resident table mirror
and the aggregate call.  Used by bench.py, the parity tests and the smoke test; the
production host shim is the C++ one in viyadb_amd/host/ (same C-ABI underneath).

Mirrors, in shape, what the reference's generated ``viya_query_agg`` consumes and
produces (src/codegen/query/agg_query.cc:26-75): a table of SoA segments in, a set of
(group key, metric state) rows + QueryStats counters out.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import capi
from .capi import VhError  # noqa: F401  (re-export)


def init(device: int = 0, stream: Optional[int] = None) -> None:
    lib = capi.load()
    capi.check(lib.vh_init(device))
    if stream is not None:
        capi.check(lib.vh_set_stream(C.c_void_p(stream)))


def set_stream(stream: Optional[int]) -> None:
    """stream: a hipStream_t as int (0 = the legacy default stream); None = the library's own stream."""
    capi.check(capi.load().vh_set_stream(C.c_void_p(-1 & 0xFFFFFFFFFFFFFFFF) if stream is None else C.c_void_p(stream)))


def anynum(elem: int, value) -> capi.AnyNum:
    """Pack a literal the way db::AnyNum does: the column's own type in the low bytes."""
    a = capi.AnyNum()
    a.u64 = 0
    raw = np.array([value]).astype(capi.ELEM_NP[elem]).tobytes()
    C.memmove(C.byref(a), raw, len(raw))
    return a


@dataclass
class GroupSpec:
    col: int
    granularity: int = capi.T_NONE
    rollup: Sequence = ()          # [(unit, before_ts), ...] in the reference's rule order
    micro: bool = False
    cardinality: int = 0


@dataclass
class AggPlan:
    """Postfix filter + group columns + metric columns (see include/viya_hip.h)."""
    filter: Sequence = ()          # ("rel", col, op, value) | ("in", col, equal, [values]) | ("and"|"or", n) | ("true",)
    groups: Sequence[GroupSpec] = ()
    metrics: Sequence[int] = ()
    seg_rows: Optional[Sequence[int]] = None
    flags: int = 0
    groups_hint: int = 0
    having: Sequence = ()          # same node tuples; `col` = result column (group i, or len(groups) + metric j)
    top: Optional[tuple] = None    # (result column, descending, k): device top-N superset (see include/viya_hip.h)


@dataclass
class AggResult:
    keys: List[np.ndarray]
    states: List[np.ndarray]
    hidden_count: Optional[np.ndarray]
    ngroups: int
    scanned_recs: int
    scanned_segments: int
    passed_recs: int
    path: str
    scan_kernel_ms: float
    total_ms: float
    algorithmic_bytes: int
    retries: int
    fast: bool = False
    lanes: bool = False            # the no-compaction variant of the fast kernel ran
    packed: bool = False           # group / metric values were gathered from a payload projection (vh_table_pack)
    returned: int = 0              # rows delivered (= ngroups unless a HAVING was pushed down)
    kernel: str = ""               # symbol(s) of the scan kernel(s) that ran, as rocprofv3 prints them
    narrow: bool = False           # predicate columns were streamed from 8- / 16-bit copies (vh_table_narrow)
    jit: bool = False              # a scan kernel compiled for this plan shape ran (viyadb_amd/csrc/vh_jit.hip)
    hpart: bool = False            # hashed partitioning of the hash path ran (vh_hpart.h)
    packed_compressed: bool = False  # ... from compressed records (integers stored at the width their values need)
    hp_packed: bool = False        # hashed partitioning with packed 16-byte tuples (a count-distinct's ids inside the payload word)
    predpack: bool = False         # predicate columns were streamed as byte planes of a bit-packed predicate projection (vh_table_predpack)
    sliced: bool = False           # ... of its bit-sliced form (comparisons bit-serial on 32 rows per lane)
    streamed_payload: bool = False  # ... and the payload records streamed beside them, a survivor's record queued in its row's place (no gathers)
    flags: int = 0                 # vh_result_info.reserved as it came



class DeviceTable:
    """vh_table: the HBM mirror of a db::Table's segment store."""

    def __init__(self, cols: Sequence, segment_rows: int, reserve_segments: int = 1):
        """cols: [(kind, elem), ...] in table order (dimensions, metrics[, hidden count])."""
        self.lib = capi.load()
        self.cols = [(int(k), int(e)) for k, e in cols]
        self.segment_rows = int(segment_rows)
        arr = (capi.ColDesc * len(self.cols))(*[capi.ColDesc(k, e) for k, e in self.cols])
        h = C.c_void_p()
        capi.check(self.lib.vh_table_create(arr, len(self.cols), self.segment_rows, int(reserve_segments), C.byref(h)))
        self.handle = h
        # a vh_result owns its execution context (stream, device scratch, pinned staging) until vh_result_free: threads
        # that share a DeviceTable need no lock of their own. copy=False views alias the context's staging buffer and
        # outlive the free only until the second-next query that takes the same context: single-threaded callers only.

    def close(self):
        if self.handle:
            self.lib.vh_table_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- data in
    def sync_segment(self, seg: int, columns: Sequence[Optional[np.ndarray]], nrows: Optional[int] = None):
        ptrs = (C.c_void_p * len(self.cols))()
        keep = []
        n = nrows
        for i, a in enumerate(columns):
            if a is None or self.cols[i][1] >= capi.BITSET32:
                ptrs[i] = None
                continue
            a = np.ascontiguousarray(a, dtype=capi.ELEM_NP[self.cols[i][1]])
            keep.append(a)
            ptrs[i] = a.ctypes.data
            if n is None:
                n = len(a)
        capi.check(self.lib.vh_segment_sync(self.handle, seg, int(n or 0), ptrs))

    def sync_batch(self, items: Sequence[Tuple]):
        """vh_table_sync_batch. items: (seg, row_first, nrows, new_size, columns, flags) — `columns` are the segment's FULL column arrays
        (numpy, table order; None = leave alone), exactly what vh_segment_sync takes; they must stay alive until the next query or sync."""
        arr = (capi.SyncItem * len(items))()
        keep = []
        for k, (seg, row_first, nrows, new_size, columns, flags) in enumerate(items):
            ptrs = (C.c_void_p * len(self.cols))()
            for i, a in enumerate(columns):
                if a is None or self.cols[i][1] >= capi.BITSET32:
                    ptrs[i] = None
                    continue
                assert a.dtype == np.dtype(capi.ELEM_NP[self.cols[i][1]]) and a.flags["C_CONTIGUOUS"], (i, a.dtype)
                ptrs[i] = a.ctypes.data
            keep.append((ptrs, columns))
            arr[k] = capi.SyncItem(int(seg), int(flags), int(row_first), int(nrows), int(new_size), ptrs)
        capi.check(self.lib.vh_table_sync_batch(self.handle, arr, len(items)))
        self._sync_keep = keep

    def sync_stats(self):
        v = [C.c_uint64() for _ in range(5)]
        capi.check(self.lib.vh_table_sync_stats(self.handle, *[C.byref(x) for x in v]))
        return dict(zip(("batches", "runs", "bytes_pulled", "bytes_staged", "bytes_dma"), [x.value for x in v]))

    def sync_bitset(self, seg: int, col: int, offsets: np.ndarray, values: np.ndarray):
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        vdt = np.uint32 if self.cols[col][1] == capi.BITSET32 else np.uint64
        values = np.ascontiguousarray(values, dtype=vdt)
        capi.check(self.lib.vh_segment_sync_bitset(self.handle, seg, col, len(offsets) - 1,
                                                   offsets.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                   values.ctypes.data if len(values) else None))

    def generate(self, seg_first: int, nseg: int, rows_per_seg: int, row_base: int, specs: Sequence, seed: int):
        """specs: per column (mode, mod, add, scale[, param])."""
        arr = (capi.GenSpec * len(self.cols))()
        for i, sp in enumerate(specs):
            mode, mod, add, scale = sp[:4]
            arr[i] = capi.GenSpec(int(mode), int(sp[4]) if len(sp) > 4 else 0, int(mod), int(add), float(scale))
        capi.check(self.lib.vh_segment_generate(self.handle, seg_first, nseg, rows_per_seg, row_base, arr, seed))

    def read_column(self, seg: int, col: int, nrows: int) -> np.ndarray:
        out = np.empty(nrows, dtype=capi.ELEM_NP[self.cols[col][1]])
        capi.check(self.lib.vh_segment_read(self.handle, seg, col, nrows, out.ctypes.data))
        return out

    def info(self):
        nseg, srows, dbytes = C.c_uint32(), C.c_uint64(), C.c_uint64()
        capi.check(self.lib.vh_table_info(self.handle, C.byref(nseg), C.byref(srows), C.byref(dbytes)))
        return nseg.value, srows.value, dbytes.value

    def segment_stats(self, seg: int, col: int):
        lo, hi = capi.AnyNum(), capi.AnyNum()
        capi.check(self.lib.vh_segment_stats(self.handle, seg, col, C.byref(lo), C.byref(hi)))
        dt = np.dtype(capi.ELEM_NP[self.cols[col][1]])
        f = lambda a: np.frombuffer(bytes(a), dtype=dt, count=1)[0]
        return f(lo), f(hi)

    def pack(self, cols: Sequence[int], compressed: Optional[bool] = None) -> None:
        """Payload projection over `cols` (vh_table_pack): selective queries gather these columns from one record per row.
        compressed: None = the library decides, False = plain records, True = integers at the width their values need."""
        arr = (C.c_int32 * len(cols))(*[int(c) for c in cols])
        capi.check(self.lib.vh_table_pack_ex(self.handle, arr, len(cols), 0 if compressed is None else 2 if compressed else 1))

    def unpack(self) -> None:
        capi.check(self.lib.vh_table_unpack(self.handle))

    def relocate(self, which: int = 0) -> None:
        """vh_table_relocate: the derived layouts (1 projections, 2 predicate planes, 0 both) copied to fresh device memory."""
        capi.check(self.lib.vh_table_relocate(self.handle, which))

    def narrow(self, cols) -> None:
        """Keep 8- / 16-bit copies of the unsigned 32-bit columns among `cols` whose values fit (vh_table_narrow): what the
        register-resident kernels stream when a query filters on them."""
        cols = sorted(set(int(c) for c in cols))
        if cols:
            arr = (C.c_int32 * len(cols))(*cols)
            capi.check(self.lib.vh_table_narrow(self.handle, arr, len(cols)))

    def predpack(self, cols, sliced: Optional[bool] = None) -> None:
        """A bit-packed predicate projection over `cols` (vh_table_predpack): what the compiled scan streams when a query filters on them.
        sliced: None = the library decides (bit-sliced), False = byte planes, True = one plane per bit."""
        cols = sorted(set(int(c) for c in cols))
        if cols:
            arr = (C.c_int32 * len(cols))(*cols)
            capi.check(self.lib.vh_table_predpack_ex(self.handle, arr, len(cols), 0 if sliced is None else 2 if sliced else 1))

    def filter_columns(self, plan: "AggPlan"):
        """Table columns the plan's filter reads."""
        return sorted({f[1] for f in plan.filter if f[0] in ("rel", "in")})

    def gather_columns(self, plan: AggPlan) -> List[int]:
        """Columns a survivor's values are gathered from: group columns + value metrics (not bitsets, not the row id)."""
        cols = [g.col for g in plan.groups] + [m for m in plan.metrics if m != capi.COL_ROWID and self.cols[m][1] < capi.BITSET32]
        return sorted(set(cols))

    # ---- the hot path
    def prepare(self, plan: AggPlan) -> AggPlan:
        """Build the C structs of `plan` once and keep them on the plan object: repeated queries with the same plan skip
        ~16 us of ctypes assembly each. The plan must not be modified afterwards (build a new AggPlan instead)."""
        plan._prepared = None
        p, keep = self._build_plan(plan)
        plan._prepared = (self, p, keep)
        return plan

    def warm(self, plan: AggPlan) -> int:
        """vh_table_prepare: pay the plan shape's first-use costs now (kernel compile, projection, narrow copies, a measured place for the
        derived layouts). Returns vh_result_info.reserved of the last run: what steady-state queries of this shape run on."""
        p, keep = self._build_plan(plan)
        info = capi.ResultInfo()
        capi.check(self.lib.vh_table_prepare(self.handle, C.byref(p), C.byref(info)))
        return int(info.reserved)

    def _build_plan(self, plan: AggPlan):
        cached = getattr(plan, "_prepared", None)
        if cached is not None and cached[0] is self:
            return cached[1], cached[2]
        keep = []
        nodes, lits = [], []
        hnodes = []
        result_elem = [self.cols[g.col][1] for g in plan.groups] + \
                      [capi.U64 if m == capi.COL_ROWID else
                       (capi.U64 if self.cols[m][1] == capi.BITSET64 else capi.U32) if self.cols[m][1] >= capi.BITSET32 else self.cols[m][1]
                       for m in plan.metrics]
        for f in plan.having:
            k = f[0]
            if k == "true":
                hnodes.append(capi.FilterNode(capi.F_TRUE, 0, 0, 0, 0, 0))
            elif k == "rel":
                _, col, op, val = f
                hnodes.append(capi.FilterNode(capi.F_REL, col, op, 1, len(lits), 0))
                lits.append(val if isinstance(val, capi.AnyNum) else anynum(result_elem[col], val))
            elif k == "in":
                _, col, equal, vals = f
                hnodes.append(capi.FilterNode(capi.F_IN, col, 1 if equal else 0, len(vals), len(lits), 0))
                for v in vals:
                    lits.append(v if isinstance(v, capi.AnyNum) else anynum(result_elem[col], v))
            else:
                hnodes.append(capi.FilterNode(capi.F_AND if k == "and" else capi.F_OR, 0, 0, int(f[1]), 0, 0))
        for f in plan.filter:
            k = f[0]
            if k == "true":
                nodes.append(capi.FilterNode(capi.F_TRUE, 0, 0, 0, 0, 0))
            elif k == "rel":
                _, col, op, val = f
                nodes.append(capi.FilterNode(capi.F_REL, col, op, 1, len(lits), 0))
                lits.append(val if isinstance(val, capi.AnyNum) else anynum(self.cols[col][1], val))
            elif k == "in":
                _, col, equal, vals = f
                nodes.append(capi.FilterNode(capi.F_IN, col, 1 if equal else 0, len(vals), len(lits), 0))
                for v in vals:
                    lits.append(v if isinstance(v, capi.AnyNum) else anynum(self.cols[col][1], v))
            elif k in ("and", "or"):
                nodes.append(capi.FilterNode(capi.F_AND if k == "and" else capi.F_OR, 0, 0, int(f[1]), 0, 0))
            else:
                raise ValueError(f"unknown filter node {f!r}")
        p = capi.Plan()
        if nodes:
            fa = (capi.FilterNode * len(nodes))(*nodes)
            keep.append(fa)
            p.filter = fa
        p.nfilter = len(nodes)
        if lits:
            la = (capi.AnyNum * len(lits))(*lits)
            keep.append(la)
            p.lits = la
        p.nlits = len(lits)
        if plan.groups:
            ga = (capi.GroupCol * len(plan.groups))()
            for i, g in enumerate(plan.groups):
                ga[i].col = g.col
                ga[i].granularity = g.granularity
                ga[i].nrollup = len(g.rollup)
                for k, (unit, before) in enumerate(g.rollup):
                    ga[i].rollup_unit[k] = int(unit)
                    ga[i].rollup_before[k] = int(before)
                ga[i].micro = 1 if g.micro else 0
                ga[i].cardinality = int(g.cardinality)
            keep.append(ga)
            p.groups = ga
        p.ngroups = len(plan.groups)
        if plan.metrics:
            ma = (C.c_int32 * len(plan.metrics))(*[int(m) for m in plan.metrics])
            keep.append(ma)
            p.metrics = ma
        p.nmetrics = len(plan.metrics)
        if plan.seg_rows is not None:
            sa = (C.c_uint64 * max(1, len(plan.seg_rows)))(*[int(x) for x in plan.seg_rows])
            keep.append(sa)
            p.seg_rows = sa
            p.nseg = len(plan.seg_rows)
        p.flags = int(plan.flags)
        p.groups_hint = int(plan.groups_hint)
        if hnodes:
            ha = (capi.FilterNode * len(hnodes))(*hnodes)
            keep.append(ha)
            p.having = ha
        p.nhaving = len(hnodes)
        if plan.top:
            p.top_col, p.top_desc, p.top_k = int(plan.top[0]), 1 if plan.top[1] else 0, int(plan.top[2])
        return p, keep

    def _collect(self, res, plan: AggPlan, copy: bool = True) -> AggResult:
        """copy=False: the arrays alias the library's pinned staging buffer (valid until the second-next
        query on this table); copy=True: private numpy arrays."""
        info = capi.ResultInfo()
        capi.check(self.lib.vh_result_get_info(res, C.byref(info)))
        ng = info.returned_groups
        nk, nm = len(plan.groups), len(plan.metrics)
        kp = (C.c_void_p * max(1, nk))()
        spp = (C.c_void_p * max(1, nm))()
        hp = C.POINTER(C.c_uint64)()
        capi.check(self.lib.vh_result_view(res, kp, spp, C.byref(hp)))

        def view(ptr, dtype):
            dtype = np.dtype(dtype)
            if not ng or not ptr:
                return np.empty(0, dtype=dtype)
            buf = (C.c_char * (ng * dtype.itemsize)).from_address(ptr)
            a = np.frombuffer(buf, dtype=dtype, count=ng)
            return a.copy() if copy else a

        keys = [view(kp[i], capi.ELEM_NP[self.cols[g.col][1]]) for i, g in enumerate(plan.groups)]
        def state_elem(j):          # (the library says what it delivers; < 0: it does not know the metric — never index a list with that)
            e = self.lib.vh_result_state_elem(res, j)
            if e < 0 or e >= len(capi.ELEM_NP):
                raise capi.VhError(e, f"vh_result_state_elem: no element type for metric {j}")
            return capi.ELEM_NP[e]

        states = [view(spp[j], state_elem(j)) for j in range(len(plan.metrics))]
        hidden = view(C.cast(hp, C.c_void_p).value, np.uint64) if info.has_hidden_count else None
        return AggResult(keys, states, hidden, int(info.ngroups), int(info.scanned_recs), int(info.scanned_segments),
                         int(info.passed_recs), capi.PATH_NAMES[info.path], float(info.scan_kernel_ms),
                         float(info.total_ms), int(info.algorithmic_bytes), int(info.retries), bool(info.reserved & 1),
                         bool(info.reserved & 2), bool(info.reserved & 8), int(ng), (self.lib.vh_result_kernel(res) or b"").decode(),
                         bool(info.reserved & 16), bool(info.reserved & 32), bool(info.reserved & 64), bool(info.reserved & 128), bool(info.reserved & 256),
                         bool(info.reserved & 2048), bool(info.reserved & 8192), bool(info.reserved & 4096), int(info.reserved))

    def query_agg(self, plan: AggPlan, copy: bool = True) -> AggResult:
        p, keep = self._build_plan(plan)
        res = C.c_void_p()
        capi.check(self.lib.vh_query_agg(self.handle, C.byref(p), C.byref(res)))
        try:
            return self._collect(res, plan, copy)
        finally:
            self.lib.vh_result_free(res)

    # ---- select: the passing rows themselves, in storage order, through the reference's skip/limit window
    def query_select(self, filter: Sequence, cols: Sequence[int], skip: int = 0, limit: int = 0,
                     seg_rows: Optional[Sequence[int]] = None, flags: int = 0):
        """-> ([one numpy array per selected column], RowsInfo)."""
        p, keep = self._build_plan(AggPlan(filter=filter, seg_rows=seg_rows, flags=flags))
        sp = capi.SelectPlan()
        sp.filter, sp.nfilter, sp.lits, sp.nlits = p.filter, p.nfilter, p.lits, p.nlits
        sp.seg_rows, sp.nseg, sp.flags = p.seg_rows, p.nseg, p.flags
        ca = (C.c_int32 * max(1, len(cols)))(*[int(c) for c in cols])
        sp.cols, sp.ncols = ca, len(cols)
        sp.skip, sp.limit = int(skip), int(limit)
        rows = C.c_void_p()
        capi.check(self.lib.vh_query_select(self.handle, C.byref(sp), C.byref(rows)))
        try:
            info = capi.RowsInfo()
            capi.check(self.lib.vh_rows_get_info(rows, C.byref(info)))
            ptrs = (C.c_void_p * max(1, len(cols)))()
            capi.check(self.lib.vh_rows_view(rows, ptrs))
            out = []
            for i, c in enumerate(cols):
                dt = np.dtype(np.uint64 if self.cols[c][1] >= capi.BITSET32 else capi.ELEM_NP[self.cols[c][1]])
                if info.nrows and ptrs[i]:
                    buf = (C.c_char * (info.nrows * dt.itemsize)).from_address(ptrs[i])
                    out.append(np.frombuffer(buf, dtype=dt, count=info.nrows).copy())
                else:
                    out.append(np.empty(0, dtype=dt))
            return out, info
        finally:
            self.lib.vh_rows_free(rows)

    # ---- split form for multi-GPU: launch -> (caller reduces device buffers) -> finalize
    def query_launch(self, plan: AggPlan):
        p, keep = self._build_plan(plan)
        res = C.c_void_p()
        capi.check(self.lib.vh_query_launch(self.handle, C.byref(p), C.byref(res)))
        return res

    def device_buffers(self, res):
        bufs = (capi.DeviceBuffer * 32)()
        n = C.c_int32()
        capi.check(self.lib.vh_result_device_buffers(res, bufs, 32, C.byref(n)))
        return [(bufs[i].ptr, bufs[i].count, bufs[i].elem, bufs[i].reduce) for i in range(n.value)]

    def finalize(self, res, plan: AggPlan, copy: bool = True) -> AggResult:
        try:
            capi.check(self.lib.vh_result_finalize(res))
            return self._collect(res, plan, copy)
        finally:
            self.lib.vh_result_free(res)

    # ---- hash-path exchange: finalised handle -> rows regrouped by owner -> all-to-all -> merge
    def query_agg_keep(self, plan: AggPlan):
        """vh_query_agg (with its re-plan loop) returning the finalised handle; free it with discard()."""
        p, keep = self._build_plan(plan)
        res = C.c_void_p()
        capi.check(self.lib.vh_query_agg(self.handle, C.byref(p), C.byref(res)))
        return res

    def finalize_keep(self, res) -> None:
        """vh_result_finalize on a launched handle, keeping it (raises VhError if the partial needs a re-plan)."""
        capi.check(self.lib.vh_result_finalize(res))

    def collect(self, res, plan: AggPlan, copy: bool = True) -> AggResult:
        return self._collect(res, plan, copy)

    def partition(self, res, nparts: int):
        """-> (offsets[nparts + 1], [(device ptr, rows, elem, merge op)]) : key columns, metrics, hidden count."""
        offs = (C.c_uint64 * (nparts + 1))()
        bufs = (capi.DeviceBuffer * 48)()
        n = C.c_int32()
        capi.check(self.lib.vh_result_partition(res, nparts, offs, bufs, 48, C.byref(n)))
        return (np.array(list(offs), dtype=np.uint64),
                [(bufs[i].ptr, bufs[i].count, bufs[i].elem, bufs[i].reduce) for i in range(n.value)])

    def partition_pairs(self, res, metric: int, nparts: int):
        """Distinct (group, id) pairs of bitset metric `metric`, regrouped by the group's owner:
        -> (offsets[nparts + 1], [(device ptr, pairs, elem, -1)]) : key columns, then the id column."""
        offs = (C.c_uint64 * (nparts + 1))()
        bufs = (capi.DeviceBuffer * 24)()
        n = C.c_int32()
        capi.check(self.lib.vh_result_partition_pairs(res, int(metric), nparts, offs, bufs, 24, C.byref(n)))
        return (np.array(list(offs), dtype=np.uint64),
                [(bufs[i].ptr, bufs[i].count, bufs[i].elem, bufs[i].reduce) for i in range(n.value)])

    def sync_ids_device(self, seg: int, col: int, nrows: int, ptr: Optional[int]):
        """A bitset column holding ONE id per row, ids at a device address (see vh_segment_sync_ids_device)."""
        capi.check(self.lib.vh_segment_sync_ids_device(self.handle, seg, col, int(nrows), ptr or None))

    def sync_segment_device(self, seg: int, ptrs: Sequence[Optional[int]], nrows: int):
        """vh_segment_sync with DEVICE source addresses (one per column, None = leave untouched)."""
        arr = (C.c_void_p * len(self.cols))()
        for i, ptr in enumerate(ptrs):
            arr[i] = ptr or None
        capi.check(self.lib.vh_segment_sync(self.handle, seg, int(nrows), arr))

    def discard(self, res) -> None:
        """Drop a launched (not finalised) partial result: non-root ranks after the collective."""
        self.lib.vh_result_free(res)


def measure_read_bandwidth(nbytes: int = 4 << 30, iters: int = 5) -> float:
    out = C.c_double()
    capi.check(capi.load().vh_measure_read_bandwidth(int(nbytes), int(iters), C.byref(out)))
    return out.value

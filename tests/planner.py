"""Test-side planner: turn an oracle-parsed aggregate query into the C-ABI plan and mirror an
oracle Table into HBM, so the HIP path and the oracle can be compared at the level of agg_map
contents (keys + metric states) on arbitrary typed data, without going through row strings."""
from __future__ import annotations

import numpy as np

from oracle import viya_oracle as vo
from viyadb_amd import capi
from viyadb_amd.executor import AggPlan, DeviceTable, GroupSpec

_ELEM = {"ubyte": capi.U8, "ushort": capi.U16, "uint": capi.U32, "ulong": capi.U64, "byte": capi.I8, "short": capi.I16,
         "int": capi.I32, "long": capi.I64, "float": capi.F32, "double": capi.F64}
_DIMK = {"string": capi.DIM_STRING, "numeric": capi.DIM_NUMERIC, "time": capi.DIM_TIME, "boolean": capi.DIM_BOOLEAN}
_METK = {"max": capi.METRIC_MAX, "min": capi.METRIC_MIN, "sum": capi.METRIC_SUM, "avg": capi.METRIC_AVG,
         "count": capi.METRIC_COUNT, "bitset": capi.METRIC_BITSET}
_OPS = {"eq": capi.OP_EQ, "ne": capi.OP_NE, "lt": capi.OP_LT, "le": capi.OP_LE, "gt": capi.OP_GT, "ge": capi.OP_GE}
_UNIT = {vo.YEAR: capi.T_YEAR, vo.MONTH: capi.T_MONTH, vo.WEEK: capi.T_WEEK, vo.DAY: capi.T_DAY, vo.HOUR: capi.T_HOUR,
         vo.MINUTE: capi.T_MINUTE, vo.SECOND: capi.T_SECOND}


def storage_index(t: vo.Table, c: vo.Column) -> int:
    return c.index if c.is_dim else len(t.dims) + c.index


def col_descs(t: vo.Table):
    out = [(_DIMK[d.dim_type], _ELEM[d.num_type.name]) for d in t.dims]
    for m in t.metrics:
        if m.agg == "bitset":
            out.append((capi.METRIC_BITSET, capi.BITSET64 if m.num_type.size == 8 else capi.BITSET32))
        else:
            out.append((_METK[m.agg], _ELEM[m.num_type.name]))
    if t.has_hidden_count:
        out.append((capi.METRIC_HIDDEN_COUNT, capi.U64))
    return out


def mirror_table(t: vo.Table, reserve=None) -> DeviceTable:
    dt = DeviceTable(col_descs(t), t.segment_size, reserve_segments=reserve or max(1, len(t.segments)))
    for s, seg in enumerate(t.segments):
        n = seg["size"]
        cols = [a[:n] for a in seg["d"]]
        for m in t.metrics:
            cols.append(None if m.agg == "bitset" else seg["m"][m.index][:n])
        if t.has_hidden_count:
            cols.append(seg["count"][:n])
        dt.sync_segment(s, cols, n)
        for m in t.metrics:
            if m.agg != "bitset":
                continue
            sets = seg["m"][m.index][:n]
            offs = np.zeros(n + 1, dtype=np.uint64)
            offs[1:] = np.cumsum([len(x) for x in sets])
            vals = np.array([v for x in sets for v in sorted(x)], dtype=np.uint64)
            dt.sync_bitset(s, storage_index(t, m), offs, vals)
    return dt


def plan_from_query(t: vo.Table, aq: vo.AggQuery, now=None, flags=0, groups_hint=0, seg_rows=None) -> AggPlan:
    nodes = []

    def lit(col, value):
        return capi_anynum(col, vo.decode_value(t, col, value))

    def walk(f):
        if isinstance(f, vo.Empty):
            nodes.append(("true",))
        elif isinstance(f, vo.Rel):
            col = t.column(f.column)
            nodes.append(("rel", storage_index(t, col), _OPS[f.op], lit(col, f.value)))
        elif isinstance(f, vo.In):
            col = t.column(f.column)
            nodes.append(("in", storage_index(t, col), f.equal, [lit(col, v) for v in f.values]))
        else:
            for c in f.filters:
                walk(c)
            nodes.append((f.op, len(f.filters)))
    walk(aq.filter)
    import time as _t
    now = int(_t.time()) if now is None else now
    groups = []
    for oc in aq.dim_cols:
        d = oc.col
        g = GroupSpec(storage_index(t, d), micro=d.micro)
        if d.dim_type == "time" and (d.rollup_rules or oc.granularity is not None):
            g.rollup = [(_UNIT[r.granularity], b) for r, b in zip(d.rollup_rules, vo.rollup_boundaries(d, now))]
            if oc.granularity is not None:
                g.granularity = _UNIT[oc.granularity]
        if d.dim_type == "string":
            g.cardinality = len(t.dicts[d.name].c2v)
        elif d.dim_type == "boolean":
            g.cardinality = 2
        groups.append(g)
    metrics = [storage_index(t, oc.col) for oc in aq.metric_cols]
    return AggPlan(filter=nodes, groups=groups, metrics=metrics, flags=flags, groups_hint=groups_hint, seg_rows=seg_rows)


def capi_anynum(col: vo.Column, value):
    import ctypes as C
    a = capi.AnyNum()
    a.u64 = 0
    raw = np.array([value]).astype(col.num_type.dtype).tobytes()
    C.memmove(C.byref(a), raw, len(raw))
    return a

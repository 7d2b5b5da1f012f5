"""Synthetic workloads of SURVEY.md §8(d) / BASELINE.json `configs` — data definitions only.

A workload is: a table shape (column kinds/element types + a generator spec per column,
consumed by vh_segment_generate and, identically, by oracle/synth.py) and one aggregate
query in two equivalent forms: the C-ABI plan (executor.AggPlan) and the reference's JSON
descriptor (so the oracle can parse it like any other query).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Tuple

from . import capi
from .executor import AggPlan, GroupSpec

U, ROWID, CONST = capi.GEN_UNIFORM, capi.GEN_ROWID, capi.GEN_CONST
SEED = 42


@dataclass
class SynthColumn:
    name: str
    kind: int
    elem: int
    gen: Tuple[int, int, int, float]      # (mode, mod, add, scale)
    json_type: str                        # the reference's column type string


@dataclass
class Workload:
    name: str
    description: str
    columns: List[SynthColumn]
    segment_rows: int
    plan: AggPlan
    query: dict                           # reference JSON descriptor of the same query
    bytes_per_row_referenced: int         # B_ref per row
    table_bytes_per_row: int

    def table_json(self, name="synth"):
        dims = [{"name": c.name, "type": c.json_type} for c in self.columns if c.kind < 16]
        for d in dims:
            if d["type"] == "time" and getattr(self, "rollup_rules", None):
                d["rollup_rules"] = self.rollup_rules
        mets = [{"name": c.name, "type": c.json_type} for c in self.columns if c.kind >= 16]
        return {"name": name, "segment_size": self.segment_rows, "dimensions": dims, "metrics": mets}

    def col(self, name: str) -> int:
        return [c.name for c in self.columns].index(name)


def _dim(name, mod, add=0):
    return SynthColumn(name, capi.DIM_NUMERIC, capi.U32, (U, mod, add, 1.0), "uint")


def _rowid(name="id"):
    return SynthColumn(name, capi.DIM_NUMERIC, capi.U32, (ROWID, 1, 0, 1.0), "uint")


def c1(segment_rows=1_000_000) -> Workload:
    """C1: 4 columns, SELECT SUM(v) WHERE k = 7 (plumbing case; the reference's own CPU-runnable shape)."""
    cols = [_dim("k", 1000), _rowid(),
            SynthColumn("v", capi.METRIC_SUM, capi.I64, (U, 1001, 0, 1.0), "long_sum"),
            SynthColumn("count", capi.METRIC_COUNT, capi.U32, (CONST, 1, 1, 1.0), "count")]
    plan = AggPlan(filter=[("rel", 0, capi.OP_EQ, 7)], groups=[], metrics=[2])
    q = {"type": "aggregate", "table": "synth", "dimensions": [], "metrics": ["v"],
         "filter": {"op": "eq", "column": "k", "value": "7"}}
    return Workload("C1", "10M-row/4-col SUM(v) WHERE k=7", cols, segment_rows, plan, q, 4 + 8, 4 + 4 + 8 + 4)


def c2(segment_rows=1_000_000) -> Workload:
    """C2: 8 columns, range filter (50 %) + GROUP BY d0 (1 K groups), SUM m0, SUM m1."""
    cols = [_dim("d0", 1000), _dim("d1", 1_000_000), _dim("d2", 100), _rowid(),
            SynthColumn("m0", capi.METRIC_SUM, capi.I64, (U, 1001, 0, 1.0), "long_sum"),
            SynthColumn("m1", capi.METRIC_SUM, capi.I32, (U, 101, 0, 1.0), "int_sum"),
            SynthColumn("count", capi.METRIC_COUNT, capi.U32, (U, 3, 1, 1.0), "count"),
            SynthColumn("m3", capi.METRIC_MAX, capi.U32, (U, 1_000_000, 0, 1.0), "uint_max")]
    plan = AggPlan(filter=[("rel", 1, capi.OP_GE, 250000), ("rel", 1, capi.OP_LT, 750000), ("and", 2)],
                   groups=[GroupSpec(0)], metrics=[4, 5])
    q = {"type": "aggregate", "table": "synth", "dimensions": ["d0"], "metrics": ["m0", "m1"],
         "filter": {"op": "and", "filters": [{"op": "ge", "column": "d1", "value": "250000"},
                                             {"op": "lt", "column": "d1", "value": "750000"}]}}
    return Workload("C2", "100M-row/8-col range filter + GROUP BY d0 (~1K groups), SUM m0, SUM m1",
                    cols, segment_rows, plan, q, 4 + 4 + 8 + 4, 4 * 4 + 8 + 4 + 4 + 4)


def c3(segment_rows=1_000_000) -> Workload:
    """C3: 12 columns (60 B/row), 3-predicate conjunction (~5 %) + GROUP BY d0,d1 (100 K groups), SUM m0 + COUNT."""
    cols = [_dim("d0", 1000), _dim("d1", 100), _dim("d2", 4), _dim("d3", 1000), _dim("d4", 1000),
            _dim("d5", 1 << 20), _rowid(),
            SynthColumn("m0", capi.METRIC_SUM, capi.I64, (U, 1001, 0, 1.0), "long_sum"),
            SynthColumn("m1", capi.METRIC_MAX, capi.I64, (U, 1_000_000, 0, 1.0), "long_max"),
            SynthColumn("count", capi.METRIC_COUNT, capi.U32, (U, 3, 1, 1.0), "count"),
            SynthColumn("m3", capi.METRIC_SUM, capi.F64, (U, 10000, 0, 0.01), "double_sum"),
            SynthColumn("m4", capi.METRIC_MIN, capi.U32, (U, 1_000_000, 0, 1.0), "uint_min")]
    plan = AggPlan(filter=[("rel", 2, capi.OP_EQ, 1), ("rel", 3, capi.OP_LT, 447), ("rel", 4, capi.OP_GE, 553), ("and", 3)],
                   groups=[GroupSpec(0), GroupSpec(1)], metrics=[7, 9], groups_hint=100_000)
    q = {"type": "aggregate", "table": "synth", "dimensions": ["d0", "d1"], "metrics": ["m0", "count"],
         "filter": {"op": "and", "filters": [{"op": "eq", "column": "d2", "value": "1"},
                                             {"op": "lt", "column": "d3", "value": "447"},
                                             {"op": "ge", "column": "d4", "value": "553"}]}}
    return Workload("C3", "1B-row/12-col 3-predicate filter (~5%) + GROUP BY d0,d1 (~100K groups), SUM m0 + COUNT",
                    cols, segment_rows, plan, q, 3 * 4 + 2 * 4 + 8 + 4, 7 * 4 + 8 + 8 + 4 + 8 + 4)


C5_NOW = 1496570140  # the reference's own rollup test constant (test/time.cc:320)


def _rollup_before(now: int):
    """rollup_bN_i for the rules hour/1 day, day/1 week, month/1 year, in the reference's order
    (`after` descending: month/1y, day/1w, hour/1d): Duration.add_to(now, -1) (src/codegen/db/rollup.cc:44-75)."""
    import calendar
    import time
    tm = time.gmtime(now)
    year_ago = calendar.timegm((tm.tm_year - 1, tm.tm_mon, tm.tm_mday, tm.tm_hour, tm.tm_min, tm.tm_sec))
    return [(capi.T_MONTH, year_ago), (capi.T_DAY, now - 7 * 86400), (capi.T_HOUR, now - 86400)]


def c5t(segment_rows=1_000_000) -> Workload:
    """C5 without the bitset metric: time dimension with rollup rules + query granularity `hour`,
    GROUP BY (t, u) -> millions of sparse groups (hash path), COUNT."""
    two_years = 2 * 365 * 86400
    cols = [SynthColumn("t", capi.DIM_TIME, capi.U32, (U, two_years, C5_NOW - two_years, 1.0), "time"),
            _dim("u", 1_000_000), _rowid(),
            SynthColumn("count", capi.METRIC_COUNT, capi.U32, (U, 3, 1, 1.0), "count")]
    plan = AggPlan(filter=[("rel", 1, capi.OP_LT, 500_000)],
                   groups=[GroupSpec(0, granularity=capi.T_HOUR, rollup=_rollup_before(C5_NOW)), GroupSpec(1)], metrics=[3],
                   groups_hint=0)
    q = {"type": "aggregate", "table": "synth", "select": [{"column": "t", "granularity": "hour"}, {"column": "u"}, {"column": "count"}],
         "filter": {"op": "lt", "column": "u", "value": "500000"}}
    w = Workload("C5t", "time dim + rollup rules + hour granularity, GROUP BY (t,u): sparse keys, hash path, COUNT",
                 cols, segment_rows, plan, q, 4 + 4 + 4, 16)
    w.rollup_rules = [{"granularity": "hour", "after": "1 days"}, {"granularity": "day", "after": "1 weeks"},
                      {"granularity": "month", "after": "1 years"}]
    w.now = C5_NOW
    return w


def c5(segment_rows=1_000_000) -> Workload:
    """C5: C5t plus the count-distinct (bitset) metric: 2 user ids per stored row from [0, 10^7)."""
    w = c5t(segment_rows)
    w.columns = w.columns[:3] + [SynthColumn("users", capi.METRIC_BITSET, capi.BITSET32, (U, 10_000_000, 2, 1.0), "bitset"),
                                 w.columns[3]]
    w.plan = AggPlan(filter=w.plan.filter, groups=w.plan.groups, metrics=[3, 4], groups_hint=0)
    w.query = dict(w.query, select=[{"column": "t", "granularity": "hour"}, {"column": "u"}, {"column": "users"}, {"column": "count"}])
    w.name = "C5"
    w.description = "time dim + rollup rules + hour granularity, GROUP BY (t,u), COUNT DISTINCT users + COUNT (hash path)"
    w.bytes_per_row_referenced = 4 + 4 + 4 + 8 + 2 * 4
    w.table_bytes_per_row = 16 + 8 + 2 * 4
    return w


# ---- skewed shapes of the same queries (tools/skew_probe.py measures them, tests/test_gpu_skew.py checks them against the oracle)
def c3z(segment_rows=1_000_000) -> Workload:
    """C3 with a Zipf-like d0 (VH_GEN_ZIPF: value 0 alone holds a tenth of the rows, the first 7 values half of them): one of DENSE_PART's
    partitions receives several times its share of the tuples; the reference's unordered_map does not care (src/codegen/db/store.cc:67-85)."""
    w = c3(segment_rows)
    w.columns[0] = SynthColumn("d0", capi.DIM_NUMERIC, capi.U32, (capi.GEN_ZIPF, 1000, 0, 1.0), "uint")
    w.name, w.description = "C3z", "C3 with Zipf-like d0 (P(v) ~ 1/(v+1)): " + w.description
    return w


def c3s(segment_rows=1_000_000, total_segments=1000) -> Workload:
    """C3 loaded in d3 order (a time-ordered load, the reference's own scenario: test/index.cc:44-75): d3 is constant per block of segments,
    `d3 < 447` skips 55 % of the segments by their min / max and every survivor lies in the first 447 — clustered, not spread."""
    w = c3(segment_rows)
    rows_per_value = max(1, total_segments * segment_rows // 1000)
    w.columns[3] = SynthColumn("d3", capi.DIM_NUMERIC, capi.U32, (capi.GEN_SORTED, rows_per_value, 0, 1.0), "uint")
    w.name, w.description = "C3s", "C3 loaded in d3 order (%d rows per value): " % rows_per_value + w.description
    return w


def c5h(segment_rows=1_000_000) -> Workload:
    """C5 with one (t, u) pair on a tenth of the rows (VH_GEN_HOT on both columns: the same rows): one group of the ~35 M receives 10 % of the
    tuples — one digit of every level of the hashed partitioning, one LDS range of the aggregation."""
    w = c5(segment_rows)
    t, u = w.columns[0], w.columns[1]
    w.columns[0] = SynthColumn(t.name, t.kind, t.elem, (capi.GEN_HOT,) + tuple(t.gen[1:]) + (100,), t.json_type)
    w.columns[1] = SynthColumn(u.name, u.kind, u.elem, (capi.GEN_HOT, 900_000, 0, 1.0, 100), u.json_type)      # hot u = 450 000: passes `u < 500 000`
    w.name, w.description = "C5h", "C5 with one (t, u) on 10 % of the rows: " + w.description
    return w


WORKLOADS = {"C1": c1, "C2": c2, "C3": c3, "C5t": c5t, "C5": c5, "C3z": c3z, "C3s": c3s, "C5h": c5h}


def create_device_table(w: Workload, nseg: int, rows_per_seg=None, row_base=0, seed=SEED):
    """Build the HBM mirror and fill it with the workload's synthetic rows."""
    from .executor import DeviceTable
    t = DeviceTable([(c.kind, c.elem) for c in w.columns], w.segment_rows, reserve_segments=nseg)
    rps = w.segment_rows if rows_per_seg is None else rows_per_seg
    if nseg:
        t.generate(0, nseg, rps, row_base, [c.gen for c in w.columns], seed)
    return t

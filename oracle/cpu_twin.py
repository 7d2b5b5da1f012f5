"""CPU ORACLE — TEST INFRASTRUCTURE ONLY (also the `cpu_baseline` "port" in bench.py).

The reference executes an aggregate query by generating a C++ function per query shape,
compiling it with `g++ -std=c++17 -O2 -funroll-loops -march=native -shared -fPIC`
(src/codegen/compiler.cc:36-95) and running it on ONE thread over all segments
(src/codegen/query/scan.cc:42-58).  This module is a from-scratch emitter of an equivalent
function — the "CPU twin" of SURVEY.md §7 step 1(b) — so that the CPU number quoted next
to the GPU number is the reference's algorithm compiled the reference's way:

  * SoA segment arrays, scanned sequentially                     (scan.cc:42-58)
  * bitwise & / | predicate on the column's own C++ type         (filter.cc:206-261)
  * std::unordered_map<Dims, Metrics, Hash, KeyEqual>, hash =
    h ^= k + 0x9e3779b9 + (h<<6) + (h>>2) per dimension          (store.cc:67-85)
  * Metrics::Update with +=, std::max, std::min in the metric's own type (store.cc:131-161)
  * util::Time32/Time64 truncation through gmtime_r / timegm     (time.h:91-137)

What differs from the reference's generated text: segments arrive as raw column pointers
instead of a generated `Segment` class (the reference cannot be built here),
segment-skip flags are computed by the caller (viya_oracle.segment_skip), and the result is
copied out of the map into caller arrays instead of being stringified.

Bitset metrics (util::Bitset<N>, src/util/bitset.h:26-67): the reference keeps one CRoaring bitmap PER STORED ROW
(store.cc:255-259), copies the row's bitmap into agg_tuple.m (scan.cc:228-238) and ORs it into the group's (`|=`,
store.cc:131-161); the output is cardinality().  CRoaring is absent here (empty submodule), so Roaring is REPLACED BY an
append-only std::vector<id> per group that is sorted and made unique whenever it has doubled since its last compaction
(amortised O(n log n)) and once more for cardinality(); a row's set arrives as a CSR slice (offsets + ids) instead of a
bitmap object.  The observable — a set cardinality — is implementation-independent; the COST is not: for
the few-ids-per-row sets of C5 a vector append does less work than Roaring's container lookup, array-container insert and
per-row bitmap copy, so this baseline errs on the FAST side of the reference.  A predicate on a bitset metric compares the
row's cardinality (filter.cc:206-261).
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess
import time
from typing import List, Optional

import numpy as np

from . import viya_oracle as vo

HERE = os.path.dirname(os.path.abspath(__file__))
CACHE = os.path.join(HERE, "_twin_cache")
CXXFLAGS = ["-std=c++17", "-O2", "-funroll-loops", "-march=native", "-shared", "-fPIC"]  # compiler.cc:64-72

_CPP = {"byte": "int8_t", "ubyte": "uint8_t", "short": "int16_t", "ushort": "uint16_t", "int": "int32_t",
        "uint": "uint32_t", "long": "int64_t", "ulong": "uint64_t", "float": "float", "double": "double"}
_CPP_MIN = {"byte": "INT8_MIN", "ubyte": "0U", "short": "INT16_MIN", "ushort": "0U", "int": "INT32_MIN", "uint": "0U",
            "long": "INT64_MIN", "ulong": "0UL", "float": "FLT_MIN", "double": "DBL_MIN"}
_CPP_MAX = {"byte": "INT8_MAX", "ubyte": "UINT8_MAX", "short": "INT16_MAX", "ushort": "UINT16_MAX", "int": "INT32_MAX",
            "uint": "UINT32_MAX", "long": "INT64_MAX", "ulong": "UINT64_MAX", "float": "FLT_MAX", "double": "DBL_MAX"}
_OPSTR = {"eq": "==", "ne": "!=", "lt": "<", "le": "<=", "gt": ">", "ge": ">="}

_TIME_CLASSES = r"""
template <int U> static inline void trunc_tm(std::tm& tm) {
  if (U <= 5) tm.tm_sec = 0;
  if (U <= 4) tm.tm_min = 0;
  if (U <= 3) tm.tm_hour = 0;
  if (U <= 1) tm.tm_mday = 1;
  if (U <= 0) tm.tm_mon = 0;
}
struct Time32 {
  std::tm tm_{};
  void set_ts(uint32_t ts) { time_t t = (time_t)ts; gmtime_r(&t, &tm_); }
  uint32_t get_ts() { return timegm(&tm_); }
  template <int U> void trunc() { trunc_tm<U>(tm_); }
};
struct Time64 {
  uint32_t micros_ = 0; std::tm tm_{};
  void set_ts(uint64_t ts) { micros_ = ts % 1000000L; time_t t = (time_t)(ts / 1000000L); gmtime_r(&t, &tm_); }
  uint64_t get_ts() { return timegm(&tm_) * 1000000L + micros_; }
  template <int U> void trunc() { trunc_tm<U>(tm_); micros_ = 0; }
};
"""


_IDSET = r"""
template <typename T> struct IdSet {            // util::Bitset<N> with Roaring replaced by a lazily compacted vector (module docstring)
  std::vector<T> v; size_t limit = 16;
  const T* rp = nullptr; uint32_t rn = 0;       // as a ROW's value: the CSR slice that holds the row's ids
  void compact() { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); }
  IdSet& operator|=(const IdSet& o) {           // group |= row
    v.insert(v.end(), o.rp, o.rp + o.rn);
    if (v.size() >= limit) { compact(); limit = std::max<size_t>(16, 2 * v.size()); }
    return *this;
  }
  uint64_t cardinality() { compact(); return v.size(); }
};
template <typename T> static inline uint64_t row_cardinality(const T* p, uint64_t n) {   // distinct ids of one row (a handful)
  uint64_t c = 0;
  for (uint64_t i = 0; i < n; ++i) { bool seen = false; for (uint64_t j = 0; j < i; ++j) seen |= p[j] == p[i]; c += !seen; }
  return c;
}
"""


class CsrSets:
    """A bitset column of one segment as CSR — what the twin reads (and what the device mirror holds): ids of row i are
    values[offsets[i]:offsets[i + 1]]. viya_oracle keeps Python sets per row; segments built for the twin alone
    (tests.parity.build_oracle_table(..., csr=True)) carry this instead, without a million set objects per segment."""

    def __init__(self, offsets: np.ndarray, values: np.ndarray):
        self.offsets, self.values = np.ascontiguousarray(offsets, dtype=np.uint64), np.ascontiguousarray(values)

    @classmethod
    def from_sets(cls, sets, size: int, dtype):
        lens = np.fromiter((len(x) for x in sets[:size]), dtype=np.uint64, count=size)
        off = np.zeros(size + 1, dtype=np.uint64)
        np.cumsum(lens, out=off[1:])
        vals = np.fromiter((v for x in sets[:size] for v in sorted(x)), dtype=dtype, count=int(off[-1]))
        return cls(off, vals)


def _cmp_expr(table, f, args: list, getter) -> str:
    """ComparisonBuilder text; appends decoded literals to `args` in unpack order."""
    if isinstance(f, vo.Empty):
        return "true"
    if isinstance(f, vo.Rel):
        col = table.column(f.column)
        args.append((col, vo.decode_value(table, col, f.value)))
        return "(%s%sfarg%d)" % (getter(col), _OPSTR[f.op], len(args) - 1)
    if isinstance(f, vo.In):
        col = table.column(f.column)
        parts = []
        for v in f.values:
            args.append((col, vo.decode_value(table, col, v)))
            parts.append("(%s%sfarg%d)" % (getter(col), "==" if f.equal else "!=", len(args) - 1))
        return "(" + ("|" if f.equal else "&").join(parts) + ")"
    op = " & " if f.op == "and" else " | "
    return "(" + op.join(_cmp_expr(table, c, args, getter) for c in f.filters) + ")"


def emit_source(table: vo.Table, aq: vo.AggQuery) -> (str, list):
    dims = [oc.col for oc in aq.dim_cols]
    mets = [oc.col for oc in aq.metric_cols]
    has_avg = any(m.agg == "avg" for m in mets)
    has_count = any(m.agg == "count" for m in mets)
    hidden = has_avg and not has_count
    s = []
    s.append("#include <unordered_map>\n#include <cstdint>\n#include <cstddef>\n#include <cfloat>\n#include <ctime>\n"
             "#include <algorithm>\n#include <chrono>\n#include <functional>\n#include <vector>\n")
    s.append(_TIME_CLASSES)
    s.append(_IDSET)
    s.append("struct AggTuple {\n struct Dimensions {\n")
    for d in dims:
        s.append("  %s _%d;\n" % (_CPP[d.num_type.name], d.index))
    s.append("  struct KeyEqual { bool operator()(const Dimensions &d1,const Dimensions &d2) const { return ")
    s.append(" && ".join("d1._%d==d2._%d" % (d.index, d.index) for d in dims) if dims else "true")
    s.append("; } };\n  struct Hash { std::size_t operator()(const Dimensions &k) const { size_t h = 0L;\n")
    for d in dims:
        v = ("std::hash<%s>{} (k._%d)" % (_CPP[d.num_type.name], d.index)) if (d.dim_type == "numeric" and d.num_type.fp) else "k._%d" % d.index
        s.append("    h ^= %s + 0x9e3779b9 + (h<<6) + (h>>2);\n" % v)
    s.append("    return h; } };\n };\n struct Metrics {\n")
    for m in mets:
        if m.agg == "bitset":
            s.append("  IdSet<%s> _%d;\n" % (_CPP[m.num_type.name], m.index))
            continue
        init = _CPP_MIN[m.num_type.name] if m.agg == "max" else _CPP_MAX[m.num_type.name] if m.agg == "min" else "0"
        s.append("  %s _%d = %s;\n" % (_CPP[m.num_type.name], m.index, init))
    if hidden:
        s.append("  uint64_t _count=0;\n")
    s.append("  void Update(const Metrics &metrics) {\n")
    for m in mets:
        if m.agg == "bitset":
            s.append("   _%d |= metrics._%d;\n" % (m.index, m.index))
        elif m.agg in ("sum", "avg", "count"):
            s.append("   _%d += metrics._%d;\n" % (m.index, m.index))
        elif m.agg == "max":
            s.append("   _%d = std::max(_%d, metrics._%d);\n" % (m.index, m.index, m.index))
        else:
            s.append("   _%d = std::min(_%d, metrics._%d);\n" % (m.index, m.index, m.index))
    if hidden:
        s.append("   _count += metrics._count;\n")
    s.append("  }\n };\n Dimensions d; Metrics m;\n};\n")
    s.append("typedef std::unordered_map<AggTuple::Dimensions,AggTuple::Metrics,AggTuple::Dimensions::Hash,"
             "AggTuple::Dimensions::KeyEqual> AggMap;\nstatic AggMap* g_map = nullptr;\n")
    s.append('extern "C" int64_t twin_run(const void* const* const* seg_d, const void* const* const* seg_m, '
             "const uint64_t* const* seg_cnt, const uint64_t* const* const* seg_off, const uint64_t* seg_size, const uint8_t* seg_process, uint64_t nseg, "
             "const uint64_t* fargs, const uint64_t* rollup_b, double* seconds) {\n")
    s.append(" auto t0 = std::chrono::steady_clock::now();\n delete g_map; g_map = new AggMap();\n AggMap& agg_map = *g_map;\n AggTuple agg_tuple;\n")
    args: list = []
    def getter(c):
        if c.is_dim:
            return "tuple_dims_%d[tuple_idx]" % c.index
        if c.agg == "bitset":      # the predicate sees .cardinality() of the row's set
            return "((%s)row_cardinality(tuple_metrics_%d + tuple_off_%d[tuple_idx], tuple_off_%d[tuple_idx + 1] - tuple_off_%d[tuple_idx]))" % (
                _CPP[c.num_type.name], c.index, c.index, c.index, c.index)
        return "tuple_metrics_%d[tuple_idx]" % c.index
    cmp = _cmp_expr(table, aq.filter, args, getter)
    for i, (col, _) in enumerate(args):
        t = _CPP[col.num_type.name]
        s.append(" %s farg%d = *reinterpret_cast<const %s*>(&fargs[%d]);\n" % (t, i, t, i))
    # rollup definitions
    rb = 0
    roll = {}
    for oc in aq.dim_cols:
        d = oc.col
        if d.dim_type == "time" and (d.rollup_rules or oc.granularity is not None):
            s.append(" Time%d time%d;\n" % (d.num_type.size * 8, d.index))
            roll[d.index] = (rb, oc)
            for k in range(len(d.rollup_rules)):
                s.append(" %s rollup_b%d_%d = (%s)rollup_b[%d];\n" % (_CPP[d.num_type.name], d.index, k, _CPP[d.num_type.name], rb + k))
            rb += len(d.rollup_rules)
    used = {c.index for c in dims} | {a[0].index for a in args if a[0].is_dim}
    usedm = {c.index for c in mets} | {a[0].index for a in args if not a[0].is_dim}
    s.append(" for (uint64_t s = 0; s < nseg; ++s) {\n  auto segment_size = seg_size[s];\n  if (!seg_process[s]) continue;\n")
    for d in table.dims:
        if d.index in used:
            s.append("  const %s* __restrict__ tuple_dims_%d = (const %s*)seg_d[s][%d];\n" % (_CPP[d.num_type.name], d.index, _CPP[d.num_type.name], d.index))
    for m in table.metrics:
        if m.index in usedm:
            s.append("  const %s* __restrict__ tuple_metrics_%d = (const %s*)seg_m[s][%d];\n" % (_CPP[m.num_type.name], m.index, _CPP[m.num_type.name], m.index))
            if m.agg == "bitset":
                s.append("  const uint64_t* __restrict__ tuple_off_%d = seg_off[s][%d];\n" % (m.index, m.index))
    if hidden:
        s.append("  const uint64_t* __restrict__ tuple_count = seg_cnt[s];\n")
    s.append("  for (size_t tuple_idx = 0; tuple_idx < segment_size; ++tuple_idx) {\n   auto r = %s;\n   if (r) {\n" % cmp)
    for oc in aq.dim_cols:
        d = oc.col
        if d.index in roll:
            s.append("    time%d.set_ts(tuple_dims_%d[tuple_idx]);\n" % (d.index, d.index))
            for k, rule in enumerate(d.rollup_rules):
                s.append("    %sif (tuple_dims_%d[tuple_idx] < rollup_b%d_%d) { time%d.trunc<%d>(); }\n" %
                         ("else " if k else "", d.index, d.index, k, d.index, _unit_code(rule.granularity)))
            if oc.granularity is not None:
                s.append("    time%d.trunc<%d>();\n" % (d.index, _unit_code(oc.granularity)))
            s.append("    agg_tuple.d._%d = time%d.get_ts();\n" % (d.index, d.index))
        else:
            s.append("    agg_tuple.d._%d = tuple_dims_%d[tuple_idx];\n" % (d.index, d.index))
    for m in mets:
        if m.agg == "bitset":
            s.append("    agg_tuple.m._%d.rp = tuple_metrics_%d + tuple_off_%d[tuple_idx]; agg_tuple.m._%d.rn = (uint32_t)(tuple_off_%d[tuple_idx + 1] - tuple_off_%d[tuple_idx]);\n"
                     % (m.index, m.index, m.index, m.index, m.index, m.index))
            continue
        s.append("    agg_tuple.m._%d = tuple_metrics_%d[tuple_idx];\n" % (m.index, m.index))
    if hidden:
        s.append("    agg_tuple.m._count = tuple_count[tuple_idx];\n")
    s.append("    agg_map[agg_tuple.d].Update(agg_tuple.m);\n   }\n  }\n }\n")
    s.append(" *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();\n return (int64_t)agg_map.size();\n}\n")
    s.append('extern "C" void twin_fetch(void* const* keys, void* const* states, uint64_t* hidden) {\n size_t i = 0;\n for (auto& kv : *g_map) {\n')
    for k, d in enumerate(dims):
        s.append("  ((%s*)keys[%d])[i] = kv.first._%d;\n" % (_CPP[d.num_type.name], k, d.index))
    for k, m in enumerate(mets):
        if m.agg == "bitset":
            s.append("  ((uint64_t*)states[%d])[i] = kv.second._%d.cardinality();\n" % (k, m.index))
            continue
        s.append("  ((%s*)states[%d])[i] = kv.second._%d;\n" % (_CPP[m.num_type.name], k, m.index))
    if hidden:
        s.append("  hidden[i] = kv.second._count;\n")
    s.append("  ++i;\n }\n delete g_map; g_map = nullptr;\n}\n")
    return "".join(s), args


def _unit_code(unit: int) -> int:
    # trunc_tm<U>: 0 year, 1 month, 3 day, 4 hour, 5 minute, 6 second (same numbering as util::TimeUnit)
    if unit == vo.WEEK:
        raise vo.Unsupported("week")
    return unit


def compile_source(src: str) -> str:
    os.makedirs(CACHE, exist_ok=True)
    h = hashlib.sha1((src + " ".join(CXXFLAGS)).encode()).hexdigest()[:20]
    so = os.path.join(CACHE, h + ".so")
    if not os.path.exists(so):
        tmp = so + ".%d.tmp" % os.getpid()
        subprocess.run(["g++"] + CXXFLAGS + ["-x", "c++", "-", "-o", tmp], input=src.encode(), check=True)
        os.replace(tmp, so)
    return so


class Twin:
    """A compiled per-query CPU function over an oracle Table."""

    def __init__(self, table: vo.Table, q: dict):
        self.table = table
        self.aq = vo.parse_query(table, q)
        t0 = time.time()
        self.src, self.args = emit_source(table, self.aq)
        self.lib = C.CDLL(compile_source(self.src))
        self.compile_seconds = time.time() - t0
        self.lib.twin_run.restype = C.c_int64
        self.hidden = any(oc.col.agg == "avg" for oc in self.aq.metric_cols) and not any(oc.col.agg == "count" for oc in self.aq.metric_cols)

    def _csr(self, i: int, seg: dict) -> dict:
        """metric index -> CsrSets for the bitset columns the query touches (built from the oracle's per-row sets unless the segment
        already carries CSR)."""
        if not hasattr(self, "_csr_cache"):
            self._csr_cache = {}
        if i not in self._csr_cache:
            out = {}
            for m in self.table.metrics:
                if m.agg != "bitset":
                    continue
                col = seg["m"][m.index]
                out[m.index] = col if isinstance(col, CsrSets) else CsrSets.from_sets(col, seg["size"], m.num_type.dtype)
            self._csr_cache[i] = out
        return self._csr_cache[i]

    def run(self, now: Optional[int] = None, seg_rows: Optional[List[int]] = None) -> vo.AggState:
        t = self.table
        nseg = len(t.segments)
        VP = C.c_void_p
        keep = []
        seg_d = (C.POINTER(VP) * max(nseg, 1))()
        seg_m = (C.POINTER(VP) * max(nseg, 1))()
        seg_c = (C.POINTER(C.c_uint64) * max(nseg, 1))()
        seg_o = (C.POINTER(C.POINTER(C.c_uint64)) * max(nseg, 1))()
        sizes = (C.c_uint64 * max(nseg, 1))()
        proc = (C.c_uint8 * max(nseg, 1))()
        st = vo.AggState([], [], None)
        for i, seg in enumerate(t.segments):
            size = seg["size"] if seg_rows is None else int(seg_rows[i])
            sizes[i] = size
            st.scanned_recs += size
            proc[i] = 1 if vo.segment_skip(t, self.aq.filter, seg) else 0
            st.scanned_segments += proc[i]
            da = (VP * max(len(t.dims), 1))(*[a.ctypes.data for a in seg["d"]])
            csr = self._csr(i, seg)                       # bitset columns as CSR (converted once per segment, kept)
            ma = (VP * max(len(t.metrics), 1))(*[(a.ctypes.data if isinstance(a, np.ndarray) else csr[j].values.ctypes.data if j in csr else None)
                                                 for j, a in enumerate(seg["m"])])
            oa = (C.POINTER(C.c_uint64) * max(len(t.metrics), 1))()
            for j, cs in csr.items():
                oa[j] = cs.offsets.ctypes.data_as(C.POINTER(C.c_uint64))
            keep += [da, ma, oa]
            seg_d[i] = da
            seg_m[i] = ma
            seg_o[i] = oa
            if seg["count"] is not None:
                seg_c[i] = seg["count"].ctypes.data_as(C.POINTER(C.c_uint64))
        fargs = (C.c_uint64 * max(len(self.args), 1))()
        for i, (col, lit) in enumerate(self.args):
            raw = np.array([lit]).astype(col.num_type.dtype).tobytes()
            fargs[i] = int.from_bytes(raw.ljust(8, b"\0"), "little")
        rb = []
        if now is None:
            now = int(time.time())
        for oc in self.aq.dim_cols:
            d = oc.col
            if d.dim_type == "time" and (d.rollup_rules or oc.granularity is not None):
                rb += vo.rollup_boundaries(d, now)
        rba = (C.c_uint64 * max(len(rb), 1))(*rb)
        secs = C.c_double()
        n = self.lib.twin_run(seg_d, seg_m, seg_c, seg_o, sizes, proc, C.c_uint64(nseg), fargs, rba, C.byref(secs))
        self.last_seconds = secs.value
        keys = [np.empty(n, dtype=oc.col.num_type.dtype) for oc in self.aq.dim_cols]
        states = [np.empty(n, dtype=np.uint64 if oc.col.agg == "bitset" else oc.col.num_type.dtype) for oc in self.aq.metric_cols]
        hidden = np.empty(n, dtype=np.uint64) if self.hidden else None
        kp = (VP * max(len(keys), 1))(*[k.ctypes.data for k in keys])
        sp = (VP * max(len(states), 1))(*[s.ctypes.data for s in states])
        self.lib.twin_fetch(kp, sp, hidden.ctypes.data_as(C.POINTER(C.c_uint64)) if hidden is not None else None)
        st.keys, st.states, st.hidden_count = keys, states, hidden
        st.passed_recs = -1  # the reference keeps no such counter
        return st

"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product.

A restatement, in Python + numpy, of what ViyaDB's aggregate-query path computes:
table schema -> upsert ingest -> filter -> GROUP-BY aggregate -> post-aggregation.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and only as the checker.  The product path (``viyadb_amd``) never does.

Parity status: PINNED.  The oracle is checked (tests/test_oracle_golden.py) against
every known-answer test the reference's own suite holds for this path, transcribed
as data into tests/golden/reference_cases.json (test/aggregation.cc, filter.cc,
metrics.cc, time.cc, bitset.cc, boolean.cc, index.cc, limits.cc, sort.cc, select.cc, search.cc,
the upsert tests of load.cc, and the query tests of sql.cc as the descriptors parser.y builds: 85 cases), and its
calendar arithmetic against vectors produced by the reference's own
src/util/time.cc compiled in this container (oracle/_ref, tests/golden/time_golden.json).

Each function cites the reference file:line it follows.  The reference generates C++
per query; here the same semantics are interpreted.  Integer arithmetic is done in the
column's own numpy dtype so wrap-around matches the generated C++.
"""
from __future__ import annotations

import functools
import sys
import time as _time
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

UINT8_MAX, UINT16_MAX, UINT32_MAX, UINT64_MAX = 0xFF, 0xFFFF, 0xFFFFFFFF, 0xFFFFFFFFFFFFFFFF
FLT_MIN = float(np.finfo(np.float32).tiny)
FLT_MAX = float(np.finfo(np.float32).max)
DBL_MIN = float(np.finfo(np.float64).tiny)
DBL_MAX = float(np.finfo(np.float64).max)


class InvalidArgument(ValueError):
    """std::invalid_argument in the reference."""


class Unsupported(RuntimeError):
    """Constructs the reference cannot compile / run (generated code would not build)."""


# ---------------------------------------------------------------------------------
# numeric types — src/db/column.cc:98-247
# ---------------------------------------------------------------------------------
NUMERIC_TYPES = {
    # name: (numpy dtype, cpp_min_value, cpp_max_value, parse fn)
    "byte": (np.int8, -128, 127, "stoi"),
    "ubyte": (np.uint8, 0, UINT8_MAX, "stoul"),
    "short": (np.int16, -32768, 32767, "stoi"),
    "ushort": (np.uint16, 0, UINT16_MAX, "stoul"),
    "int": (np.int32, -(2 ** 31), 2 ** 31 - 1, "stoi"),
    "uint": (np.uint32, 0, UINT32_MAX, "stoul"),
    "long": (np.int64, -(2 ** 63), 2 ** 63 - 1, "stoll"),
    "ulong": (np.uint64, 0, UINT64_MAX, "stoull"),
    # cpp_min_value of float/double is FLT_MIN/DBL_MIN: the smallest POSITIVE normal
    # (src/db/column.cc:214-219) — a reference quirk that MAX metrics inherit.
    "float": (np.float32, FLT_MIN, FLT_MAX, "stof"),
    "double": (np.float64, DBL_MIN, DBL_MAX, "stod"),
}


@dataclass
class NumType:
    name: str

    @property
    def dtype(self):
        return np.dtype(NUMERIC_TYPES[self.name][0])

    @property
    def size(self):
        return self.dtype.itemsize

    @property
    def cpp_min_value(self):
        return NUMERIC_TYPES[self.name][1]

    @property
    def cpp_max_value(self):
        return NUMERIC_TYPES[self.name][2]

    @property
    def fp(self):
        return self.name in ("float", "double")

    def parse(self, s: str):
        """NumericType::Parse / UIntType::Parse (column.cc:37-50, 226-251): std::sto* then a C cast."""
        fn = NUMERIC_TYPES[self.name][3]
        if fn in ("stof", "stod"):
            v = _stod(s)
            return self.dtype.type(v)
        v = _stoi_family(s, fn)
        return wrap_int(v, self.dtype)


def uint_type_for_max(max_value: int) -> NumType:
    """max_value_to_uint_type (column.cc:54-62)."""
    m = (max_value - 1) & UINT64_MAX  # size_t arithmetic
    if m < UINT8_MAX:
        return NumType("ubyte")
    if m < UINT16_MAX:
        return NumType("ushort")
    if m < UINT32_MAX:
        return NumType("uint")
    return NumType("ulong")


def wrap_int(v: int, dtype) -> np.generic:
    dtype = np.dtype(dtype)
    bits = dtype.itemsize * 8
    v &= (1 << bits) - 1
    if dtype.kind == "i" and v >= 1 << (bits - 1):
        v -= 1 << bits
    return dtype.type(v)


def _parse_int_prefix(s: str):
    i, n = 0, len(s)
    while i < n and s[i] in " \t\n\v\f\r":
        i += 1
    neg = False
    if i < n and s[i] in "+-":
        neg = s[i] == "-"
        i += 1
    j = i
    while j < n and s[j].isdigit() and s[j].isascii():
        j += 1
    if j == i:
        raise InvalidArgument("stoi")  # std::invalid_argument
    v = int(s[i:j])
    return -v if neg else v


def _stoi_family(s: str, fn: str) -> int:
    v = _parse_int_prefix(s)
    if fn == "stoi":
        if not -(2 ** 31) <= v < 2 ** 31:
            raise InvalidArgument("stoi: out of range")
        return v
    if fn == "stoll":
        if not -(2 ** 63) <= v < 2 ** 63:
            raise InvalidArgument("stoll: out of range")
        return v
    # stoul / stoull: strtoul negates in unsigned arithmetic (64-bit unsigned long)
    if abs(v) > UINT64_MAX:
        raise InvalidArgument("stoul: out of range")
    return v & UINT64_MAX


class OutOfRange(RuntimeError):
    """std::out_of_range out of std::stod: strtod reported ERANGE."""


def _stod(s: str) -> float:
    t = s.strip()
    # longest valid prefix, like strtod
    for end in range(len(t), 0, -1):
        try:
            v = float(t[:end])
        except ValueError:
            continue
        # std::stod throws std::out_of_range when strtod sets ERANGE: overflow, or a tiny (subnormal / flushed to zero)
        # inexact result. The MAX identity DBL_MIN printed with "%.15g" is such a string ("2.2250738585072e-308" is
        # below DBL_MIN), so sorting on a double MAX column that kept its identity fails in the reference.
        body = t[:end].lower().lstrip("+-")
        if body.startswith(("inf", "nan")):
            return v
        mantissa = body.split("e")[0]
        if v in (float("inf"), float("-inf")) or (abs(v) < sys.float_info.min and any(c in "123456789" for c in mantissa)):
            raise OutOfRange("stod")
        return v
    raise InvalidArgument("stod")


# ---------------------------------------------------------------------------------
# calendar arithmetic — src/util/time.h:27-137, src/util/time.cc:49-83 (gmtime_r / timegm)
# ---------------------------------------------------------------------------------
YEAR, MONTH, WEEK, DAY, HOUR, MINUTE, SECOND, UNDEFINED = range(8)
TIME_UNIT_NAMES = ["year", "month", "week", "day", "hour", "minute", "second"]


def time_unit_by_name(name: str) -> int:
    if name in TIME_UNIT_NAMES:
        return TIME_UNIT_NAMES.index(name)
    raise InvalidArgument("Unsupported time unit: " + name)


def days_from_civil(y: int, m: int, d: int) -> int:
    y -= m <= 2
    era = y // 400
    yoe = y - era * 400
    doy = (153 * (m + (-3 if m > 2 else 9)) + 2) // 5 + d - 1
    doe = yoe * 365 + yoe // 4 - yoe // 100 + doy
    return era * 146097 + doe - 719468


def civil_from_days(z: int):
    z += 719468
    era = z // 146097
    doe = z - era * 146097
    yoe = (doe - doe // 1460 + doe // 36524 - doe // 146096) // 365
    y = yoe + era * 400
    doy = doe - (365 * yoe + yoe // 4 - yoe // 100)
    mp = (5 * doy + 2) // 153
    d = doy - (153 * mp + 2) // 5 + 1
    m = mp + (3 if mp < 10 else -9)
    return y + (m <= 2), m, d


def gmtime(t: int):
    """gmtime_r: (year, mon 1-12, mday, hour, min, sec)."""
    days, rem = divmod(t, 86400)
    y, m, d = civil_from_days(days)
    return [y, m, d, rem // 3600, rem % 3600 // 60, rem % 60]


def timegm(tm) -> int:
    """timegm with glibc's normalisation of out-of-range fields (month first, rest additive)."""
    y, m, d, hh, mm, ss = tm
    y += (m - 1) // 12
    m = (m - 1) % 12 + 1
    return (days_from_civil(y, m, 1) + d - 1) * 86400 + hh * 3600 + mm * 60 + ss


def trunc_tm(tm, unit: int):
    """Truncator::trunc<U> (time.h:57-89). WEEK has no specialisation in the reference."""
    if unit == WEEK or unit == UNDEFINED:
        raise Unsupported("Truncator::trunc<WEEK> does not exist in the reference (time.h:57-89)")
    if unit <= MINUTE:
        tm[5] = 0
    if unit <= HOUR:
        tm[4] = 0
    if unit <= DAY:
        tm[3] = 0
    if unit <= MONTH:
        tm[2] = 1
    if unit <= YEAR:
        tm[1] = 1
    return tm


def trunc_ts(secs: int, unit: int) -> int:
    return timegm(trunc_tm(gmtime(secs), unit))


@dataclass
class Duration:
    time_unit: int
    count: int

    @staticmethod
    def parse(desc: str) -> "Duration":
        """Duration(const std::string&) (time.cc:36-47): "<n> <unit>s"."""
        parts = desc.split()
        try:
            n = int(parts[0])
            tu = parts[1]
        except (IndexError, ValueError):
            raise InvalidArgument("Wrong duration description: " + desc)
        if n <= 0:
            raise InvalidArgument("Wrong duration description: " + desc)
        return Duration(time_unit_by_name(tu[:-1]), n)

    def add_to(self, timestamp: int, sign: int) -> int:
        """Duration::add_to(uint32_t, int) (time.cc:49-79); result truncated to uint32."""
        tm = gmtime(timestamp & UINT32_MAX)
        n = sign * self.count
        if self.time_unit == YEAR:
            tm[0] += n
        elif self.time_unit == MONTH:
            tm[1] += n
        elif self.time_unit == WEEK:
            tm[2] += 7 * n
        elif self.time_unit == DAY:
            tm[2] += n
        elif self.time_unit == HOUR:
            tm[3] += n
        elif self.time_unit == MINUTE:
            tm[4] += n
        elif self.time_unit == SECOND:
            tm[5] += n
        else:
            raise Unsupported("Unsupported duration")
        return timegm(tm) & UINT32_MAX

    def key(self):
        return self.add_to(0, 1)  # operator> compares add_to(0, 1) (time.h:45-47)


def strptime_subset(value: str, fmt: str, tm: list):
    """glibc strptime for the directives the reference's tests/formats use.
    Returns the index in `value` where parsing stopped, or None on mismatch.
    `tm` ([y, mon, mday, h, m, s]) keeps fields the format does not set."""
    fmt = fmt.replace("%T", "%H:%M:%S").replace("%D", "%m/%d/%y").replace("%F", "%Y-%m-%d")
    i, n = 0, len(value)
    k = 0
    while k < len(fmt):
        c = fmt[k]
        if c.isspace():
            while i < n and value[i].isspace():
                i += 1
            k += 1
            continue
        if c != "%":
            if i >= n or value[i] != c:
                return None
            i += 1
            k += 1
            continue
        k += 1
        d = fmt[k]
        k += 1
        widths = {"Y": 4, "m": 2, "d": 2, "H": 2, "M": 2, "S": 2, "y": 2, "e": 2}
        if d not in widths:
            raise Unsupported("strptime directive %" + d)
        while i < n and value[i].isspace():
            i += 1
        j = i
        while j < n and j - i < widths[d] and value[j].isdigit():
            j += 1
        if j == i:
            return None
        num = int(value[i:j])
        i = j
        if d == "Y":
            tm[0] = num
        elif d == "y":
            tm[0] = 1900 + num if num >= 69 else 2000 + num
        elif d == "m":
            tm[1] = num
        elif d in ("d", "e"):
            tm[2] = num
        elif d == "H":
            tm[3] = num
        elif d == "M":
            tm[4] = num
        elif d == "S":
            tm[5] = num
    return i


class TimeN:
    """util::Time32 / util::Time64 (time.h:91-137)."""

    def __init__(self, micro: bool):
        self.micro = micro
        self.tm = [1900, 1, 0, 0, 0, 0]  # std::tm{}: year 1900, mon 0(+1), mday 0
        self.micros = 0

    def parse(self, fmt: str, value: str):
        strptime_subset(value, fmt, self.tm)
        self.micros = 0

    def set_ts(self, ts: int):
        if self.micro:
            self.micros = ts % 1000000
            self.tm = gmtime(ts // 1000000)
        else:
            self.tm = gmtime(ts)

    def get_ts(self) -> int:
        if self.micro:
            return (timegm(self.tm) * 1000000 + self.micros) & UINT64_MAX
        return timegm(self.tm) & UINT32_MAX

    def trunc(self, unit: int):
        trunc_tm(self.tm, unit)
        if self.micro:
            self.micros = 0


def rollup_ts(ts: int, micro: bool, rules: List[Tuple[int, int]], granularity: Optional[int]) -> int:
    """set_ts -> TimestampRollup chain -> query granularity -> get_ts
    (src/codegen/query/scan.cc:197-218, src/codegen/db/rollup.cc:77-95).
    rules: [(granularity unit, boundary)], already in the reference's order."""
    t = TimeN(micro)
    t.set_ts(ts)
    for unit, before in rules:
        if ts < before:
            t.trunc(unit)
            break
    if granularity is not None:
        t.trunc(granularity)
    return t.get_ts()


def rollup_array(vals: np.ndarray, micro: bool, rules, granularity) -> np.ndarray:
    """Vectorised rollup_ts via the distinct values (exact, per-value scalar path)."""
    uniq, inv = np.unique(vals, return_inverse=True)
    out = np.array([rollup_ts(int(u), micro, rules, granularity) for u in uniq], dtype=vals.dtype)
    return out[inv]


# ---------------------------------------------------------------------------------
# schema — src/db/table.cc:47-96, src/db/column.cc:253-402
# ---------------------------------------------------------------------------------
@dataclass
class RollupRule:
    granularity: int
    after: Duration


@dataclass
class Column:
    name: str
    index: int
    is_dim: bool
    num_type: NumType
    # dimensions
    dim_type: str = ""  # string | numeric | time | boolean
    cardinality: int = UINT32_MAX
    length: int = -1
    format: str = ""
    granularity: Optional[int] = None
    rollup_rules: List[RollupRule] = field(default_factory=list)
    micro: bool = False
    guard: Optional[dict] = None
    # metrics
    agg: str = ""  # max | min | sum | avg | count | bitset
    field: str = ""

    @property
    def sort_type(self) -> str:
        """Column::sort_type (column.h:176-283)."""
        if self.is_dim:
            if self.dim_type in ("string", "time"):
                return "string"
            if self.dim_type == "boolean":
                return "integer"
            return "float" if self.num_type.fp else "integer"
        if self.agg == "bitset":
            return "integer"
        return "float" if self.num_type.fp else "integer"


class Dictionary:
    """db::DimensionDict (src/db/dictionary.cc:24-75): code 0 = "__exceeded"."""

    def __init__(self, num_type: NumType):
        self.num_type = num_type
        self.c2v = ["__exceeded"]
        self.v2c = {"__exceeded": 0}

    def decode(self, value: str) -> int:
        """Decode: a miss gives UINTn_MAX (dictionary.cc:46-75)."""
        return self.v2c.get(value, (1 << (self.num_type.size * 8)) - 1)


class Table:
    def __init__(self, conf: dict, dicts: Optional[Dict[str, Dictionary]] = None):
        self.name = conf["name"]
        self.segment_size = int(conf.get("segment_size", 1000000))
        self.dims: List[Column] = []
        self.metrics: List[Column] = []
        self.dicts: Dict[str, Dictionary] = dicts if dicts is not None else {}
        for i, dc in enumerate(conf["dimensions"]):
            t = dc.get("type", "string")
            col = Column(dc["name"], i, True, NumType("uint"), field=dc.get("field", ""))
            if t == "string":
                col.dim_type = "string"
                col.cardinality = int(dc.get("cardinality", UINT32_MAX))
                col.num_type = uint_type_for_max(col.cardinality)
                col.length = int(dc.get("length", -1))
                self.dicts.setdefault(col.name, Dictionary(col.num_type))
            elif t == "boolean":
                col.dim_type = "boolean"
                col.num_type = NumType("ubyte")
            elif t in ("time", "microtime"):
                col.dim_type = "time"
                col.micro = t == "microtime"
                col.num_type = NumType("ulong" if col.micro else "uint")
                col.format = dc.get("format", "")
                if "granularity" in dc:
                    col.granularity = time_unit_by_name(dc["granularity"])
                elif "rollup_rules" in dc:
                    rules = [RollupRule(time_unit_by_name(r["granularity"]), Duration.parse(r["after"]))
                             for r in dc["rollup_rules"]]
                    # sorted by `after` descending (column.cc:346-349)
                    col.rollup_rules = sorted(rules, key=lambda r: -r.after.key())
            else:
                col.dim_type = "numeric"
                if t == "numeric":
                    big = uint_type_for_max(int(dc.get("max", UINT32_MAX))).size == 8
                    col.num_type = NumType("ulong" if big else "uint")
                else:
                    if t not in NUMERIC_TYPES:
                        raise InvalidArgument("Unsupported metric type: " + t)
                    col.num_type = NumType(t)
            if "cardinality_guard" in dc:
                col.guard = dc["cardinality_guard"]
            self.dims.append(col)
        for i, mc in enumerate(conf["metrics"]):
            t = mc["type"]
            col = Column(mc["name"], i, False, NumType("uint"), field=mc.get("field", ""))
            if t == "bitset":
                col.agg = "bitset"
                col.num_type = uint_type_for_max(int(mc.get("max", UINT32_MAX)))
                if col.num_type.size < 4:  # util::Bitset<N> only distinguishes 8 from not-8
                    pass
            elif t == "count":
                col.agg = "count"
                big = uint_type_for_max(int(mc.get("max", UINT32_MAX))).size == 8
                col.num_type = NumType("ulong" if big else "uint")
            else:
                pre, _, suf = t.partition("_")
                if suf not in ("sum", "max", "min", "avg"):
                    raise InvalidArgument("Unsupported metric type: " + t)
                if pre not in NUMERIC_TYPES:
                    raise InvalidArgument("Unsupported metric type: " + pre)
                col.agg = suf
                col.num_type = NumType(pre)
            self.metrics.append(col)
        for d in self.dims:
            if d.guard is not None:
                if d.dim_type == "numeric":
                    raise InvalidArgument("Can't define cardinality guard on a numeric dimension")
                d.guard = {"dims": [self.dimension(n) for n in d.guard["dimensions"]], "limit": int(d.guard["limit"]),
                           "stats": {}}
        self.has_hidden_count = any(m.agg == "avg" for m in self.metrics) and not any(m.agg == "count" for m in self.metrics)
        self.segments: List[dict] = []  # {"size", "d": [arrays], "m": [arrays / list of sets], "count": arr, "dmin", "dmax"}
        self.tuple_offsets: Dict[tuple, int] = {}
        self.ingest_time = {d.index: TimeN(d.micro) for d in self.dims if d.dim_type == "time"}
        self.ingest_rollup: Dict[int, List[int]] = {}

    # ---- lookups (table.cc:110-150)
    def column(self, name: str) -> Column:
        for c in self.dims + self.metrics:
            if c.name == name:
                return c
        raise InvalidArgument("No such column: " + name)

    def dimension(self, name: str) -> Column:
        for c in self.dims:
            if c.name == name:
                return c
        raise InvalidArgument("No such dimension: " + name)

    def metric(self, name: str) -> Column:
        for c in self.metrics:
            if c.name == name:
                return c
        raise InvalidArgument("No such metric: " + name)

    # ---- storage (src/codegen/db/store.cc:203-356)
    def _new_segment(self) -> dict:
        s = self.segment_size
        seg = {"size": 0, "d": [np.zeros(s, dtype=d.num_type.dtype) for d in self.dims], "m": [], "count": None,
               "dmin": {}, "dmax": {}}
        for m in self.metrics:
            if m.agg == "bitset":
                seg["m"].append([set() for _ in range(s)])
            elif m.agg == "max":
                seg["m"].append(np.full(s, m.num_type.cpp_min_value, dtype=m.num_type.dtype))
            elif m.agg == "min":
                seg["m"].append(np.full(s, m.num_type.cpp_max_value, dtype=m.num_type.dtype))
            else:
                seg["m"].append(np.zeros(s, dtype=m.num_type.dtype))
        if self.has_hidden_count:
            seg["count"] = np.zeros(s, dtype=np.uint64)
        for d in self.dims:  # SegmentStats (store.cc:171-201)
            if d.dim_type in ("numeric", "time"):
                seg["dmax"][d.index] = d.num_type.dtype.type(d.num_type.cpp_min_value)
                seg["dmin"][d.index] = d.num_type.dtype.type(d.num_type.cpp_max_value)
        self.segments.append(seg)
        return seg

    def add_segment_arrays(self, dims: List[np.ndarray], metrics: List, count: Optional[np.ndarray] = None, size=None):
        """Install a ready-made SoA segment (synthetic data; bypasses upsert)."""
        n = size if size is not None else len(dims[0] if dims else metrics[0])
        seg = {"size": n, "d": list(dims), "m": list(metrics), "count": count, "dmin": {}, "dmax": {}}
        for d in self.dims:
            if d.dim_type in ("numeric", "time"):
                col = dims[d.index][:n]
                ident_max = d.num_type.dtype.type(d.num_type.cpp_min_value)
                ident_min = d.num_type.dtype.type(d.num_type.cpp_max_value)
                seg["dmax"][d.index] = max(col.max(), ident_max) if n else ident_max
                seg["dmin"][d.index] = min(col.min(), ident_min) if n else ident_min
        self.segments.append(seg)
        return seg

    # ---- ingest (src/codegen/db/upsert.cc:29-151, 340-421)
    def before_load(self, now: Optional[int] = None):
        """viya_upsert_before -> UpsertContext::Reset -> RollupReset (rollup.cc:44-75)."""
        if now is None:
            now = int(_time.time())
        for d in self.dims:
            if d.dim_type == "time":
                self.ingest_rollup[d.index] = rollup_boundaries(d, now)

    def load(self, rows: List[List[str]], now: Optional[int] = None, columns: Optional[List[str]] = None):
        self.before_load(now)
        input_cols = list(self.dims) + [m for m in self.metrics if m.agg != "count"]
        if columns is not None:  # LoaderDesc::InitTupleIdxMap (loader_desc.cc:54-96)
            idx_map = []
            for c in input_cols:
                nm = c.field or c.name
                if nm not in columns:
                    raise RuntimeError("Column name '%s' is not specified in load spec" % nm)
                idx_map.append(columns.index(nm))
        else:
            if any(c.field for c in self.dims + self.metrics):
                raise RuntimeError("Column names must be specified, because one or more columns define field name mapping")
            idx_map = list(range(len(input_cols)))
        for r in rows:
            self._upsert(list(r), idx_map)

    def _upsert(self, values: List[str], idx_map: List[int]):
        vi = 0
        dvals = []
        for d in self.dims:
            v = values[idx_map[vi]]
            if d.dim_type == "string":
                if d.length != -1 and len(v) > d.length:
                    v = v[:d.length]
                dic = self.dicts[d.name]
                code = dic.v2c.get(v)
                if code is None:
                    code = len(dic.c2v)
                    if d.cardinality < UINT64_MAX - 1 and not code <= d.cardinality:
                        code = 0
                    else:
                        dic.v2c[v] = code
                        dic.c2v.append(v)
                dvals.append(wrap_int(code, d.num_type.dtype))
            elif d.dim_type == "numeric":
                dvals.append(d.num_type.parse(v))
            elif d.dim_type == "boolean":
                dvals.append(np.uint8(1 if v == "true" else 0))
            else:
                dvals.append(self._ingest_time(d, v))
            vi += 1
        mvals = []
        for m in self.metrics:
            if m.agg == "count":
                mvals.append(m.num_type.dtype.type(1))
                continue
            v = values[idx_map[vi]]
            vi += 1
            if m.agg == "bitset":
                mvals.append({int(m.num_type.parse(v))})
            else:
                mvals.append(m.num_type.parse(v))
        # CardinalityProtection (upsert.cc:266-300)
        for d in self.dims:
            if d.guard is None:
                continue
            key = tuple(int(dvals[p.index]) for p in d.guard["dims"])
            stats = d.guard["stats"]
            code = int(dvals[d.index])
            if key not in stats:
                stats[key] = {code}
            else:
                bs = stats[key]
                if len(bs) >= d.guard["limit"]:
                    if code not in bs:
                        dvals[d.index] = d.num_type.dtype.type(0)
                else:
                    bs.add(code)
        key = tuple(_hashable(v) for v in dvals)
        off = self.tuple_offsets.get(key)
        if off is not None:  # Metrics::Update in place (store.cc:311-340)
            seg = self.segments[off // self.segment_size]
            ti = off % self.segment_size
            for m, nv in zip(self.metrics, mvals):
                arr = seg["m"][m.index]
                if m.agg in ("sum", "avg", "count"):
                    with np.errstate(over="ignore"):
                        arr[ti] = arr[ti] + nv
                elif m.agg == "max":
                    arr[ti] = max(arr[ti], nv)
                elif m.agg == "min":
                    arr[ti] = min(arr[ti], nv)
                else:
                    arr[ti] |= nv
            if self.has_hidden_count:
                seg["count"][ti] += np.uint64(1)
        else:
            if not self.segments or self.segments[-1]["size"] == self.segment_size:
                self._new_segment()
            seg = self.segments[-1]
            ti = seg["size"]
            for d, v in zip(self.dims, dvals):
                seg["d"][d.index][ti] = v
            for m, nv in zip(self.metrics, mvals):
                seg["m"][m.index][ti] = nv
            if self.has_hidden_count:
                seg["count"][ti] = 1
            seg["size"] += 1
            for d, v in zip(self.dims, dvals):
                if d.dim_type in ("numeric", "time"):
                    seg["dmax"][d.index] = max(v, seg["dmax"][d.index])
                    seg["dmin"][d.index] = min(v, seg["dmin"][d.index])
            self.tuple_offsets[key] = (len(self.segments) - 1) * self.segment_size + ti

    def _ingest_time(self, d: Column, value: str):
        """ValueParser::Visit(TimeDimension) (upsert.cc:82-139)."""
        t = self.ingest_time[d.index]
        fmt = d.format
        is_num = fmt in ("", "posix", "millis", "micros")
        dt = d.num_type.dtype
        tup = None
        if is_num:
            ts = _stoi_family(value, "stoull")
            if d.micro:
                if fmt == "posix":
                    ts *= 1000000
                elif fmt == "millis":
                    ts *= 1000
            else:
                if fmt == "millis":
                    ts //= 1000
                elif fmt == "micros":
                    ts //= 1000000
            tup = int(wrap_int(ts, dt))
            if d.rollup_rules or d.granularity is not None:
                t.set_ts(tup)
        else:
            t.parse(fmt, value)
            if d.rollup_rules:
                tup = t.get_ts()
        if d.rollup_rules:
            for rule, before in zip(d.rollup_rules, self.ingest_rollup[d.index]):
                if tup < before:
                    t.trunc(rule.granularity)
                    break
        elif d.granularity is not None:
            t.trunc(d.granularity)
        if not is_num or d.rollup_rules or d.granularity is not None:
            tup = t.get_ts()
        return wrap_int(tup, dt)


def _hashable(v):
    if isinstance(v, (np.floating, float)):
        return float(v)
    return int(v)


def rollup_boundaries(d: Column, now: int) -> List[int]:
    """RollupReset (src/codegen/db/rollup.cc:44-75): b_i = Duration.add_to((uint32_t)now, -1) [* 1e6]."""
    out = []
    for r in d.rollup_rules:
        b = r.after.add_to(now & UINT32_MAX, -1)
        if d.micro:
            b = (b * 1000000) & UINT64_MAX
        out.append(b)
    return out


class Database:
    """db::Database reduced to what the path needs (src/db/database.cc:57-125)."""

    def __init__(self, conf: dict):
        self.dicts: Dict[str, Dictionary] = {}
        self.tables: Dict[str, Table] = {}
        for tc in conf.get("tables", []):
            self.create_table(tc)

    def create_table(self, tc: dict):
        self.tables[tc["name"]] = Table(tc, self.dicts)

    def table(self, name: str) -> Table:
        if name not in self.tables:
            raise InvalidArgument("No such table: " + name)
        return self.tables[name]

    def query(self, q: dict, now: Optional[int] = None):
        t = q.get("type")
        if t == "aggregate":
            return aggregate_query(self.table(q["table"]), q, now)
        if t == "select":
            return select_query(self.table(q["table"]), q)
        if t == "search":
            return search_query(self.table(q["table"]), q)
        raise Unsupported("oracle covers aggregate, select and search queries only")


# ---------------------------------------------------------------------------------
# filters — src/query/filter.h:38-134, src/query/filter.cc:36-108
# ---------------------------------------------------------------------------------
OPS = ["eq", "ne", "lt", "le", "gt", "ge"]
_NEGATED = {"eq": "ne", "ne": "eq", "lt": "ge", "le": "gt", "gt": "le", "ge": "lt"}


@dataclass
class Rel:
    op: str
    column: str
    value: str
    precedence: int = 1


@dataclass
class In:
    column: str
    values: List[str]
    equal: bool
    precedence: int = 4


@dataclass
class Composite:
    op: str  # "and" | "or"
    filters: list
    precedence: int = 2


@dataclass
class Empty:
    precedence: int = 0


def make_filter(conf: Optional[dict], negate: bool = False):
    """FilterFactory::Create (filter.cc:36-108): NOT is pushed down by De Morgan / negated
    operators; children of a composite are sorted by precedence."""
    if not conf or "op" not in conf:
        return Empty()
    op = conf["op"]
    if op in ("and", "or"):
        kids = [make_filter(c, negate) for c in conf["filters"]]
        kids.sort(key=lambda f: f.precedence)
        eff = op if not negate else ("or" if op == "and" else "and")
        return Composite(eff, kids, 2 if eff == "and" else 3)
    if op == "not":
        return make_filter(conf["filter"], not negate)
    column = conf["column"]
    if op == "in":
        return In(column, [str(v) for v in conf["values"]], not negate)
    if op not in OPS:
        raise InvalidArgument("Unsupported filter operataor: " + op)
    return Rel(_NEGATED[op] if negate else op, column, str(conf["value"]))


def filter_columns(f, out=None):
    out = set() if out is None else out
    if isinstance(f, (Rel, In)):
        out.add(f.column)
    elif isinstance(f, Composite):
        for c in f.filters:
            filter_columns(c, out)
    return out


def decode_value(table: Table, col: Column, value: str):
    """ValueDecoder (src/codegen/query/filter.cc:154-204) + ArgsUnpacker's get_<type>()."""
    if col.is_dim and col.dim_type == "string":
        return wrap_int(table.dicts[col.name].decode(value), col.num_type.dtype)
    if col.is_dim and col.dim_type == "boolean":
        return np.uint8(1 if value == "true" else 0)
    if col.is_dim and col.dim_type == "time":
        if all(ch.isdigit() for ch in value):
            return col.num_type.parse(value)
        mult = 1000000 if col.micro else 1
        ts = 0
        tm = [1900, 1, 0, 0, 0, 0]
        r = strptime_subset(value, "%Y-%m-%d %T", tm)
        if r is not None and r == len(value):
            ts = timegm(tm) * mult
        elif r is not None and col.micro and value[r:r + 1] == ".":
            # reference: timegm(&tm) * multiplier + std::stoul(r) with r pointing AT the '.',
            # so stoul finds no digits and throws std::invalid_argument
            ts = timegm(tm) * mult + _stoi_family(value[r:], "stoul")
        else:
            tm = [1900, 1, 0, 0, 0, 0]
            r = strptime_subset(value, "%Y-%m-%d", tm)
            if r is not None and r == len(value):
                ts = timegm(tm) * mult
        if ts > 0:
            return wrap_int(ts, col.num_type.dtype)
        raise InvalidArgument("Unrecognized time format: " + value)
    # numeric dimension, value metric, bitset metric
    if col.num_type.name in ("byte", "short"):
        # generated code calls AnyNum::get_int8_t()/get_int16_t(), which do not exist
        # (src/db/column.h:110-117 vs src/codegen/query/filter.cc:126-132)
        raise Unsupported("filters on byte/short columns do not compile in the reference")
    return col.num_type.parse(value)


_CMP = {"eq": np.equal, "ne": np.not_equal, "lt": np.less, "le": np.less_equal, "gt": np.greater, "ge": np.greater_equal}


def eval_filter(table: Table, f, getcol) -> np.ndarray:
    """ComparisonBuilder (filter.cc:206-261): comparisons in the column's own type,
    composites bitwise & / | without short circuit. `getcol(col)` returns the array the
    predicate sees (bitset metrics: cardinalities)."""
    if isinstance(f, Empty):
        return None  # "true"
    if isinstance(f, Rel):
        col = table.column(f.column)
        lit = decode_value(table, col, f.value)
        return _CMP[f.op](getcol(col), lit)
    if isinstance(f, In):
        col = table.column(f.column)
        arr = getcol(col)
        lits = [decode_value(table, col, v) for v in f.values]
        if f.equal:
            r = np.zeros(len(arr), dtype=bool)
            for l in lits:
                r |= arr == l
        else:
            r = np.ones(len(arr), dtype=bool)
            for l in lits:
                r &= arr != l
        if not lits:
            raise Unsupported("IN with no values generates '()' in the reference")
        return r
    parts = [eval_filter(table, c, getcol) for c in f.filters]
    r = None
    for p in parts:
        if p is None:
            p = True
        r = p if r is None else ((r & p) if f.op == "and" else (r | p))
    return r


def segment_skip(table: Table, f, seg) -> bool:
    """SegmentSkipBuilder (filter.cc:263-335): True = process the segment."""
    if isinstance(f, Empty):
        return True
    if isinstance(f, Rel):
        col = table.column(f.column)
        lit = decode_value(table, col, f.value)  # args are unpacked whether or not they are used
        if col.is_dim and col.dim_type in ("numeric", "time"):
            dmin, dmax = seg["dmin"][col.index], seg["dmax"][col.index]
            if f.op == "eq":
                return bool((dmin <= lit) & (dmax >= lit))
            if f.op in ("lt", "le"):
                return bool(dmin <= lit)
            if f.op in ("gt", "ge"):
                return bool(dmax >= lit)
        return True
    if isinstance(f, In):
        col = table.column(f.column)
        lits = [decode_value(table, col, v) for v in f.values]
        if col.is_dim and col.dim_type in ("numeric", "time"):
            dmin, dmax = seg["dmin"][col.index], seg["dmax"][col.index]
            r = False
            for lit in lits:  # NOT IN takes the same expression: the reference ignores equal()
                r = r | bool((dmin <= lit) & (dmax >= lit))
            return r
        return True
    r = None
    for c in f.filters:
        p = segment_skip(table, c, seg)
        r = p if r is None else ((r & p) if f.op == "and" else (r | p))
    return bool(r)


# ---------------------------------------------------------------------------------
# the aggregate query — src/query/query.cc:48-135, src/codegen/query/scan.cc:168-247,
# src/codegen/db/store.cc:31-169, src/codegen/query/post_agg.cc:26-147, sort.cc:24-75
# ---------------------------------------------------------------------------------
@dataclass
class OutCol:
    col: Column
    index: int
    format: str = ""
    granularity: Optional[int] = None


@dataclass
class AggQuery:
    table: Table
    dim_cols: List[OutCol]
    metric_cols: List[OutCol]
    filter: object
    having: object
    sort: List[Tuple[Column, int, bool]]
    skip: int
    limit: int
    header: bool


def parse_query(table: Table, q: dict) -> AggQuery:
    dim_cols, metric_cols = [], []
    idx = 0
    if "select" in q:  # query.cc:56-74
        for sc in q["select"]:
            name = sc["column"]
            cols = (table.dims + table.metrics) if name == "*" else [table.column(name)]
            for c in cols:
                if c.is_dim:
                    oc = OutCol(c, idx)
                    if c.dim_type == "time":  # DimOutputColumn (query.cc:37-46)
                        oc.format = sc.get("format", c.format)
                        if "granularity" in sc:
                            oc.granularity = time_unit_by_name(sc["granularity"])
                    dim_cols.append(oc)
                else:
                    metric_cols.append(OutCol(c, idx))
                idx += 1
    else:
        if "dimensions" not in q or "metrics" not in q:
            raise InvalidArgument("dimensions and metrics are mandatory")
        for n in q["dimensions"]:
            dim_cols.append(OutCol(table.dimension(n), idx))
            idx += 1
        for n in q["metrics"]:
            metric_cols.append(OutCol(table.metric(n), idx))
            idx += 1
    sort = []
    for sc in q.get("sort", []):  # query.cc:90-118
        col = table.column(sc["column"])
        ci = -1
        for oc in dim_cols + metric_cols:
            if oc.col is col:
                ci = oc.index
                break
        if ci == -1:
            raise InvalidArgument("Sort column '%s' is not selected" % sc["column"])
        sort.append((col, ci, bool(sc.get("ascending", False))))
    having = None
    if "having" in q:
        having = make_filter(q["having"])
        names = [oc.col.name for oc in dim_cols] + [oc.col.name for oc in metric_cols]
        for hc in filter_columns(having):
            if hc not in names:
                raise InvalidArgument("Column '%s is not selected" % hc)
    return AggQuery(table, dim_cols, metric_cols, make_filter(q.get("filter")), having, sort,
                    int(q.get("skip", 0)), int(q.get("limit", 0)), bool(q.get("header", False)))


@dataclass
class AggState:
    """The contents of agg_map after the scan: one row per group."""
    keys: List[np.ndarray]        # per dimension_cols entry, the column's dtype
    states: List[np.ndarray]      # per metric_cols entry (bitset: uint64 cardinality)
    hidden_count: Optional[np.ndarray]
    scanned_recs: int = 0
    scanned_segments: int = 0
    passed_recs: int = 0
    bitsets: Optional[Dict[int, list]] = None   # metric_cols position -> per group, the set itself (cluster partial states)

    @property
    def ngroups(self):
        if self.keys:
            return len(self.keys[0])
        if self.states:
            return len(self.states[0])
        return 0


def scan_aggregate(aq: AggQuery, now: Optional[int] = None, seg_rows: Optional[List[int]] = None) -> AggState:
    """ScanVisitor::Visit(AggregateQuery*) (scan.cc:168-247) + TupleStruct::Update (store.cc:131-161)."""
    table = aq.table
    if now is None:
        now = int(_time.time())
    rules = {}
    for oc in aq.dim_cols:
        d = oc.col
        if d.dim_type == "time" and (d.rollup_rules or oc.granularity is not None):
            rules[oc.index] = (list(zip([r.granularity for r in d.rollup_rules], rollup_boundaries(d, now))), oc.granularity)
            for r in d.rollup_rules:
                if r.granularity == WEEK:
                    raise Unsupported("week rollup")
            if oc.granularity == WEEK:
                raise Unsupported("week granularity")
    has_avg = any(oc.col.agg == "avg" for oc in aq.metric_cols)
    has_count = any(oc.col.agg == "count" for oc in aq.metric_cols)
    need_hidden = has_avg and not has_count
    if need_hidden and not table.has_hidden_count:
        # the generated loop copies tuple_metrics._count, which the table's Metrics struct only has when the TABLE has an
        # AVG and no COUNT metric (src/codegen/db/store.cc:126-129, src/codegen/query/scan.cc:239-241)
        raise Unsupported("AVG selected without the table's COUNT metric does not compile in the reference")
    key_parts = [[] for _ in aq.dim_cols]
    val_parts = [[] for _ in aq.metric_cols]
    hid_parts = []
    st = AggState([], [], None)
    for si, seg in enumerate(table.segments):
        size = seg["size"] if seg_rows is None else int(seg_rows[si])
        st.scanned_recs += size
        if not segment_skip(table, aq.filter, seg):
            continue
        st.scanned_segments += 1

        def getcol(c, seg=seg, size=size):
            if c.is_dim:
                return seg["d"][c.index][:size]
            if c.agg == "bitset":
                dt = np.uint64 if c.num_type.size == 8 else np.uint32
                return np.array([len(s) for s in seg["m"][c.index][:size]], dtype=dt)
            return seg["m"][c.index][:size]

        r = eval_filter(table, aq.filter, getcol)
        idx = np.arange(size) if r is None else np.nonzero(r)[0]
        st.passed_recs += len(idx)
        if not len(idx):
            continue
        for k, oc in enumerate(aq.dim_cols):
            v = seg["d"][oc.col.index][:size][idx]
            if oc.index in rules:
                v = rollup_array(v, oc.col.micro, rules[oc.index][0], rules[oc.index][1])
            key_parts[k].append(v)
        for k, oc in enumerate(aq.metric_cols):
            if oc.col.agg == "bitset":
                col = seg["m"][oc.col.index]
                val_parts[k].append([col[i] for i in idx])
            else:
                val_parts[k].append(seg["m"][oc.col.index][:size][idx])
        if need_hidden:
            hid_parts.append(seg["count"][:size][idx])

    n = st.passed_recs
    keys = [np.concatenate(p) if p else np.zeros(0, dtype=oc.col.num_type.dtype) for p, oc in zip(key_parts, aq.dim_cols)]
    # group: lexsort on the key columns, boundaries where any column changes (KeyEqual: field-wise ==)
    change = np.zeros(n, dtype=bool)
    if n:
        change[0] = True
    if keys and n:
        order = np.lexsort([k for k in reversed(keys)])
        for k in keys:
            ks = k[order]
            change[1:] |= ks[1:] != ks[:-1]
    else:
        order = np.arange(n)
    starts = np.nonzero(change)[0]
    st.keys = [k[order][starts] for k in keys]
    gid_sorted = np.cumsum(change) - 1
    for k, oc in enumerate(aq.metric_cols):
        m = oc.col
        if m.agg == "bitset":
            flat = [s for part in val_parts[k] for s in part]
            sets = [set() for _ in range(len(starts))]
            for pos, oi in enumerate(order):
                sets[gid_sorted[pos]] |= flat[oi]
            st.states.append(np.array([len(s) for s in sets], dtype=np.uint64))
            st.bitsets = st.bitsets or {}
            st.bitsets[k] = sets
            continue
        vals = np.concatenate(val_parts[k]) if val_parts[k] else np.zeros(0, dtype=m.num_type.dtype)
        vals = vals[order]
        dt = m.num_type.dtype
        if not len(starts):
            st.states.append(np.zeros(0, dtype=dt))
        elif m.agg in ("sum", "avg", "count"):
            with np.errstate(over="ignore"):
                st.states.append(np.add.reduceat(vals, starts, dtype=dt))
        elif m.agg == "max":  # identity cpp_min_value (store.cc:107-110)
            st.states.append(np.maximum(np.maximum.reduceat(vals, starts), dt.type(m.num_type.cpp_min_value)).astype(dt))
        else:
            st.states.append(np.minimum(np.minimum.reduceat(vals, starts), dt.type(m.num_type.cpp_max_value)).astype(dt))
    if need_hidden:
        hv = np.concatenate(hid_parts)[order] if hid_parts else np.zeros(0, dtype=np.uint64)
        st.hidden_count = np.add.reduceat(hv, starts, dtype=np.uint64) if len(starts) else np.zeros(0, dtype=np.uint64)
    return st


def fmt_num(v) -> str:
    """util::Format::num (src/util/format.h:32-65): ints decimal, double "%.15g", float via
    fmt 4.x's default (== "%g")."""
    if isinstance(v, (np.float64, float)):
        return "%.15g" % float(v)
    if isinstance(v, np.float32):
        return "%g" % float(v)
    return str(int(v))


def fmt_date(fmt: str, ts: int) -> str:
    """Format::date(const char*, uint32_t) (format.h:72-76): the argument is a uint32."""
    return _time.strftime(fmt, _time.gmtime(ts & UINT32_MAX))


def _cmp_strings(sort_type: str, asc: bool):
    """StringNumCmp (src/util/string.h:28-49) / plain string compare."""
    if sort_type == "string":
        return (lambda a, b: a < b) if asc else (lambda a, b: a > b)
    if sort_type == "integer":
        if asc:
            return lambda a, b: (len(a) < len(b)) if len(a) != len(b) else a < b
        return lambda a, b: (len(a) > len(b)) if len(a) != len(b) else a > b
    if asc:
        return lambda a, b: _stod(a) < _stod(b)
    return lambda a, b: _stod(a) > _stod(b)


def post_aggregate(aq: AggQuery, st: AggState):
    """PostAggVisitor::Visit(AggregateQuery*) (post_agg.cc:26-147) + SortVisitor (sort.cc:24-75).
    Returns (rows, output_recs). Group iteration order (std::unordered_map) is unspecified in the
    reference; here it is the oracle's group order."""
    table = aq.table
    n = st.ngroups
    rows = []
    ncols = len(aq.dim_cols) + len(aq.metric_cols)
    skip = min(n, aq.skip)
    limit = min(aq.limit, n - skip)
    lo, hi = 0, n
    if not aq.sort:
        lo = skip
        if limit > 0:
            hi = lo + limit
    if aq.header:
        hdr = [""] * ncols
        for oc in aq.dim_cols + aq.metric_cols:
            hdr[oc.index] = oc.col.name
        rows.append(hdr)
    keep = np.ones(n, dtype=bool)
    if aq.having is not None:
        def getcol(c):
            for k, oc in enumerate(aq.dim_cols):
                if oc.col is c:
                    return st.keys[k]
            for k, oc in enumerate(aq.metric_cols):
                if oc.col is c:
                    s = st.states[k]
                    if c.agg == "bitset":
                        return s.astype(np.uint64 if c.num_type.size == 8 else np.uint32)
                    return s
            raise InvalidArgument("having column not selected")
        r = eval_filter(table, aq.having, getcol)
        if r is not None:
            keep = np.asarray(r, dtype=bool)
    count_k = None
    for k, oc in enumerate(aq.metric_cols):
        if oc.col.agg == "count":
            count_k = k
            break
    body = []
    for g in range(lo, hi):
        if not keep[g]:
            continue
        row = [""] * ncols
        for k, oc in enumerate(aq.dim_cols):
            d, v = oc.col, st.keys[k][g]
            if d.dim_type == "string":
                row[oc.index] = table.dicts[d.name].c2v[int(v)]
            elif d.dim_type == "time" and oc.format:
                row[oc.index] = fmt_date(oc.format, int(v))
            elif d.dim_type == "boolean":
                row[oc.index] = "true" if v else "false"
            else:
                row[oc.index] = fmt_num(v)
        for k, oc in enumerate(aq.metric_cols):
            m, v = oc.col, st.states[k][g]
            if m.agg == "avg":
                cnt = st.states[count_k][g] if count_k is not None else st.hidden_count[g]
                row[oc.index] = fmt_num(np.float64(float(v) / float(cnt)))
            elif m.agg == "bitset":
                row[oc.index] = str(int(v))
            else:
                row[oc.index] = fmt_num(v)
        body.append(row)
    if aq.sort:
        cmps = [(ci, _cmp_strings(col.sort_type, asc)) for col, ci, asc in aq.sort]

        def cmp(a, b):
            for ci, lt in cmps:
                if lt(a[ci], b[ci]):
                    return -1
                if lt(b[ci], a[ci]):
                    return 1
            return 0
        body.sort(key=functools.cmp_to_key(cmp))
        end = skip + limit if limit > 0 else len(body)
        body = body[skip:min(end, len(body))]
    rows.extend(body)
    return rows, len(body)


def aggregate_query(table: Table, q: dict, now: Optional[int] = None):
    """Database::Query for type=aggregate. Returns (rows, stats dict)."""
    aq = parse_query(table, q)
    st = scan_aggregate(aq, now)
    rows, out = post_aggregate(aq, st)
    return rows, {"scanned_recs": st.scanned_recs, "scanned_segments": st.scanned_segments,
                  "aggregated_recs": st.ngroups, "output_recs": out, "passed_recs": st.passed_recs}


# ---------------------------------------------------------------------------------
# select / search — the other two FilterBasedQuery kinds on the same scan
# ---------------------------------------------------------------------------------
def _passing_rows(table: Table, flt, seg, size):
    def getcol(c, seg=seg, size=size):
        if c.is_dim:
            return seg["d"][c.index][:size]
        if c.agg == "bitset":
            dt = np.uint64 if c.num_type.size == 8 else np.uint32
            return np.array([len(x) for x in seg["m"][c.index][:size]], dtype=dt)
        return seg["m"][c.index][:size]
    r = eval_filter(table, flt, getcol)
    return np.arange(size) if r is None else np.nonzero(r)[0]


def scan_select(aq: AggQuery, seg_rows: Optional[List[int]] = None):
    """ScanVisitor::Visit(SelectQuery*) (scan.cc:75-166) up to the point where a row is formatted:
    -> ([(segment index, row index)...] in send order, stats). `break` leaves the tuple loop only
    (scan.cc:158-160), so once the limit is reached every later segment still sends one row."""
    table = aq.table
    picked = []
    stats = {"scanned_recs": 0, "scanned_segments": 0, "passed_recs": 0}
    row_index, output_recs = 0, 0
    for si, seg in enumerate(table.segments):
        size = seg["size"] if seg_rows is None else int(seg_rows[si])
        stats["scanned_recs"] += size
        if not segment_skip(table, aq.filter, seg):
            continue
        stats["scanned_segments"] += 1
        idx = _passing_rows(table, aq.filter, seg, size)
        stats["passed_recs"] += len(idx)
        for i in idx:
            if aq.skip > 0:
                skipit = row_index < aq.skip
                row_index += 1
                if skipit:
                    continue
            picked.append((si, int(i)))
            output_recs += 1
            if aq.limit > 0 and output_recs >= aq.limit:
                break
    stats["output_recs"] = output_recs
    stats["aggregated_recs"] = 0
    return picked, stats


def select_query(table: Table, q: dict):
    """Database::Query for type=select: rows of strings in send order (header first if asked)."""
    aq = parse_query(table, {k: v for k, v in q.items() if k not in ("sort", "having")})
    if any(oc.col.agg == "avg" for oc in aq.metric_cols) and not any(oc.col.agg == "count" for oc in aq.metric_cols) \
            and not table.has_hidden_count:
        # the generated row loop divides by `_<count field>[idx]` (scan.cc:136-152): without a selected COUNT metric that
        # is the hidden `_count`, which the Segment class only has when the TABLE has an AVG and no COUNT metric — the
        # query fails at JIT compile time, whatever the data
        raise Unsupported("AVG in a select without a COUNT metric or hidden count does not compile in the reference")
    picked, stats = scan_select(aq)
    ncols = len(aq.dim_cols) + len(aq.metric_cols)
    rows = []
    if aq.header:
        hdr = [""] * ncols
        for oc in aq.dim_cols + aq.metric_cols:
            hdr[oc.index] = oc.col.name
        rows.append(hdr)
    count_col = None
    for oc in aq.metric_cols:
        if oc.col.agg == "count":
            count_col = oc.col
            break
    for si, i in picked:
        seg = table.segments[si]
        row = [""] * ncols
        for oc in aq.dim_cols:
            d, v = oc.col, seg["d"][oc.col.index][i]
            if d.dim_type == "string":
                row[oc.index] = table.dicts[d.name].c2v[int(v)]
            elif d.dim_type == "time" and oc.format:
                row[oc.index] = fmt_date(oc.format, int(v))
            elif d.dim_type == "boolean":
                row[oc.index] = "true" if v else "false"
            else:
                row[oc.index] = fmt_num(v)
        for oc in aq.metric_cols:
            m = oc.col
            if m.agg == "bitset":
                row[oc.index] = str(len(seg["m"][m.index][i]))
            elif m.agg == "avg":   # `_j[idx] / (double) _<count field>[idx]`, count field = a selected COUNT metric or `_count`
                if count_col is not None:
                    cnt = seg["m"][count_col.index][i]
                elif seg.get("count") is not None:
                    cnt = seg["count"][i]
                else:
                    raise Unsupported("AVG in a select without a COUNT metric or hidden count does not compile in the reference")
                row[oc.index] = fmt_num(np.float64(float(seg["m"][m.index][i]) / float(cnt)))
            else:
                row[oc.index] = fmt_num(seg["m"][m.index][i])
        rows.append(row)
    return rows, stats


def scan_search(table: Table, dim: Column, flt, term: str, limit: int, seg_rows: Optional[List[int]] = None):
    """ScanVisitor::Visit(SearchQuery*) (scan.cc:249-299): -> (values in push order, stats)."""
    codes = set()
    values = []
    stats = {"scanned_recs": 0, "scanned_segments": 0, "passed_recs": 0}
    for si, seg in enumerate(table.segments):
        size = seg["size"] if seg_rows is None else int(seg_rows[si])
        stats["scanned_recs"] += size
        if not segment_skip(table, flt, seg):
            continue
        stats["scanned_segments"] += 1
        idx = _passing_rows(table, flt, seg, size)
        stats["passed_recs"] += len(idx)
        col = seg["d"][dim.index]
        for i in idx:
            v = col[i]
            key = _hashable(v)
            if key in codes:
                continue
            codes.add(key)
            if dim.dim_type == "string":
                check = table.dicts[dim.name].c2v[int(v)]
            elif dim.dim_type == "boolean":
                check = "true" if v else "false"
            else:
                check = fmt_num(v)
            if term in check:
                values.append(check)
                if limit > 0 and len(values) >= limit:
                    break
    stats["aggregated_recs"] = len(codes)
    stats["output_recs"] = len(values)
    return values, stats


def search_query(table: Table, q: dict):
    """Database::Query for type=search: optional header row, then ONE row holding all values (SendAsCol)."""
    dim = table.dimension(q["dimension"])
    values, stats = scan_search(table, dim, make_filter(q.get("filter")), q["term"], int(q.get("limit", 0)))
    rows = []
    if q.get("header", False):
        rows.append([dim.name])
    rows.append(values)
    return rows, stats

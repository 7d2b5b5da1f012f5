"""Independent reader / writer of the partial-state wire layout (viyadb_amd/host/partial_state.h), for tests.

Written from the layout comment, not from the C++: if the two disagree, one of them is wrong. `oracle_partial` builds
the same structure from the CPU oracle's aggregation states, so a worker's blob can be compared with the oracle's
partial, and a blob written here can be fed to the product's merge."""
import struct

import numpy as np

MAGIC = b"VIYAPS01"
DIM_TYPES = ["string", "numeric", "time", "boolean"]          # db::Column::DimType order (host/viya_db.h)
AGG_TYPES = ["max", "min", "sum", "avg", "count", "bitset"]   # db::Column::AggregationType order


class _Reader:
    def __init__(self, b):
        self.b, self.pos = b, 0

    def take(self, n):
        if self.pos + n > len(self.b):
            raise ValueError("truncated")
        out = self.b[self.pos:self.pos + n]
        self.pos += n
        return out

    def unpack(self, fmt):
        return struct.unpack("<" + fmt, self.take(struct.calcsize("<" + fmt)))

    def align8(self):
        self.take((8 - self.pos % 8) % 8)

    def array(self, n, es):
        a = np.frombuffer(self.take(n * es), dtype=np.dtype("u%d" % es)) if es else np.zeros(0, dtype=np.uint8)
        self.align8()
        return a


def decode(blob: bytes) -> dict:
    r = _Reader(blob)
    magic, ndims, nmetrics, ngroups, has_hidden, _, scanned_recs, scanned_segments, passed_recs, aggregated_recs = r.unpack("8sIIQIIQQQQ")
    assert magic == MAGIC
    p = {"ngroups": ngroups, "has_hidden": has_hidden, "dims": [], "metrics": [], "hidden": None,
         "stats": {"scanned_recs": scanned_recs, "scanned_segments": scanned_segments, "passed_recs": passed_recs,
                   "aggregated_recs": aggregated_recs}}
    for _ in range(ndims):
        dim_type, es, _, name_len, ndict = r.unpack("BBHIQ")
        name = r.take(name_len).decode()
        r.align8()
        keys = r.array(ngroups, es)
        d = {}
        for _ in range(ndict):
            code, ln = r.unpack("QI")
            d[code] = r.take(ln).decode()
            r.align8()
        p["dims"].append({"name": name, "dim_type": DIM_TYPES[dim_type], "es": es, "keys": keys, "dict": d})
    for _ in range(nmetrics):
        agg, es, id_size, _, name_len, npairs = r.unpack("BBBBIQ")
        name = r.take(name_len).decode()
        r.align8()
        m = {"name": name, "agg": AGG_TYPES[agg], "es": es, "id_size": id_size, "npairs": npairs}
        if AGG_TYPES[agg] != "bitset":
            m["states"] = r.array(ngroups, es)
        else:
            m["pair_keys"] = [r.array(npairs, d["es"]) for d in p["dims"]]
            m["ids"] = r.array(npairs, id_size)
        p["metrics"].append(m)
    if has_hidden:
        p["hidden"] = r.array(ngroups, 8)
    assert r.pos == len(blob), "trailing bytes"
    return p


def encode(p: dict) -> bytes:
    out = bytearray()

    def align8():
        out.extend(b"\0" * ((8 - len(out) % 8) % 8))

    def arr(a, es):
        out.extend(np.ascontiguousarray(a).astype(np.dtype("u%d" % es), copy=False).tobytes() if es else b"")
        align8()

    s = p["stats"]
    out += struct.pack("<8sIIQIIQQQQ", MAGIC, len(p["dims"]), len(p["metrics"]), p["ngroups"], 1 if p["hidden"] is not None else 0, 0,
                       s["scanned_recs"], s["scanned_segments"], s["passed_recs"], s["aggregated_recs"])
    for d in p["dims"]:
        name = d["name"].encode()
        out += struct.pack("<BBHIQ", DIM_TYPES.index(d["dim_type"]), d["es"], 0, len(name), len(d["dict"]))
        out += name
        align8()
        arr(d["keys"], d["es"])
        for code, v in d["dict"].items():
            v = v.encode()
            out += struct.pack("<QI", code, len(v)) + v
            align8()
    for m in p["metrics"]:
        name = m["name"].encode()
        out += struct.pack("<BBBBIQ", AGG_TYPES.index(m["agg"]), m["es"], m["id_size"], 0, len(name), m["npairs"])
        out += name
        align8()
        if m["agg"] != "bitset":
            arr(m["states"], m["es"])
        else:
            for d, k in zip(p["dims"], m["pair_keys"]):
                arr(k, d["es"])
            arr(m["ids"], m["id_size"])
    if p["hidden"] is not None:
        arr(p["hidden"], 8)
    return bytes(out)


def _bits(a):
    """Any numeric column as unsigned integers of the same width (bit patterns)."""
    a = np.ascontiguousarray(a)
    return a.view(np.dtype("u%d" % a.dtype.itemsize))


def oracle_partial(odb, q: dict, now=None) -> dict:
    """The partial state the reference-equivalent CPU path holds after the scan of `q` (worker query: no having / sort /
    skip / limit), in the structure decode() returns. Dictionary codes are the oracle database's own."""
    from oracle import viya_oracle as vo
    table = odb.table(q["table"])
    wq = {k: v for k, v in q.items() if k not in ("header", "having", "sort", "skip", "limit")}
    aq = vo.parse_query(table, wq)
    st = vo.scan_aggregate(aq, now)
    n = st.ngroups
    p = {"ngroups": n, "has_hidden": 1 if st.hidden_count is not None else 0, "dims": [], "metrics": [],
         "hidden": None if st.hidden_count is None else st.hidden_count.astype(np.uint64),
         "stats": {"scanned_recs": st.scanned_recs, "scanned_segments": st.scanned_segments, "passed_recs": st.passed_recs,
                   "aggregated_recs": n}}
    for k, oc in enumerate(aq.dim_cols):
        keys = _bits(st.keys[k])
        d = {}
        if oc.col.dim_type == "string":
            c2v = table.dicts[oc.col.name].c2v
            d = {int(c): c2v[int(c)] for c in np.unique(keys)}
        p["dims"].append({"name": oc.col.name, "dim_type": oc.col.dim_type, "es": keys.dtype.itemsize, "keys": keys, "dict": d})
    for k, oc in enumerate(aq.metric_cols):
        m = oc.col
        if m.agg != "bitset":
            s = _bits(st.states[k])
            p["metrics"].append({"name": m.name, "agg": m.agg, "es": s.dtype.itemsize, "id_size": 4, "npairs": 0, "states": s})
            continue
        sets = st.bitsets[k] if st.bitsets and k in st.bitsets else [set() for _ in range(n)]
        gidx = np.array([g for g, s_ in enumerate(sets) for _ in s_], dtype=np.int64)
        id_size = 8 if m.num_type.size == 8 else 4
        ids = np.array([i for s_ in sets for i in sorted(s_)], dtype=np.dtype("u%d" % id_size))
        p["metrics"].append({"name": m.name, "agg": "bitset", "es": 0, "id_size": id_size, "npairs": len(ids),
                             "pair_keys": [d["keys"][gidx] for d in p["dims"]], "ids": ids})
    return p


def canonical(p: dict):
    """Order- and dictionary-independent form: {group key tuple: (states, hidden, {bitset name: frozenset(ids)})}."""
    def key_cols(cols):
        out = []
        for d, col in zip(p["dims"], cols):
            out.append([d["dict"][int(c)] for c in col] if d["dim_type"] == "string" else [int(c) for c in col])
        return list(zip(*out)) if out else None

    n = p["ngroups"]
    gkeys = key_cols([d["keys"] for d in p["dims"]]) or [()] * n
    groups = {}
    for i, gk in enumerate(gkeys):
        states = tuple((m["name"], m["agg"], m["es"], int(m["states"][i])) for m in p["metrics"] if m["agg"] != "bitset")
        groups[gk] = [states, None if p["hidden"] is None else int(p["hidden"][i]), {m["name"]: set() for m in p["metrics"] if m["agg"] == "bitset"}]
    assert len(groups) == n, "duplicate group keys in a partial state"
    for m in p["metrics"]:
        if m["agg"] != "bitset":
            continue
        pk = key_cols(m["pair_keys"]) or [()] * m["npairs"]
        for gk, i in zip(pk, m["ids"]):
            assert int(i) not in groups[gk][2][m["name"]], "duplicate (group, id) pair"
            groups[gk][2][m["name"]].add(int(i))
    return {gk: (v[0], v[1], {k: frozenset(s) for k, s in v[2].items()}) for gk, v in groups.items()}

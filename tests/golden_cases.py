"""Loader/driver for tests/golden/reference_cases.json — usable with any Database-like object
that offers create_table(json) / load(table, rows, now=...) / query(json, now=...) -> (rows, stats),
so the same transcribed reference tests run against the oracle (CPU) and the product host (GPU)."""
from __future__ import annotations

import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
DOC = json.load(open(os.path.join(HERE, "golden", "reference_cases.json")))
CASES = DOC["cases"]
CASE_IDS = [c["id"] for c in CASES]


def case_by_id(cid):
    return CASES[CASE_IDS.index(cid)]


def table_conf(case):
    return DOC["tables"][case["table"]]


def materialise_loads(case):
    """-> list of row batches (one SimpleLoader::Load call each)."""
    conf = table_conf(case)
    seg = int(conf.get("segment_size", 1000000))
    out = []
    for ld in case["loads"]:
        if "rows" in ld:
            rows = DOC["row_sets"][ld["rows"]]
            if "partition_filter" in ld:
                # the loader's partition filter (generated upsert code, src/codegen/db/upsert.cc:316-337): crc32 — util/crc32.h:27-39, the
                # zlib polynomial, chained over the partition columns' INPUT strings — modulo total_partitions must be one of `values`
                import zlib
                pf = ld["partition_filter"]
                names = [d["name"] for d in conf["dimensions"]]
                idx = [names.index(c) for c in pf["columns"]]

                def keep(r):
                    h = 0
                    for i in idx:
                        h = zlib.crc32(r[i].encode(), h)
                    return h % pf["total_partitions"] in pf["values"]
                rows = [r for r in rows if keep(r)]
            out.append(rows)
        elif "inline" in ld:
            out.append(ld["inline"])
        else:
            g = ld["generate"]
            assert g["kind"] == "consecutive_time"
            n = seg + min(seg, 100)
            out.append([[str(g["start"] + i), g["second_column"]] for i in range(n)])
    return out


def materialise_query(case):
    conf = table_conf(case)
    seg = int(conf.get("segment_size", 1000000))

    def fix(f):
        if isinstance(f, dict):
            f = dict(f)
            if "value_expr" in f:
                start = case["loads"][0]["generate"]["start"]
                assert f.pop("value_expr") == "start + segment_size"
                f["value"] = str(start + seg)
            for k in ("filters",):
                if k in f:
                    f[k] = [fix(x) for x in f[k]]
            if "filter" in f and isinstance(f["filter"], dict):
                f["filter"] = fix(f["filter"])
        return f
    q = dict(case["query"])
    if "filter" in q:
        q["filter"] = fix(q["filter"])
    return q


def check_case(case, run):
    """run(table_conf, load_batches, query, now) -> (rows, stats, table_info) or raises."""
    import pytest
    q = materialise_query(case)
    loads = materialise_loads(case)
    now = case.get("now")
    if "throws" in case:
        with pytest.raises(Exception) as ei:
            run(table_conf(case), loads, q, now)
        assert case["throws"] in (getattr(ei.value, "reference_exception", None) or type(ei.value).__name__.lower()
                                   or ""), ei.value
        return
    rows, stats, info = run(table_conf(case), loads, q, now)
    if "rows" in case:
        exp = [list(r) for r in case["rows"]]
        got = [list(r) for r in rows]
        if case.get("sort_values"):
            exp, got = [sorted(r) for r in exp], [sorted(r) for r in got]
        if not case.get("ordered"):
            exp, got = sorted(exp), sorted(got)
        assert got == exp, f"{case['id']} ({case['source']}): got {got} expected {exp}"
    if "nrows" in case:
        assert len(rows) == case["nrows"]
    for k, v in case.get("stats", {}).items():
        if k == "segments":
            assert info["segments"] == v
        elif k == "segment0_size":
            assert info["segment_sizes"][0] == v
        else:
            assert stats[k] == v, f"{case['id']}: stats[{k}] = {stats[k]} != {v}"

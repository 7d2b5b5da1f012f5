"""CPU-side checks of the C++ host shim: it builds, exports its facade, ingests like the oracle,
and refuses to answer a query without the GPU path (no CPU fallback)."""
import pytest

from tests import golden_cases as gc


def _db(conf):
    import __graft_entry__ as g
    g.build()
    from viyadb_amd import hostdb
    return hostdb.Database(conf)


def test_facade_symbols():
    import __graft_entry__ as g
    g.build()
    from viyadb_amd import hostdb
    lib = hostdb.load()
    import os
    import re
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "viya_host.h")).read()
    declared = set(re.findall(r"VDB_API\s+[\w\s\*]+?\b(vdb_\w+)\s*\(", header))
    assert declared and declared == set(hostdb.SYMBOLS), declared ^ set(hostdb.SYMBOLS)
    for s in hostdb.SYMBOLS:
        assert getattr(lib, s) is not None


def test_ingest_segment_accounting_matches_reference_cases():
    """DynamicRollup cases pin upsert semantics: merged rows, one segment, N stored rows."""
    for cid in ("time.DynamicRollup.TimestampIngestion", "time.DynamicRollup.TimestampMicroIngestion",
                "time.DynamicRollup.FormatIngestion"):
        case = gc.case_by_id(cid)
        tconf = gc.table_conf(case)
        db = _db({"tables": [tconf]})
        for batch in gc.materialise_loads(case):
            db.load(tconf["name"], batch, now=case["now"])
        ti = db.table_info(tconf["name"])
        assert ti["segments"] == case["stats"]["segments"]
        assert ti["first_segment_size"] == case["stats"]["segment0_size"]
        db.close()


def test_descriptor_errors_are_invalid_argument():
    from viyadb_amd import hostdb
    case = gc.case_by_id("aggregation.HavingExtraColumn")
    tconf = gc.table_conf(case)
    db = _db({"tables": [tconf]})
    db.load(tconf["name"], gc.materialise_loads(case)[0])
    with pytest.raises(hostdb.HostError) as ei:
        db.query(case["query"])
    assert ei.value.reference_exception == "invalid_argument"
    with pytest.raises(hostdb.HostError):
        db.query({"type": "aggregate", "table": "nope", "dimensions": [], "metrics": []})
    db.close()


def test_query_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by -m gpu tests")
    from viyadb_amd import hostdb
    case = gc.case_by_id("aggregation.BasicQuery")
    tconf = gc.table_conf(case)
    db = _db({"tables": [tconf]})
    db.load(tconf["name"], gc.materialise_loads(case)[0])
    with pytest.raises(hostdb.HostError) as ei:
        db.query(case["query"])
    assert "viya_hip" in str(ei.value) or "hip" in str(ei.value).lower()
    db.close()


# ---- cluster partial states: wire codec and controller-side validation, no GPU needed
_PTCONF = {"name": "events", "dimensions": [{"name": "country"}, {"name": "day", "type": "uint"}],
           "metrics": [{"name": "count", "type": "count"}, {"name": "revenue", "type": "double_sum"},
                       {"name": "avg_len", "type": "long_avg"}, {"name": "users", "type": "bitset"}]}
_PROWS = [["US", "1", "1.5", "10", "7"], ["US", "1", "2.5", "20", "8"], ["IL", "2", "4", "5", "7"], ["US", "3", "1", "1", "9"],
          ["IL", "2", "0.25", "6", "3"]]
_PQ = {"type": "aggregate", "table": "events", "dimensions": ["country"], "metrics": ["revenue", "avg_len", "users", "count"]}


def _oracle_blob():
    from oracle import viya_oracle as vo
    from tests import partial_wire as pw
    odb = vo.Database({"tables": [_PTCONF]})
    odb.table("events").load(_PROWS)
    return pw, odb, pw.oracle_partial(odb, _PQ)


def test_partial_wire_codec_round_trips_the_oracle_partial():
    pw, odb, part = _oracle_blob()
    blob = pw.encode(part)
    assert len(blob) % 8 == 0 and blob[:8] == b"VIYAPS01"
    back = pw.decode(blob)
    assert pw.canonical(back) == pw.canonical(part)
    canon = pw.canonical(part)
    us = canon[("US",)]
    assert us[2]["users"] == frozenset({7, 8, 9})                      # the set itself travels, not its cardinality
    assert dict((n, v) for n, _, _, v in us[0])["avg_len"] == 31       # AVG travels as its sum ...
    assert dict((n, v) for n, _, _, v in us[0])["count"] == 3          # ... next to the count that divides it


def test_merge_rejects_malformed_partials_before_touching_the_device():
    from viyadb_amd import hostdb
    pw, odb, part = _oracle_blob()
    db = _db({"tables": [_PTCONF]})
    try:
        good = pw.encode(part)
        for bad, what in ((b"NOTAPART" + good[8:], "magic"), (good[:len(good) - 16], "truncated"), (good + b"\0" * 8, "trailing")):
            with pytest.raises(hostdb.HostError) as ei:
                db.query_merge(_PQ, [bad])
            assert what in str(ei.value)
        other = dict(_PQ, metrics=["revenue", "avg_len", "users"])         # a different column list than the blob's
        with pytest.raises(hostdb.HostError) as ei:
            db.query_merge(other, [good])
        assert "column list" in str(ei.value)
        swapped = dict(_PQ, metrics=["avg_len", "revenue", "users", "count"])
        with pytest.raises(hostdb.HostError) as ei:
            db.query_merge(swapped, [good])
        assert "does not match metric" in str(ei.value)
        # no partials at all: nothing to merge, the header still goes out (no device work, so this runs without a GPU)
        rows, st = db.query_merge(dict(_PQ, header=True), [])
        assert rows == [["country", "revenue", "avg_len", "users", "count"]] and st["aggregated_recs"] == 0
    finally:
        db.close()


def test_merge_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by -m gpu tests")
    from viyadb_amd import hostdb
    pw, odb, part = _oracle_blob()
    db = _db({"tables": [_PTCONF]})
    try:
        with pytest.raises(hostdb.HostError) as ei:
            db.query_merge(_PQ, [pw.encode(part)])
        assert "hip" in str(ei.value).lower()
    finally:
        db.close()

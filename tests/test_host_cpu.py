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

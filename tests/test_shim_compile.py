"""The drop-in boundary, proven by a compiler: tools/gen_shim_tu.py emits — per table and query, the way
AggQueryGenerator::GenerateCode does (src/codegen/query/agg_query.cc:26-71) — the translation unit that stands where the
reference's JIT-compiled function stands: extern "C" viya_query_agg with the exact query::AggQueryFn signature
(src/query/runner.h:33-35) and the generated per-table `Segment` class (src/codegen/db/store.cc:203-356). Here the text is
checked with `g++ -std=c++17 -fsyntax-only` against the reference's REAL headers — db/table.h, db/dictionary.h, db/store.h,
db/segment.h, query/output.h, query/stats.h — plus a forward declaration standing for <nlohmann/json_fwd.hpp> (the JSON
submodule is not in the mount). Skipped where /root/reference does not exist (the GPU box)."""
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
sys.path.insert(0, os.path.join(ROOT, "tools"))

needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")

TABLES = {
    "inapp": {"name": "events", "dimensions": [{"name": "country"}, {"name": "event_name", "cardinality": 200}, {"name": "install_time", "type": "time"},
                                                {"name": "is_organic", "type": "boolean"}],
              "metrics": [{"name": "count", "type": "count"}, {"name": "revenue", "type": "double_sum"}]},
    "numeric": {"name": "nums", "segment_size": 5000,
                "dimensions": [{"name": "b", "type": "byte"}, {"name": "f", "type": "float"}, {"name": "d", "type": "double"}, {"name": "ul", "type": "ulong"},
                               {"name": "ts", "type": "microtime"}, {"name": "wide", "cardinality": 100000000000}],
                "metrics": [{"name": "count", "type": "count", "max": 100000000000}, {"name": "mx", "type": "short_max"}, {"name": "mn", "type": "ulong_min"},
                            {"name": "s", "type": "float_sum"}]},
    "avg_only": {"name": "avgs", "dimensions": [{"name": "country"}], "metrics": [{"name": "avg_revenue", "type": "double_avg"}]},   # hidden _count array
    # the reference's UserEvents (test/db.h:151-164) + a 64-bit id set: columns of util::Bitset<4> / util::Bitset<8> OBJECTS (store.cc:255-259)
    "user_events": {"name": "user_events", "dimensions": [{"name": "country"}, {"name": "event_name"}, {"name": "time", "type": "uint"}],
                    "metrics": [{"name": "user_id", "type": "bitset"}, {"name": "device_id", "type": "bitset", "max": 100000000000}, {"name": "count", "type": "count"}]},
}
QUERIES = {
    "inapp": {"type": "aggregate", "table": "events", "dimensions": ["event_name", "country"], "metrics": ["revenue", "count"],
              "filter": {"op": "eq", "column": "country", "value": "US"},
              "having": {"op": "and", "filters": [{"op": "gt", "column": "revenue", "value": "1"}, {"op": "ge", "column": "count", "value": "2"}]}},
    "numeric": {"type": "aggregate", "table": "nums", "dimensions": ["b", "f", "d"], "metrics": ["count", "mx"], "filter": {"op": "gt", "column": "count", "value": "0"}},
    "avg_only": {"type": "aggregate", "table": "avgs", "dimensions": ["country"], "metrics": ["avg_revenue"], "filter": {"op": "ge", "column": "avg_revenue", "value": "1"}},
    "user_events": {"type": "aggregate", "table": "user_events", "dimensions": ["country"], "metrics": ["user_id", "device_id"],
                    "filter": {"op": "gt", "column": "time", "value": "1495475514"}},          # test/bitset.cc:32-53
}


def _syntax_check(text, tmp_path, name):
    src = tmp_path / (name + ".cc")
    src.write_text(text)
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-I" + REF, "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "tools", "shim_include"), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[:4000]


@needs_ref
@pytest.mark.parametrize("name", sorted(TABLES))
def test_generated_shim_compiles_against_the_reference_headers(tmp_path, name):
    import gen_shim_tu
    _syntax_check(gen_shim_tu.emit(TABLES[name], QUERIES[name]), tmp_path, name)


@needs_ref
@pytest.mark.parametrize("name", sorted(TABLES))
def test_generated_shim_compiles_to_object_code(tmp_path, name):
    """One step past -fsyntax-only: `g++ -c` of the generated text against the reference's headers, so that the
    generated `Segment` class is laid out, `&segment->d._i[0]` / `&segment->m._j[0]` are real address computations and the call into
    viya::shim is an unresolved symbol of the right mangled name — the object must define `viya_query_agg` and only LACK what
    libviya_host.so (shim) and the reference's own libraries (db::Table, Dictionary, ...) provide. It has still never been LINKED behind
    QueryRunner::Visit: the reference cannot be built here (INTEGRATION.md says so)."""
    import gen_shim_tu
    src = tmp_path / (name + ".cc")
    obj = tmp_path / (name + ".o")
    src.write_text(gen_shim_tu.emit(TABLES[name], QUERIES[name]))
    r = subprocess.run(["g++", "-std=c++17", "-O2", "-fPIC", "-c", "-Wall", "-I" + REF, "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "tools", "shim_include"), str(src), "-o", str(obj)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[:4000]
    syms = subprocess.run(["nm", "-C", str(obj)], capture_output=True, text=True, check=True).stdout
    assert re.search(r"\bT viya_query_agg\b", syms), syms[:2000]
    undefined = [l.split(None, 1)[1] for l in syms.splitlines() if l.strip().startswith("U ")]
    assert any(u.startswith("viya::shim::Run(") for u in undefined) and any(u.startswith("viya::shim::Sync(") for u in undefined), undefined
    host = subprocess.run(["nm", "-DC", os.path.join(ROOT, "viyadb_amd", "libviya_host.so")], capture_output=True, text=True).stdout
    for u in undefined:
        if u.startswith("viya::shim::"):
            assert u.split("(")[0] + "(" in host, u          # every shim entry point the object calls is exported by the host library
    if name == "user_events":      # the bitset columns are walked: the Roaring sets' own calls stay unresolved, the CSR hand-over is a shim call
        assert any(u.startswith("viya::shim::SyncBitset(") for u in undefined) and any(u.startswith("viya::shim::BitsetStale(") for u in undefined), undefined
        assert any("Roaring::toUint32Array" in u for u in undefined) and any("Roaring64Map::toUint64Array" in u for u in undefined), undefined


@needs_ref
def test_signature_is_the_one_the_reference_emits():
    """The declaration text of the swap point, character for character (modulo whitespace): agg_query.cc:35-44."""
    import gen_shim_tu
    ref = open(os.path.join(REF, "codegen/query/agg_query.cc")).read()
    pieces = re.findall(r'code << ((?:"(?:[^"\\]|\\.)*"\s*)+);', ref)
    emitted = ["".join(re.findall(r'"((?:[^"\\]|\\.)*)"', p)).replace('\\"', '"').replace("\\n", "\n") for p in pieces]
    sigs = [e for e in emitted if "viya_query_agg" in e]
    assert len(sigs) == 2
    text = gen_shim_tu.emit(TABLES["inapp"], QUERIES["inapp"])
    norm = lambda s: re.sub(r"\s+", "", s)
    for s in sigs:
        assert norm(s) in norm(text), s
    # and the function type it is called through (runner.h:33-35) has the same parameter list
    runner = open(os.path.join(REF, "query/runner.h")).read()
    m = re.search(r"using AggQueryFn = void \(\*\)\(([^;]*)\);", runner, re.S)
    params = [norm(x) for x in m.group(1).split(",")]
    assert params == ["db::Table&", "RowOutput&", "QueryStats&", "std::vector<db::AnyNum>", "size_t", "size_t", "std::vector<db::AnyNum>"]


@pytest.mark.parametrize("name", sorted(TABLES))
def test_cxx_emitter_writes_what_the_python_generator_writes(name):
    """The emitter a maintainer links — viya::shim::codegen::AggQueryText in libviya_host.so, a C++ function returning the text like
    codegen::Code (src/codegen/generator.h:77-97) — against tools/gen_shim_tu.py, its specification: character for character. The C++ side
    takes its column types from the descriptor-only db::Table it parses; the Python side restates the rules (max_value_to_uint_type,
    parse_value_metric_type: src/db/column.cc:54-62,275-286): two derivations of one text."""
    import gen_shim_tu
    from viyadb_amd import hostdb
    tj, qj = json.dumps(TABLES[name]), json.dumps(QUERIES[name])
    assert hostdb.shim_text(tj, qj) == gen_shim_tu.emit(TABLES[name], QUERIES[name])
    assert hostdb.shim_text(None, None) == gen_shim_tu.upsert_hook()


@needs_ref
def test_upsert_hook_compiles_where_the_reference_would_emit_it(tmp_path):
    """UpsertGenerator writes the in-place branch of viya_upsert_do as text (src/codegen/db/upsert.cc:384-396):
    `static_cast<Segment*>(segments[segment_idx])->m.Update(upsert_tuple.m,tuple_idx);`. The hook line goes right behind it. Here that
    branch is rebuilt around the REAL db/table.h / db/store.h / db/segment.h — `lctx->table` is a db::Table*, `segments` the store's
    vector, the Segment class the generated one — and compiled to an object: the names the line uses exist there with these types, and
    the call resolves to viya::shim::Touch(const void*, unsigned, unsigned long, unsigned long)."""
    import gen_shim_tu
    table = TABLES["user_events"]
    text = gen_shim_tu.emit(table, QUERIES["user_events"])
    classes = text[text.index("struct Tuple {"):text.index("static const char kTable[]")]       # Tuple / SegmentStats / Segment, as StoreDefs emits them
    src = tmp_path / "upsert_hook.cc"
    src.write_text("#include <vector>\n#include <cstdint>\n#include <cstddef>\n#include <cfloat>\n#include <algorithm>\n#include <db/table.h>\n#include <db/store.h>\n#include <db/segment.h>\n"
                   "#include <util/bitset.h>\n#include <viya_shim.h>\nnamespace db = viya::db;\nnamespace util = viya::util;\n"
                   "struct LoaderContext { db::Table* table; };                                   // (input/loader: lctx->table)\n"
                   + classes +
                   "void viya_upsert_do_update_branch(LoaderContext* lctx, Tuple& upsert_tuple, size_t global_idx) {\n"
                   " auto* store = lctx->table->store();\n auto& segments = store->segments();\n"
                   " size_t segment_idx = global_idx / %d;\n size_t tuple_idx = global_idx %% %d;\n"
                   " (void)upsert_tuple; (void)static_cast<Segment*>(segments[segment_idx]);\n" % (1000000, 1000000)
                   + gen_shim_tu.upsert_hook() + "}\n")
    obj = tmp_path / "upsert_hook.o"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-fPIC", "-c", "-Wall", "-I" + REF, "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "tools", "shim_include"), str(src), "-o", str(obj)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[:4000]
    syms = subprocess.run(["nm", "-C", str(obj)], capture_output=True, text=True, check=True).stdout
    assert "U viya::shim::Touch(void const*, unsigned int, unsigned long, unsigned long)" in syms, syms[:1500]


@needs_ref
def test_every_reference_test_table_gets_a_compilable_shim(tmp_path):
    """The tables and aggregate queries of the reference's own known-answer tests (tests/golden/reference_cases.json), bitset tables
    included (util/bitset.h is the reference's; CRoaring's two classes are declared — not defined — in tools/shim_include/)."""
    import gen_shim_tu
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_cases.json")))
    tables = cases["tables"] if isinstance(cases, dict) and "tables" in cases else {}
    done = 0
    seen = set()
    for case in (cases["cases"] if isinstance(cases, dict) else cases):
        q = case.get("query") or {}
        t = case.get("table") if isinstance(case.get("table"), dict) else tables.get(case.get("table") or q.get("table"))
        if not isinstance(t, dict) or q.get("type") != "aggregate":
            continue
        key = json.dumps([t, q], sort_keys=True)
        if key in seen:
            continue
        seen.add(key)
        _syntax_check(gen_shim_tu.emit(t, q), tmp_path, "case%d" % done)
        done += 1
        if done >= 16:
            break
    assert done >= 5, done


def test_shim_header_is_plain_cxx_without_reference_types():
    """include/viya_shim.h must compile on its own (no reference header, no HIP): it is included by generated code."""
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-x", "c++", os.path.join(ROOT, "include", "viya_shim.h")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lib = os.path.join(ROOT, "viyadb_amd", "libviya_host.so")
    syms = subprocess.run(["nm", "-DC", lib], capture_output=True, text=True).stdout
    for fn in ("viya::shim::Open(", "viya::shim::Sync(", "viya::shim::Touch(", "viya::shim::BindDict(", "viya::shim::Run(", "viya::shim::Close(",
               "viya::shim::BitsetStale(", "viya::shim::SyncBitset(", "viya::shim::codegen::AggQueryText(", "viya::shim::codegen::UpsertHookText"):
        assert fn in syms, fn

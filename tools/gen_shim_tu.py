#!/usr/bin/env python3
"""Emit the translation unit that stands where the reference's JIT-compiled aggregate function stands.

The reference builds one C++ function per (table, query shape): AggQueryGenerator::GenerateCode writes its text
(src/codegen/query/agg_query.cc:26-71) — headers, the extern "C" signature of query::AggQueryFn (src/query/runner.h:33-35),
the AggTuple struct, the per-table Tuple / SegmentStats / Segment classes (StoreDefs, src/codegen/db/store.cc:203-356), the
scan loop (ScanVisitor) and the post-aggregation (PostAggVisitor) — and Compiler builds it with g++ at run time.

This generator writes the text a ViyaDB maintainer's `AggQueryGenerator::GenerateCode` would emit for the GPU path: the SAME
signature and the SAME Segment class, so that `static_cast<Segment*>(s)` is valid and column addresses are taken exactly as
the reference takes them (`&segment->d._i[0]`, `&segment->m._j[0]`), but the scan loop and the post-aggregation are calls
into libviya_host (include/viya_shim.h), which drives the C ABI of libviya_hip (include/viya_hip.h).

It only needs the two JSON descriptors the reference itself starts from. tests/test_shim_compile.py runs
`g++ -std=c++17 -fsyntax-only` on the output against the reference's REAL headers (db/table.h, db/dictionary.h, db/store.h,
db/segment.h, query/output.h, query/stats.h) wherever /root/reference exists.

usage: gen_shim_tu.py table.json query.json [out.cc]"""
import json
import sys

CPP = {"byte": "int8_t", "ubyte": "uint8_t", "short": "int16_t", "ushort": "uint16_t", "int": "int32_t", "uint": "uint32_t",
       "long": "int64_t", "ulong": "uint64_t", "float": "float", "double": "double"}
CPP_MIN = {"byte": "INT8_MIN", "ubyte": "0U", "short": "INT16_MIN", "ushort": "0U", "int": "INT32_MIN", "uint": "0U",
           "long": "INT64_MIN", "ulong": "0UL", "float": "FLT_MIN", "double": "DBL_MIN"}      # NumericType::cpp_min_value (src/db/column.cc)
CPP_MAX = {"byte": "INT8_MAX", "ubyte": "UINT8_MAX", "short": "INT16_MAX", "ushort": "UINT16_MAX", "int": "INT32_MAX",
           "uint": "UINT32_MAX", "long": "INT64_MAX", "ulong": "UINT64_MAX", "float": "FLT_MAX", "double": "DBL_MAX"}


def uint_for_max(max_value):
    """max_value_to_uint_type (src/db/column.cc:54-62)."""
    m = (int(max_value) - 1) & 0xFFFFFFFFFFFFFFFF
    return "ubyte" if m < 0xFF else "ushort" if m < 0xFFFF else "uint" if m < 0xFFFFFFFF else "ulong"


def dim_type(d):
    t = d.get("type", "string")
    if t == "string":
        return uint_for_max(d.get("cardinality", 0xFFFFFFFF))       # StrDimension (column.cc:288-300)
    if t == "boolean":
        return "ubyte"
    if t == "time":
        return "uint"
    if t == "microtime":
        return "ulong"
    if t == "numeric":
        return uint_for_max(d.get("max", 0xFFFFFFFF))               # deprecated spelling of a uint dimension
    return t


def metric_type(m):
    t = m["type"]
    if t == "count":
        return "ulong" if uint_for_max(m.get("max", 0xFFFFFFFF)) == "ulong" else "uint"   # parse_value_metric_type
    if t == "bitset":
        return "ulong" if uint_for_max(m.get("max", 0xFFFFFFFF)) == "ulong" else "uint"   # the ids' type: util::Bitset<8> / util::Bitset<4>
    return t.split("_")[0]


def metric_agg(m):
    t = m["type"]
    return t if t in ("count", "bitset") else t.split("_", 1)[1]


def emit(table, query):
    dims, mets = table["dimensions"], table["metrics"]
    size = int(table.get("segment_size", 1000000))
    has_avg = any(metric_agg(m) == "avg" for m in mets)
    has_count = any(metric_agg(m) == "count" for m in mets)
    hidden = has_avg and not has_count
    o = []
    w = o.append
    bitsets = [j for j, m in enumerate(mets) if metric_agg(m) == "bitset"]
    w("// GENERATED: the GPU-path body of AggQueryGenerator::GenerateCode (src/codegen/query/agg_query.cc:26-71) — viya::shim::codegen::AggQueryText / tools/gen_shim_tu.py\n")
    for h in ("unordered_map", "vector", "string", "stdexcept", "cstdio", "cstdint", "cstddef", "cfloat", "algorithm"):
        w("#include <%s>\n" % h)
    for h in ("query/output.h", "query/stats.h", "db/table.h", "db/dictionary.h", "db/store.h", "db/segment.h"):   # agg_query.cc:28-30, store.cc:205
        w("#include <%s>\n" % h)
    if bitsets:
        w("#include <util/bitset.h>                                                 // store.cc:255-259\n")
    w("#include <viya_shim.h>   // libviya_host: mirror sync, plan, vh_query_agg, post-aggregation\n")
    w("namespace db = viya::db;\nnamespace query = viya::query;\nnamespace util = viya::util;\n\n")
    if bitsets:
        # util::Bitset keeps its Roaring private and offers no iteration (src/util/bitset.h:26-67): the ids are read through a pointer to
        # that member obtained by explicit instantiation — the one place the language lets a private member be named from outside —, so
        # that the reference's header stays as it is. (A maintainer may prefer a `const RoaringType& roaring() const` accessor there.)
        w("namespace viya_shim_detail {\n")
        w("template <class Tag, typename Tag::type M> struct Expose { friend typename Tag::type get(Tag) { return M; } };\n")
        for n, rt in ((4, "Roaring"), (8, "Roaring64Map")):
            if any(CPP[metric_type(mets[j])] == ("uint64_t" if n == 8 else "uint32_t") for j in bitsets):
                w("struct Roaring%d { typedef %s util::Bitset<%d>::*type; friend type get(Roaring%d); };\n" % (n, rt, n, n))
                w("template struct Expose<Roaring%d, &util::Bitset<%d>::roaring_>;\n" % (n, n))
        w("}  // namespace viya_shim_detail\n\n")
    sig = ("extern \"C\" void viya_query_agg(db::Table& table, query::RowOutput& output, query::QueryStats& stats,"
           "std::vector<db::AnyNum> fargs, size_t skip, size_t limit, std::vector<db::AnyNum> hargs)")
    w(sig + " __attribute__((__visibility__(\"default\")));\n")                    # agg_query.cc:35-39
    w(sig + " {\n")                                                                # agg_query.cc:41-44
    # ---- StoreDefs (store.cc:203-356): the data members of the classes the store was built with; layout must be identical
    w("struct Tuple {\n struct Dimensions {\n")
    for i, d in enumerate(dims):
        w("  %s _%d;\n" % (CPP[dim_type(d)], i))
    w(" };\n struct Metrics {\n")
    for j, m in enumerate(mets):
        if metric_agg(m) == "bitset":
            w("  util::Bitset<%d> _%d;\n" % (8 if metric_type(m) == "ulong" else 4, j))
        else:
            w("  %s _%d;\n" % (CPP[metric_type(m)], j))
    if hidden:
        w("  uint64_t _count;\n")
    w(" };\n Dimensions d; Metrics m;\n};\n")
    w("struct SegmentStats {\n")                                                    # store.cc:171-201: NUMERIC and TIME dimensions only
    for i, d in enumerate(dims):
        if d.get("type", "string") not in ("string", "boolean"):
            t = dim_type(d)
            w(" %s dmax%d = %s; %s dmin%d = %s;\n" % (CPP[t], i, CPP_MIN[t], CPP[t], i, CPP_MAX[t]))
    w("};\n")
    w("class Segment: public db::SegmentBase {\npublic:\n struct Dimensions {\n")
    for i, d in enumerate(dims):
        w("  %s _%d[%d];\n" % (CPP[dim_type(d)], i, size))
    w(" };\n struct Metrics {\n")
    fills = []
    for j, m in enumerate(mets):
        t, a = metric_type(m), metric_agg(m)
        if a == "bitset":
            w("  util::Bitset<%d> _%d[%d];\n" % (8 if t == "ulong" else 4, j, size))
        elif a == "max":
            w("  %s _%d[%d];\n" % (CPP[t], j, size)); fills.append("std::fill_n(_%d,%d,%s);" % (j, size, CPP_MIN[t]))
        elif a == "min":
            w("  %s _%d[%d];\n" % (CPP[t], j, size)); fills.append("std::fill_n(_%d,%d,%s);" % (j, size, CPP_MAX[t]))
        else:
            w("  %s _%d[%d] = {0};\n" % (CPP[t], j, size))
    if hidden:
        w("  uint64_t _count[%d] = {0};\n" % size)
    w("  Metrics() { %s }\n };\n Dimensions d; Metrics m; SegmentStats stats;\n Segment():SegmentBase(%d) {}\n};\n" % (" ".join(fills), size))
    # ---- the GPU path instead of ScanVisitor + PostAggVisitor
    w("static const char kTable[] = R\"viya(%s)viya\";\n" % json.dumps(table))
    w("static const char kQuery[] = R\"viya(%s)viya\";\n" % json.dumps(query))
    w("viya::shim::Session* session = viya::shim::Open(&table, kTable, kQuery);   // this CALL's state: read_pool threads run this function side by side\n")
    w("struct SessionGuard { viya::shim::Session* s; ~SessionGuard() { viya::shim::Release(s); } } session_guard{session};\n")
    w("uint32_t seg_index = 0;\n")
    w("for (auto* s : table.store()->segments_copy()) {                      // scan.cc:42\n")
    w(" auto segment_size = s->size();                                       // scan.cc:43: the size() snapshot the query sees\n")
    w(" auto segment = static_cast<Segment*>(s);\n")
    ptrs = ["&segment->d._%d[0]" % i for i in range(len(dims))] + \
           ["nullptr" if j in bitsets else "&segment->m._%d[0]" % j for j in range(len(mets))]     # (a bitset column is not an array of numbers: below)
    if hidden:
        ptrs.append("&segment->m._count[0]")
    w(" const void* cols[] = { %s };\n" % ", ".join(ptrs))
    w(" viya::shim::Pin(session, seg_index, segment, sizeof(Segment));          // registered with the device once: later ranges are read in place\n")
    w(" viya::shim::Sync(session, seg_index, segment_size, cols);\n")
    if bitsets:
        w(" if (const uint64_t walk_rows = viya::shim::BitsetStale(session, seg_index, segment_size)) {   // rows appended, or a row's set grown in place (Touch), since the mirror saw them\n")
        w("  std::vector<uint64_t> offsets(walk_rows + 1);\n")
        for j in bitsets:
            n = 8 if metric_type(mets[j]) == "ulong" else 4
            w("  { std::vector<uint%d_t> ids; offsets[0] = 0;\n" % (n * 8))
            w("    for (size_t r = 0; r < walk_rows; ++r) {\n")
            w("      const auto& roaring = segment->m._%d[r].*get(viya_shim_detail::Roaring%d());\n" % (j, n))
            w("      const uint64_t n = roaring.cardinality();\n")
            w("      ids.resize(offsets[r] + n);\n")
            w("      if (n) roaring.toUint%dArray(ids.data() + offsets[r]);\n" % (n * 8))
            w("      offsets[r + 1] = offsets[r] + n;\n    }\n")
            w("    viya::shim::SyncBitset(session, seg_index, %d, walk_rows, offsets.data(), ids.data()); }\n" % j)
        w(" }\n")
    w(" ++seg_index;\n}\n")
    for i, d in enumerate(dims):
        if d.get("type", "string") == "string":
            w("{ auto dict%d = static_cast<const db::StrDimension*>(table.dimension(%d))->dict();      // post_agg.cc:32-40\n" % (i, i))
            w("  dict%d->lock().lock_shared(); viya::shim::BindDict(session, %d, &dict%d->c2v()); dict%d->lock().unlock_shared(); }\n" % (i, i, i, i))
    w("std::vector<uint64_t> fa, ha;\n")
    w("for (auto& a : fargs) fa.push_back(a.get_uint64_t());                  // db::AnyNum: the column's own type in the low bytes\n")
    w("for (auto& a : hargs) ha.push_back(a.get_uint64_t());\n")
    w("viya::shim::Stats st{};\n")
    w("output.Start();                                                        // post_agg.cc:30\n")
    w("viya::shim::Run(session, fa.data(), fa.size(), ha.data(), ha.size(), skip, limit,\n")
    w("  [](void* ctx, const std::vector<std::string>& row) { static_cast<query::RowOutput*>(ctx)->Send(row); }, &output, &st);\n")
    w("output.Flush();                                                        // post_agg.cc:146\n")
    w("stats.scanned_recs += st.scanned_recs; stats.scanned_segments += st.scanned_segments;   // scan.cc:44,51\n")
    w("stats.aggregated_recs = st.aggregated_recs; stats.output_recs += st.output_recs;         // scan.cc:246, post_agg.cc:137\n")
    w("}\n")
    return "".join(o)


def upsert_hook():
    """The line UpsertGenerator adds behind `static_cast<Segment*>(segments[segment_idx])->m.Update(upsert_tuple.m,tuple_idx);`
    (src/codegen/db/upsert.cc:384-396): the in-place branch of viya_upsert_do tells the mirror which row changed; the append branch
    needs nothing (Sync copies what lies beyond the rows it has seen)."""
    return "  viya::shim::Touch(lctx->table, segment_idx, tuple_idx, tuple_idx + 1);\n"


def main():
    table, query = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
    text = emit(table, query)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()

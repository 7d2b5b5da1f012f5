// shim_codegen.cc — viya::shim::codegen (include/viya_shim.h): the text a ViyaDB maintainer's generators emit at the two swap points.
//
// The reference writes one C++ function per (table, query shape): AggQueryGenerator::GenerateCode returns its text as a codegen::Code
// (src/codegen/query/agg_query.cc:26-75, src/codegen/generator.h:77-97) — headers, the extern "C" signature of query::AggQueryFn
// (src/query/runner.h:33-35), the table's Tuple / SegmentStats / Segment classes (StoreDefs, src/codegen/db/store.cc:203-356), then the
// scan loop and the post-aggregation. AggQueryText() is that function for the GPU path: the SAME signature and the SAME Segment class —
// `static_cast<Segment*>(s)` must stay valid, column addresses are taken as the reference takes them (`&segment->d._i[0]`) — and, where
// the loops stood, calls into viya::shim (Open / Sync / BitsetStale / SyncBitset / BindDict / Run). A maintainer's GenerateCode becomes
//     Code code; code << viya::shim::codegen::AggQueryText(table_json, query_json); return code;
// tools/gen_shim_tu.py is the same generator in Python; tests/test_shim_compile.py holds the two against each other character for character
// and compiles the text against the reference's real headers. Column types come from the descriptor-only db::Table the host shim parses
// from the same JSON (viya_db.h), so a type rule lives in one place (max_value_to_uint_type, parse_value_metric_type: src/db/column.cc).
#include <string>

#include "../../include/viya_shim.h"
#include "viya_db.h"

namespace viya {
namespace shim {
namespace codegen {

namespace {
const char* min_literal(db::Num t) {      // NumericType::cpp_min_value as the reference SPELLS it (src/db/column.cc:189-221)
  switch (t) {
    case db::Num::BYTE: return "INT8_MIN"; case db::Num::SHORT: return "INT16_MIN"; case db::Num::INT: return "INT32_MIN"; case db::Num::LONG: return "INT64_MIN";
    case db::Num::ULONG: return "0UL"; case db::Num::FLOAT: return "FLT_MIN"; case db::Num::DOUBLE: return "DBL_MIN";
    default: return "0U";
  }
}
const char* max_literal(db::Num t) {
  switch (t) {
    case db::Num::BYTE: return "INT8_MAX"; case db::Num::UBYTE: return "UINT8_MAX"; case db::Num::SHORT: return "INT16_MAX"; case db::Num::USHORT: return "UINT16_MAX";
    case db::Num::INT: return "INT32_MAX"; case db::Num::UINT: return "UINT32_MAX"; case db::Num::LONG: return "INT64_MAX"; case db::Num::ULONG: return "UINT64_MAX";
    case db::Num::FLOAT: return "FLT_MAX"; default: return "DBL_MAX";
  }
}
std::string num(size_t v) { return std::to_string(v); }
}  // namespace

std::string AggQueryText(const std::string& table_json, const std::string& query_json) {
  db::Dictionaries dicts;
  const db::Table table(util::Config(table_json), dicts);      // descriptors only
  const auto& dims = table.dimensions();
  const auto& mets = table.metrics();
  const std::string size = num(table.segment_size());
  const bool hidden = table.has_hidden_count();
  bool any_bitset = false, bitset4 = false, bitset8 = false;
  for (auto* m : mets) if (m->agg_type() == db::Column::BITSET) { any_bitset = true; (m->num_type().size() == 8 ? bitset8 : bitset4) = true; }
  auto bits_of = [](const db::Metric* m) { return m->num_type().size() == 8 ? 8 : 4; };
  std::string o;
  o += "// GENERATED: the GPU-path body of AggQueryGenerator::GenerateCode (src/codegen/query/agg_query.cc:26-71) — viya::shim::codegen::AggQueryText / tools/gen_shim_tu.py\n";
  for (const char* h : {"unordered_map", "vector", "string", "stdexcept", "cstdio", "cstdint", "cstddef", "cfloat", "algorithm"}) o += std::string("#include <") + h + ">\n";
  for (const char* h : {"query/output.h", "query/stats.h", "db/table.h", "db/dictionary.h", "db/store.h", "db/segment.h"}) o += std::string("#include <") + h + ">\n";
  if (any_bitset) o += "#include <util/bitset.h>                                                 // store.cc:255-259\n";
  o += "#include <viya_shim.h>   // libviya_host: mirror sync, plan, vh_query_agg, post-aggregation\n";
  o += "namespace db = viya::db;\nnamespace query = viya::query;\nnamespace util = viya::util;\n\n";
  if (any_bitset) {
    o += "namespace viya_shim_detail {\n";
    o += "template <class Tag, typename Tag::type M> struct Expose { friend typename Tag::type get(Tag) { return M; } };\n";
    for (int n : {4, 8}) {
      if (!(n == 4 ? bitset4 : bitset8)) continue;
      const std::string N = num(n), rt = n == 8 ? "Roaring64Map" : "Roaring";
      o += "struct Roaring" + N + " { typedef " + rt + " util::Bitset<" + N + ">::*type; friend type get(Roaring" + N + "); };\n";
      o += "template struct Expose<Roaring" + N + ", &util::Bitset<" + N + ">::roaring_>;\n";
    }
    o += "}  // namespace viya_shim_detail\n\n";
  }
  const std::string sig = "extern \"C\" void viya_query_agg(db::Table& table, query::RowOutput& output, query::QueryStats& stats,"
                          "std::vector<db::AnyNum> fargs, size_t skip, size_t limit, std::vector<db::AnyNum> hargs)";
  o += sig + " __attribute__((__visibility__(\"default\")));\n";
  o += sig + " {\n";
  // ---- StoreDefs (store.cc:203-356): the data members of the classes the store was built with; layout must be identical
  o += "struct Tuple {\n struct Dimensions {\n";
  for (auto* d : dims) o += "  " + d->num_type().cpp_type() + " _" + num(d->index()) + ";\n";
  o += " };\n struct Metrics {\n";
  for (auto* m : mets) {
    if (m->agg_type() == db::Column::BITSET) o += "  util::Bitset<" + num(bits_of(m)) + "> _" + num(m->index()) + ";\n";
    else o += "  " + m->num_type().cpp_type() + " _" + num(m->index()) + ";\n";
  }
  if (hidden) o += "  uint64_t _count;\n";
  o += " };\n Dimensions d; Metrics m;\n};\n";
  o += "struct SegmentStats {\n";                                   // store.cc:171-201: NUMERIC and TIME dimensions only
  for (auto* d : dims) {
    if (d->dim_type() == db::Column::DIM_STRING || d->dim_type() == db::Column::DIM_BOOLEAN) continue;
    const std::string T = d->num_type().cpp_type(), i = num(d->index());
    o += " " + T + " dmax" + i + " = " + min_literal(d->num_type().type()) + "; " + T + " dmin" + i + " = " + max_literal(d->num_type().type()) + ";\n";
  }
  o += "};\n";
  o += "class Segment: public db::SegmentBase {\npublic:\n struct Dimensions {\n";
  for (auto* d : dims) o += "  " + d->num_type().cpp_type() + " _" + num(d->index()) + "[" + size + "];\n";
  o += " };\n struct Metrics {\n";
  std::string fills;
  for (auto* m : mets) {
    const std::string T = m->num_type().cpp_type(), j = num(m->index());
    auto fill = [&](const char* lit) { if (!fills.empty()) fills += " "; fills += "std::fill_n(_" + j + "," + size + "," + lit + ");"; };
    if (m->agg_type() == db::Column::BITSET) o += "  util::Bitset<" + num(bits_of(m)) + "> _" + j + "[" + size + "];\n";
    else if (m->agg_type() == db::Column::MAX) { o += "  " + T + " _" + j + "[" + size + "];\n"; fill(min_literal(m->num_type().type())); }
    else if (m->agg_type() == db::Column::MIN) { o += "  " + T + " _" + j + "[" + size + "];\n"; fill(max_literal(m->num_type().type())); }
    else o += "  " + T + " _" + j + "[" + size + "] = {0};\n";
  }
  if (hidden) o += "  uint64_t _count[" + size + "] = {0};\n";
  o += "  Metrics() { " + fills + " }\n };\n Dimensions d; Metrics m; SegmentStats stats;\n Segment():SegmentBase(" + size + ") {}\n};\n";
  // ---- the GPU path instead of ScanVisitor + PostAggVisitor
  o += "static const char kTable[] = R\"viya(" + table_json + ")viya\";\n";
  o += "static const char kQuery[] = R\"viya(" + query_json + ")viya\";\n";
  o += "viya::shim::Session* session = viya::shim::Open(&table, kTable, kQuery);   // this CALL's state: read_pool threads run this function side by side\n";
  o += "struct SessionGuard { viya::shim::Session* s; ~SessionGuard() { viya::shim::Release(s); } } session_guard{session};\n";
  o += "uint32_t seg_index = 0;\n";
  o += "for (auto* s : table.store()->segments_copy()) {                      // scan.cc:42\n";
  o += " auto segment_size = s->size();                                       // scan.cc:43: the size() snapshot the query sees\n";
  o += " auto segment = static_cast<Segment*>(s);\n";
  std::string ptrs;
  for (auto* d : dims) { if (!ptrs.empty()) ptrs += ", "; ptrs += "&segment->d._" + num(d->index()) + "[0]"; }
  for (auto* m : mets) { if (!ptrs.empty()) ptrs += ", "; ptrs += m->agg_type() == db::Column::BITSET ? std::string("nullptr") : "&segment->m._" + num(m->index()) + "[0]"; }
  if (hidden) ptrs += ", &segment->m._count[0]";
  o += " const void* cols[] = { " + ptrs + " };\n";
  o += " viya::shim::Pin(session, seg_index, segment, sizeof(Segment));          // registered with the device once: later ranges are read in place\n";
  o += " viya::shim::Sync(session, seg_index, segment_size, cols);\n";
  if (any_bitset) {
    o += " if (const uint64_t walk_rows = viya::shim::BitsetStale(session, seg_index, segment_size)) {   // rows appended, or a row's set grown in place (Touch), since the mirror saw them\n";
    o += "  std::vector<uint64_t> offsets(walk_rows + 1);\n";
    for (auto* m : mets) {
      if (m->agg_type() != db::Column::BITSET) continue;
      const int n = bits_of(m);
      const std::string j = num(m->index());
      o += "  { std::vector<uint" + num(n * 8) + "_t> ids; offsets[0] = 0;\n";
      o += "    for (size_t r = 0; r < walk_rows; ++r) {\n";
      o += "      const auto& roaring = segment->m._" + j + "[r].*get(viya_shim_detail::Roaring" + num(n) + "());\n";
      o += "      const uint64_t n = roaring.cardinality();\n";
      o += "      ids.resize(offsets[r] + n);\n";
      o += "      if (n) roaring.toUint" + num(n * 8) + "Array(ids.data() + offsets[r]);\n";
      o += "      offsets[r + 1] = offsets[r] + n;\n    }\n";
      o += "    viya::shim::SyncBitset(session, seg_index, " + j + ", walk_rows, offsets.data(), ids.data()); }\n";
    }
    o += " }\n";
  }
  o += " ++seg_index;\n}\n";
  for (auto* d : dims) {
    if (d->dim_type() != db::Column::DIM_STRING) continue;
    const std::string i = num(d->index());
    o += "{ auto dict" + i + " = static_cast<const db::StrDimension*>(table.dimension(" + i + "))->dict();      // post_agg.cc:32-40\n";
    o += "  dict" + i + "->lock().lock_shared(); viya::shim::BindDict(session, " + i + ", &dict" + i + "->c2v()); dict" + i + "->lock().unlock_shared(); }\n";
  }
  o += "std::vector<uint64_t> fa, ha;\n";
  o += "for (auto& a : fargs) fa.push_back(a.get_uint64_t());                  // db::AnyNum: the column's own type in the low bytes\n";
  o += "for (auto& a : hargs) ha.push_back(a.get_uint64_t());\n";
  o += "viya::shim::Stats st{};\n";
  o += "output.Start();                                                        // post_agg.cc:30\n";
  o += "viya::shim::Run(session, fa.data(), fa.size(), ha.data(), ha.size(), skip, limit,\n";
  o += "  [](void* ctx, const std::vector<std::string>& row) { static_cast<query::RowOutput*>(ctx)->Send(row); }, &output, &st);\n";
  o += "output.Flush();                                                        // post_agg.cc:146\n";
  o += "stats.scanned_recs += st.scanned_recs; stats.scanned_segments += st.scanned_segments;   // scan.cc:44,51\n";
  o += "stats.aggregated_recs = st.aggregated_recs; stats.output_recs += st.output_recs;         // scan.cc:246, post_agg.cc:137\n";
  o += "}\n";
  return o;
}

std::string UpsertHookText() {
  return "  viya::shim::Touch(lctx->table, segment_idx, tuple_idx, tuple_idx + 1);\n";
}

}  // namespace codegen
}  // namespace shim
}  // namespace viya

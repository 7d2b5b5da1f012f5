/*
 * viya_shim.h — what the GENERATED viya_query_agg translation unit links against (libviya_host.so).
 *
 * The reference swaps in one C++ function per (table, query shape): AggQueryGenerator::GenerateCode emits its text
 * (src/codegen/query/agg_query.cc:26-71), Compiler builds it with g++ and QueryRunner calls it through
 * query::AggQueryFn (src/query/runner.h:33-35, runner.cc:45-64). tools/gen_shim_tu.py emits the replacement text for
 * that swap point: the SAME extern "C" signature, the SAME per-table `Segment` class (StoreDefs, src/codegen/db/store.cc:
 * 203-356) so that column addresses are taken from the generated class (`&segment->d._i[0]`), and — instead of the row
 * loop, the std::unordered_map and the post-aggregation text — calls into this header:
 *
 *   Open      once per (table, query text): descriptors parsed, HBM mirror created (the analogue of compile + cache)
 *   Sync      per segment of table.store()->segments_copy(): rows appended since the last call are copied to HBM
 *   Touch     from the upsert path when metrics of EXISTING rows change in place (src/codegen/db/upsert.cc:384-411)
 *   BindDict  the reference's dictionaries stay the only ones: c2v() of every string dimension, by dimension index
 *   Run       filter / having literals exactly as the JIT function receives them (db::AnyNum = 8 bytes, the column's own
 *             type in the low bytes, src/db/column.h:98-121), skip, limit; rows come back through `send` in the
 *             reference's order and formatting (post_agg.cc:26-147, sort.cc:24-75)
 *
 * Nothing here mentions a reference type: the generated text adapts (RowOutput::Send behind `send`, QueryStats fields
 * from `Stats`). C++ because rows are std::vector<std::string>, like RowOutput::Send (src/query/output.h:26-48).
 */
#ifndef VIYA_SHIM_H_
#define VIYA_SHIM_H_
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace viya {
namespace shim {

struct Session;
struct Stats { uint64_t scanned_segments, scanned_recs, aggregated_recs, output_recs; };
typedef void (*SendFn)(void* ctx, const std::vector<std::string>& row);

__attribute__((visibility("default"))) Session* Open(const void* table_key, const char* table_json, const char* query_json);
/* col_ptrs: one per storage column — dimensions, then metrics, in table order; NULL for a bitset metric (not mirrored
 * through this entry) and for the hidden count when the table has none. */
__attribute__((visibility("default"))) void Sync(Session* s, uint32_t seg, uint64_t nrows, const void* const* col_ptrs);
__attribute__((visibility("default"))) void Touch(const void* table_key, uint32_t seg, uint64_t row_first, uint64_t row_last);
__attribute__((visibility("default"))) void BindDict(Session* s, size_t dim_index, const std::vector<std::string>* c2v);
__attribute__((visibility("default"))) void Run(Session* s, const uint64_t* fargs, size_t nfargs, const uint64_t* hargs, size_t nhargs,
                                                 size_t skip, size_t limit, SendFn send, void* ctx, Stats* stats);
/* Drop everything kept for a table (mirror, sessions): Database::DropTable / process exit. */
__attribute__((visibility("default"))) void Close(const void* table_key);

}  // namespace shim
}  // namespace viya
#endif

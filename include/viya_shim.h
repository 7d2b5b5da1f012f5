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
 *   Open      per CALL: a fresh Session — this call's size() snapshot and column addresses live in it and nowhere else, so `query_threads`
 *             read-pool threads may run the function on one table at once (src/db/database.cc:28-34, src/server/http/service.cc:119) while
 *             the writer thread upserts (service.cc:103). The first Open of a (table, query text) parses the descriptors and creates the
 *             HBM mirror (the analogue of compile + cache). Release ends the call (the generated text holds it in a scope guard).
 *   Pin       per segment: the Segment OBJECT (`segment`, sizeof(Segment)). A segment never moves once created (SegmentStore keeps pointers,
 *             src/db/store.h:40-52), so it is registered with the device when first seen and every later range of it is read in place over
 *             PCIe — no host-side copy of an upsert's rows at all. Optional: without it (or when the runtime refuses) ranges are staged.
 *   Sync      per segment of table.store()->segments_copy(): this call's size() and the segment's column addresses are NOTED (no lock, no
 *             device work); Run ships what the mirror lacks — rows appended since, rows Touched since — as ONE batch (vh_table_sync_batch:
 *             one kernel launch, no per-segment synchronisation) before it plans the query
 *   BitsetStale / SyncBitset   a bitset metric's column is an array of util::Bitset<N> OBJECTS (store.cc:255-259, util/bitset.h:26-67), not
 *             of numbers: the generated text walks the rows' Roaring sets into CSR (offsets, ids) — only for segments whose rows grew or
 *             were Touched since the mirror last saw them, as many rows as BitsetStale says — and hands them over
 *   Touch     from the upsert path when metrics of EXISTING rows change in place (src/codegen/db/upsert.cc:384-411): the generated
 *             viya_upsert_do calls it next to `m.Update(...)` (viya::shim::codegen::UpsertHookText() is that line)
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
struct Stats {
  uint64_t scanned_segments, scanned_recs, aggregated_recs, output_recs;      /* query::QueryStats (src/query/stats.h:35-58) */
  /* what the device path ran on, for a maintainer's logs (the reference reports compile_time separately for the same reason): the flags of
   * vh_result_info.reserved — bit 5 a scan kernel compiled for the plan shape, bit 3 a payload projection, bit 4 narrow predicate copies,
   * bit 9 a tuple pool placed by vh_table_prepare, ... (include/viya_hip.h) —, re-plans of this query, and the kernels' time */
  uint32_t device_flags, retries;
  double scan_kernel_ms;
  double sync_ms;            /* host time Run spent bringing the mirror up to date (assembling and enqueueing the batch; 0.00x when nothing changed) */
};
typedef void (*SendFn)(void* ctx, const std::vector<std::string>& row);

__attribute__((visibility("default"))) Session* Open(const void* table_key, const char* table_json, const char* query_json);
__attribute__((visibility("default"))) void Release(Session* s);
__attribute__((visibility("default"))) void Pin(Session* s, uint32_t seg, const void* segment_object, size_t bytes);
/* col_ptrs: one per storage column — dimensions, then metrics, in table order; NULL for a bitset metric (not mirrored
 * through this entry) and for the hidden count when the table has none. */
__attribute__((visibility("default"))) void Sync(Session* s, uint32_t seg, uint64_t nrows, const void* const* col_ptrs);
/* > 0: segment `seg`'s bitset columns must be walked again before the query runs (more rows than the mirror holds, or rows Touched) — and THAT
 * many rows of them: this call's size() or, when another call brought a larger snapshot before, the rows the mirror already holds (they exist
 * on the host; a CSR mirror never shrinks under a query that planned for more). One call walks at a time: the call holds the table's walk
 * lock from here until it has handed over the segment's last bitset metric (SyncBitset) or ends. */
__attribute__((visibility("default"))) uint64_t BitsetStale(Session* s, uint32_t seg, uint64_t nrows);
/* metric_index: the bitset metric's index among the table's metrics; offsets[nrows + 1] (offsets[0] = 0), ids: uint32_t (Bitset<4>) or
 * uint64_t (Bitset<8>) values, row after row. Call for every bitset metric of a stale segment, before Run. */
__attribute__((visibility("default"))) void SyncBitset(Session* s, uint32_t seg, size_t metric_index, uint64_t nrows, const uint64_t* offsets, const void* ids);
__attribute__((visibility("default"))) void Touch(const void* table_key, uint32_t seg, uint64_t row_first, uint64_t row_last);
__attribute__((visibility("default"))) void BindDict(Session* s, size_t dim_index, const std::vector<std::string>* c2v);
__attribute__((visibility("default"))) void Run(Session* s, const uint64_t* fargs, size_t nfargs, const uint64_t* hargs, size_t nhargs,
                                                 size_t skip, size_t limit, SendFn send, void* ctx, Stats* stats);
/* Drop everything kept for a table (mirror, parsed queries, registrations): Database::DropTable / process exit. No call on the table may be in flight. */
__attribute__((visibility("default"))) void Close(const void* table_key);

/* The text a ViyaDB maintainer's generators emit at the two swap points — C++ functions returning it, like codegen::Code (src/codegen/
 * generator.h:77-97); tools/gen_shim_tu.py is the same generator in Python and the test oracle of this one (tests/test_shim_compile.py). */
namespace codegen {
/* AggQueryGenerator::GenerateCode (src/codegen/query/agg_query.cc:26-75) for the GPU path: headers, the query::AggQueryFn signature, the
 * table's Tuple / SegmentStats / Segment classes (StoreDefs, src/codegen/db/store.cc:203-356) and the calls above. */
__attribute__((visibility("default"))) std::string AggQueryText(const std::string& table_json, const std::string& query_json);
/* The line UpsertGenerator adds behind `static_cast<Segment*>(segments[segment_idx])->m.Update(upsert_tuple.m,tuple_idx);`
 * (src/codegen/db/upsert.cc:384-396): the mirror learns which row changed in place. */
__attribute__((visibility("default"))) std::string UpsertHookText();
}  // namespace codegen

}  // namespace shim
}  // namespace viya
#endif

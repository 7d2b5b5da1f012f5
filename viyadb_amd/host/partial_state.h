// partial_state.h — binary partial-state exchange for the cluster merge.
//
// What it replaces in the reference (src/cluster/query/agg_runner.cc:45-140, client.cc:64-138,
// src/server/http/output.h:31-94): the controller strips header/having/sort/skip/limit from the query
// (CreateWorkerQuery), every worker answers with its aggregated rows as TSV text, the controller re-upserts that
// text into a temporary table (count -> long_sum) and runs the original query, minus its filter, on it. The text
// hop is lossy: an AVG arrives already divided (and is then averaged again), a bitset arrives as its cardinality.
//
// Here a worker answers with a PartialState blob: typed key columns, the raw aggregation states (AVG as its sum
// next to the count it divides by, a bitset as its distinct (group, id) pairs) and the strings behind the
// dictionary codes it used (dictionaries are per process). The controller turns all blobs into the rows of one
// temporary device table and re-aggregates them on the GPU (SUM for sums and counts, MIN/MAX, set union for
// bitsets), then applies HAVING / sort / skip / limit / formatting exactly as a local query would.
//
// Wire layout (little endian, every section 8-byte aligned):
//   Header   { "VIYAPS01", u32 ndims, u32 nmetrics, u64 ngroups, u32 has_hidden, u32 reserved,
//              u64 scanned_recs, scanned_segments, passed_recs, aggregated_recs }
//   per dim  { u8 dim_type, u8 elem_size, u16 reserved, u32 name_len, u64 ndict } name, keys[ngroups],
//            ndict x { u64 code, u32 len, bytes }                     (string dimensions only)
//   per metric { u8 agg_type, u8 elem_size, u8 id_size, u8 reserved, u32 name_len, u64 npairs } name,
//            states[ngroups]                                          (all but bitsets)
//            per dim pair_keys[npairs], ids[npairs]                   (bitsets only)
//   hidden   u64[ngroups]                                             (if has_hidden)
#pragma once
#include <string>
#include <vector>

#include "viya_query.h"

namespace viya {
namespace cluster {
namespace query {

// Worker side: runs the aggregate query without header/having/sort/skip/limit (AggQueryRunner::CreateWorkerQuery)
// and returns its partial state.
std::string AggregatePartial(viya::query::AggregateQuery& query, viya::query::QueryStats& stats,
                             std::vector<db::AnyNum> fargs, int64_t now);

// Controller side: merges the workers' partial states on the GPU and finishes the query (HAVING, sort, skip,
// limit, formatting) into `output`. Strings the controller has not seen are interned into the table's dictionaries
// (the reference's temporary table shares them by dimension name: db::Dictionaries).
void MergePartials(viya::query::AggregateQuery& query, const std::vector<std::string>& partials,
                   viya::query::RowOutput& output, viya::query::QueryStats& stats);

}  // namespace query
}  // namespace cluster
}  // namespace viya

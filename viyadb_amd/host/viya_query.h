// viya_query.h — host-side mirror of the reference's query layer for aggregate (and select / search) queries.
//
//   query::Filter (one value type; FromConfig)  src/query/filter.h:38-134, src/query/filter.cc:36-108
//   query::AggregateQuery (+ SelectQuery)   src/query/query.h:152-205, src/query/query.cc:48-135
//   query::RowOutput / MemoryRowOutput      src/query/output.h:26-48
//   query::QueryStats                       src/query/stats.h:35-58
//   query::QueryRunner::Visit(AggregateQuery*)  src/query/runner.cc:45-64
//   cg::FilterArgsPacker / ValueDecoder     src/codegen/query/filter.cc:100-204
//   db::Database::{CreateTable,GetTable,Query,Load}  src/db/database.h:41-84
//
// The swap point: where the reference's QueryRunner asks AggQueryGenerator for a
// JIT-compiled `viya_query_agg` and calls it, this runner calls query::GpuAggregate — a function
// with the same argument list (table, output, stats, fargs, skip, limit, hargs) that drives the
// HIP path through the C-ABI (include/viya_hip.h) and then runs the reference's post-aggregation
// (src/codegen/query/post_agg.cc:26-147, sort.cc:24-75) on the host.
#pragma once
#include <chrono>
#include <memory>
#include <string>
#include <vector>

#include "viya_db.h"

namespace viya {
namespace query {

// One node of a predicate tree (a query's "filter" or "having"). The reference models it as a class per node kind plus a
// visitor (src/query/filter.h:38-134); here a node is a value — its kind, the column and literal strings of a leaf, or the
// parts of a conjunction / disjunction — and the four consumers (literal decoding, the device filter program, the device
// HAVING program, HAVING on the host) walk it with a switch. Built from the JSON descriptor by FromConfig.
class Filter {
public:
  enum Kind { PASS_ALL, COMPARE, MEMBER, ALL_OF, ANY_OF };
  enum Relation { EQUAL = 0, NOT_EQUAL, LESS, LESS_EQUAL, GREATER, GREATER_EQUAL };   // numbered like vh_op (include/viya_hip.h)

  Filter() = default;                                                     // PASS_ALL: "no filter"
  static Filter Compare(Relation r, std::string column, std::string literal);
  static Filter Member(std::string column, std::vector<std::string> literals, bool inside);   // IN (inside) / NOT IN
  static Filter Combine(bool all, std::vector<Filter> parts);

  // src/query/filter.cc:36-108 — "not" never survives: it flips and / or (De Morgan), negates relations and turns IN into
  // NOT IN; the parts of a composite are ordered comparisons < conjunctions < disjunctions < IN lists (stable).
  static Filter FromConfig(const util::Config& config, bool negate = false);

  Kind kind() const { return kind_; }
  Relation relation() const { return relation_; }
  bool inside() const { return inside_; }
  const std::string& column() const { return column_; }
  const std::vector<std::string>& literals() const { return literals_; }   // COMPARE: one; MEMBER: the list
  const std::vector<Filter>& parts() const { return parts_; }
  int rank() const { return kind_ == COMPARE ? 1 : kind_ == ALL_OF ? 2 : kind_ == ANY_OF ? 3 : kind_ == MEMBER ? 4 : 0; }

  // every leaf (COMPARE / MEMBER), in evaluation order — which is also the order literals are packed in
  template <class Fn> void EachLeaf(Fn&& fn) const {
    if (kind_ == COMPARE || kind_ == MEMBER) fn(*this);
    for (const Filter& part : parts_) part.EachLeaf(fn);
  }

private:
  Kind kind_ = PASS_ALL;
  Relation relation_ = EQUAL;
  bool inside_ = true;
  std::string column_;
  std::vector<std::string> literals_;
  std::vector<Filter> parts_;
};

// ---- output / stats
class RowOutput {
public:
  using Row = std::vector<std::string>;
  virtual ~RowOutput() {}
  virtual void Start() {}
  virtual void Send(const Row& row) = 0;
  virtual void SendAsCol(const Row& col) = 0;
  virtual void Flush() {}
};

class MemoryRowOutput : public RowOutput {
public:
  void Send(const Row& row) override { rows_.push_back(row); }
  void SendAsCol(const Row& row) override { rows_.push_back(row); }
  const std::vector<Row>& rows() const { return rows_; }

private:
  std::vector<Row> rows_;
};

struct QueryStats {
  size_t scanned_segments = 0, scanned_recs = 0, aggregated_recs = 0, output_recs = 0;
  double compile_time = 0, whole_time = 0;  // seconds (plan build / everything)
  // GPU-path extras (not in the reference)
  double scan_kernel_ms = 0, device_total_ms = 0;
  int path = 0;
  size_t passed_recs = 0;
  unsigned device_flags = 0, retries = 0;   // vh_result_info.reserved (compiled kernel, projection, narrow copies, placed pool, ...) and re-plans: the cliffs a maintainer should see
};

// ---- query model
class SortColumn {
public:
  SortColumn(const db::Column* col, size_t index, bool ascending) : col_(col), index_(index), ascending_(ascending) {}
  const db::Column* col() const { return col_; }
  size_t index() const { return index_; }
  bool ascending() const { return ascending_; }

private:
  const db::Column* col_;
  size_t index_;
  bool ascending_;
};

class DimOutputColumn {
public:
  DimOutputColumn(const db::Dimension* dim, size_t index) : index_(index), dim_(dim) {}
  DimOutputColumn(const util::Config& config, const db::Dimension* dim, size_t index);
  size_t index() const { return index_; }
  const db::Dimension* dim() const { return dim_; }
  const std::string& format() const { return format_; }
  util::TimeUnit granularity() const { return granularity_; }
  bool has_granularity() const { return granularity_ != util::_UNDEFINED; }

private:
  size_t index_;
  const db::Dimension* dim_;
  std::string format_;
  util::TimeUnit granularity_ = util::_UNDEFINED;
};

class MetricOutputColumn {
public:
  MetricOutputColumn(const db::Metric* metric, size_t index) : index_(index), metric_(metric) {}
  size_t index() const { return index_; }
  const db::Metric* metric() const { return metric_; }

private:
  size_t index_;
  const db::Metric* metric_;
};

class AggregateQuery {
public:
  // select_only: the SelectQuery base of the reference (src/query/query.cc:48-83) — no sort / having parsing
  AggregateQuery(const util::Config& config, db::Table& table, bool select_only = false);
  db::Table& table() { return table_; }
  bool header() const { return header_; }
  const Filter* filter() const { return filter_.get(); }
  const Filter* having() const { return having_.get(); }
  const std::vector<DimOutputColumn>& dimension_cols() const { return dimension_cols_; }
  const std::vector<MetricOutputColumn>& metric_cols() const { return metric_cols_; }
  const std::vector<SortColumn>& sort_cols() const { return sort_cols_; }
  std::vector<std::string> column_names() const;
  size_t skip() const { return skip_; }
  size_t limit() const { return limit_; }

private:
  db::Table& table_;
  bool header_;
  std::unique_ptr<Filter> filter_, having_;
  std::vector<DimOutputColumn> dimension_cols_;
  std::vector<MetricOutputColumn> metric_cols_;
  std::vector<SortColumn> sort_cols_;
  size_t skip_, limit_;
};

// query::SelectQuery (src/query/query.h:152-183): the column list, filter, skip and limit of an AggregateQuery.
class SelectQuery : public AggregateQuery {
public:
  SelectQuery(const util::Config& config, db::Table& table) : AggregateQuery(config, table, true) {}
};

// query::SearchQuery (src/query/query.h:207-224, query.cc:139-144)
class SearchQuery {
public:
  SearchQuery(const util::Config& config, db::Table& table);
  db::Table& table() { return table_; }
  bool header() const { return header_; }
  const Filter* filter() const { return filter_.get(); }
  const db::Dimension* dimension() const { return dimension_; }
  const std::string& term() const { return term_; }
  size_t limit() const { return limit_; }

private:
  db::Table& table_;
  bool header_;
  std::unique_ptr<Filter> filter_;
  const db::Dimension* dimension_;
  std::string term_;
  size_t limit_;
};

// FilterArgsPacker: literals decoded to the column's type, in traversal order.
std::vector<db::AnyNum> PackFilterArgs(const db::Table& table, const Filter* filter);

// The function that stands where the JIT-compiled viya_query_agg stood (src/query/runner.h:33-35):
// same argument meaning; `now` < 0 means std::time(nullptr) (VIYA_TEST_ROLLUP_TS in the reference).
void GpuAggregate(AggregateQuery& query, RowOutput& output, QueryStats& stats, std::vector<db::AnyNum> fargs,
                  size_t skip, size_t limit, std::vector<db::AnyNum> hargs, int64_t now, void* node_comm = nullptr);

// ... and where viya_query_select / viya_query_search stood (src/query/runner.h:29-31,37-39): same arguments.
void GpuSelect(SelectQuery& query, RowOutput& output, QueryStats& stats, std::vector<db::AnyNum> fargs, size_t skip,
               size_t limit);
void GpuSearch(SearchQuery& query, RowOutput& output, QueryStats& stats, std::vector<db::AnyNum> fargs,
               const std::string& term, size_t limit);

}  // namespace query

namespace db {

class Database {
public:
  explicit Database(const util::Config& config, int device = 0);
  ~Database();
  void CreateTable(const util::Config& table_conf);
  Table* GetTable(const std::string& name);
  query::QueryStats Query(const util::Config& query_conf, query::RowOutput& output, int64_t now = -1);
  // One node, one process per GPU, every process holding ITS segments of the tables (same descriptors and the same
  // dictionary codes everywhere): after JoinNode, aggregate queries run over all ranks' rows — the same Query() on every
  // rank, rows delivered on rank 0 (vh_query_agg_sharded; replaces the HTTP + TSV merge of src/cluster/query/agg_runner.cc:83-140
  // inside a node). `comm` is a vh_comm* (include/viya_hip.h: vh_comm_init / vh_comm_init_custom), owned by the caller.
  void JoinNode(void* comm) { comm_ = comm; }
  void* node_comm() const { return comm_; }
  void Load(const std::string& table, const std::vector<std::vector<std::string>>& rows, int64_t now = -1);
  // Cluster aggregate (src/cluster/query/agg_runner.cc:83-140) with binary partial states (partial_state.h):
  // a worker answers QueryPartial; the controller hands all answers to QueryMerge, which finishes the query.
  std::string QueryPartial(const util::Config& query_conf, query::QueryStats& stats, int64_t now = -1);
  query::QueryStats QueryMerge(const util::Config& query_conf, const std::vector<std::string>& partials, query::RowOutput& output);

private:
  Dictionaries dicts_;
  std::map<std::string, std::unique_ptr<Table>> tables_;
  void* comm_ = nullptr;
};

}  // namespace db
}  // namespace viya

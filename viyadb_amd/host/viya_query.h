// viya_query.h — host-side mirror of the reference's query layer for aggregate (and select / search) queries.
//
//   query::Filter tree + FilterFactory      src/query/filter.h:38-134, src/query/filter.cc:36-108
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

class FilterVisitor;

class Filter {
public:
  explicit Filter(int precedence) : precedence_(precedence) {}
  virtual ~Filter() {}
  int precedence() const { return precedence_; }
  virtual void Accept(FilterVisitor& v) const = 0;

private:
  const int precedence_;
};

class RelOpFilter : public Filter {
public:
  enum Operator { EQUAL = 0, NOT_EQUAL, LESS, LESS_EQUAL, GREATER, GREATER_EQUAL };
  RelOpFilter(Operator op, const std::string& column, const std::string& value) : Filter(1), op_(op), column_(column), value_(value) {}
  Operator op() const { return op_; }
  const std::string& column() const { return column_; }
  const std::string& value() const { return value_; }
  void Accept(FilterVisitor& v) const override;

private:
  Operator op_;
  std::string column_, value_;
};

class InFilter : public Filter {
public:
  InFilter(const std::string& column, const std::vector<std::string>& values, bool equal) : Filter(4), column_(column), values_(values), equal_(equal) {}
  const std::string& column() const { return column_; }
  const std::vector<std::string>& values() const { return values_; }
  bool equal() const { return equal_; }
  void Accept(FilterVisitor& v) const override;

private:
  std::string column_;
  std::vector<std::string> values_;
  bool equal_;
};

class CompositeFilter : public Filter {
public:
  enum Operator { AND, OR };
  CompositeFilter(Operator op, std::vector<std::unique_ptr<Filter>> filters) : Filter(op == AND ? 2 : 3), op_(op), filters_(std::move(filters)) {}
  Operator op() const { return op_; }
  const std::vector<std::unique_ptr<Filter>>& filters() const { return filters_; }
  void Accept(FilterVisitor& v) const override;

private:
  Operator op_;
  std::vector<std::unique_ptr<Filter>> filters_;
};

class EmptyFilter : public Filter {
public:
  EmptyFilter() : Filter(0) {}
  void Accept(FilterVisitor& v) const override;
};

class FilterVisitor {
public:
  virtual ~FilterVisitor() {}
  virtual void Visit(const RelOpFilter*) = 0;
  virtual void Visit(const InFilter*) = 0;
  virtual void Visit(const CompositeFilter*) = 0;
  virtual void Visit(const EmptyFilter*) = 0;
};

class FilterFactory {
public:
  std::unique_ptr<Filter> Create(const util::Config& config, bool negate = false);
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

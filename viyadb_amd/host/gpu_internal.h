// gpu_internal.h — pieces shared by the host shim's translation units (gpu_aggregate.cc: single-node queries;
// partial_state.cc: the cluster partial-state exchange). Not part of the mirrored reference interface.
#pragma once
#include <cstring>
#include <mutex>
#include <stdexcept>

#include "../../include/viya_hip.h"
#include "viya_query.h"

namespace viya {
namespace query {
namespace detail {

struct GpuMirror {
  vh_table* handle = nullptr;
  std::vector<uint64_t> synced_version;
  ~GpuMirror() { if (handle) vh_table_destroy(handle); }
};
inline void free_mirror(void* p) { delete static_cast<GpuMirror*>(p); }

inline void vh_check(int rc) {
  if (rc != VH_OK) throw std::runtime_error(std::string("viya_hip: ") + vh_last_error());
}

inline int dim_kind(const db::Column* d) {
  switch (d->dim_type()) {
    case db::Column::DIM_STRING: return VH_DIM_STRING;
    case db::Column::DIM_NUMERIC: return VH_DIM_NUMERIC;
    case db::Column::DIM_TIME: return VH_DIM_TIME;
    default: return VH_DIM_BOOLEAN;
  }
}
inline int metric_kind(const db::Column* m) {
  switch (m->agg_type()) {
    case db::Column::MAX: return VH_METRIC_MAX;
    case db::Column::MIN: return VH_METRIC_MIN;
    case db::Column::SUM: return VH_METRIC_SUM;
    case db::Column::AVG: return VH_METRIC_AVG;
    case db::Column::COUNT: return VH_METRIC_COUNT;
    default: return VH_METRIC_BITSET;
  }
}

void ensure_device();   // vh_init once per process (VIYA_HIP_DEVICE)
GpuMirror* ensure_mirror(db::Table& t);
// Bring the HBM mirror up to date with the host segments; returns the per-segment size() snapshot.
std::vector<uint64_t> sync_mirror(db::Table& t, GpuMirror* mir);

// query::Filter -> postfix vh_filter_node program (parts first, then the node that combines them); the literals decoded by
// PackFilterArgs are consumed in the same evaluation order. One builder serves the scan filter (columns are table columns:
// storage index) and the device HAVING (columns are RESULT columns: group column k, or ngroups + metric k).
class PlanProgram {
public:
  PlanProgram(const std::vector<db::AnyNum>& args, std::vector<vh_anynum>& lits) : args_(args), lits_(lits) {}
  template <class ColumnIndex>      // (const std::string& name) -> int32_t
  void Add(const Filter& f, ColumnIndex&& column_index) {
    switch (f.kind()) {
      case Filter::PASS_ALL: nodes.push_back({VH_F_TRUE, 0, 0, 0, 0, 0}); break;
      case Filter::COMPARE:
        nodes.push_back({VH_F_REL, column_index(f.column()), (int32_t)f.relation(), 1, (int32_t)lits_.size(), 0});
        take(1);
        break;
      case Filter::MEMBER:
        if (f.literals().empty()) throw std::runtime_error("IN filter with no values does not compile in the reference");
        nodes.push_back({VH_F_IN, column_index(f.column()), f.inside() ? 1 : 0, (int32_t)f.literals().size(), (int32_t)lits_.size(), 0});
        take(f.literals().size());
        break;
      default:
        for (const Filter& part : f.parts()) Add(part, column_index);
        nodes.push_back({f.kind() == Filter::ALL_OF ? VH_F_AND : VH_F_OR, 0, 0, (int32_t)f.parts().size(), 0, 0});
    }
  }
  std::vector<vh_filter_node> nodes;

private:
  void take(size_t n) {
    for (size_t i = 0; i < n; ++i) {
      vh_anynum a;
      a.u64 = args_.at(next_++).bits;
      lits_.push_back(a);
    }
  }
  const std::vector<db::AnyNum>& args_;
  std::vector<vh_anynum>& lits_;
  size_t next_ = 0;
};

// scan filter: a column is its storage index in the table (a bitset metric compares its per-row cardinality on the device: filter.cc:216,235)
struct PlanFilterBuilder {
  PlanFilterBuilder(const db::Table& t, const std::vector<db::AnyNum>& args) : table(t), prog(args, lits), nodes(prog.nodes) {}
  void Add(const Filter& f) { prog.Add(f, [this](const std::string& name) { return (int32_t)table.column(name)->storage_index; }); }
  const db::Table& table;
  std::vector<vh_anynum> lits;
  PlanProgram prog;
  std::vector<vh_filter_node>& nodes;
};

// HAVING on the device: a column is its position among the query's result columns
struct PlanHavingBuilder {
  PlanHavingBuilder(AggregateQuery& q, const std::vector<db::AnyNum>& args, std::vector<vh_anynum>& lits) : q_(q), prog(args, lits), nodes(prog.nodes) {}
  void Add(const Filter& f) {
    prog.Add(f, [this](const std::string& name) -> int32_t {
      const db::Column* c = q_.table().column(name);
      for (size_t k = 0; k < q_.dimension_cols().size(); ++k)
        if (q_.dimension_cols()[k].dim() == c) return (int32_t)k;
      for (size_t k = 0; k < q_.metric_cols().size(); ++k)
        if (q_.metric_cols()[k].metric() == c) return (int32_t)(q_.dimension_cols().size() + k);
      throw std::invalid_argument("Column '" + name + " is not selected");
    });
  }
  AggregateQuery& q_;
  PlanProgram prog;
  std::vector<vh_filter_node>& nodes;
};

// One aggregated group as the post-aggregation sees it.
struct Groups {
  size_t n = 0;
  std::vector<std::vector<char>> keys;    // per dimension_cols entry, n elements of the dim's type
  std::vector<std::vector<char>> states;  // per metric_cols entry, n elements of the metric's type (bitset: u64)
  std::vector<uint64_t> hidden;
};


// HAVING runs on the device when that cannot change which rows the reference would return (see GpuAggregate).
bool HavingOnDevice(AggregateQuery& query, size_t skip, size_t limit);
// sort + limit on a numeric first sort column: ask the device for the superset of groups that can make the window.
void ConfigureTopN(AggregateQuery& query, size_t skip, size_t limit, bool having_on_device, vh_plan& plan);
std::vector<vh_group_col> PlanGroupCols(AggregateQuery& query, int64_t now);
// vh_result -> typed host columns (query column order) + stats.
// extra_count_state: the plan carries one more metric than the query, a u64 SUM that plays the hidden count (cluster merge).
void FetchGroups(vh_result* res, AggregateQuery& query, Groups& groups, QueryStats& stats, bool extra_count_state = false);
void AggregateOnMirror(AggregateQuery& query, vh_table* mirror, const std::vector<uint64_t>& seg_rows, bool having_on_device,
                       const std::vector<db::AnyNum>& fargs, const std::vector<db::AnyNum>& hargs, size_t skip, size_t limit,
                       int64_t now, Groups& groups, QueryStats& stats, void* node_comm = nullptr);
// PostAggVisitor + SortVisitor (post_agg.cc:26-147, sort.cc:24-75) over fetched groups.
void PostAggregate(AggregateQuery& query, const Groups& groups, bool having_on_device, const std::vector<db::AnyNum>& hargs,
                   size_t skip, size_t limit, RowOutput& output, QueryStats& stats);

}  // namespace detail
}  // namespace query
}  // namespace viya

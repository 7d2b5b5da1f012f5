// gpu_aggregate.cc — the host shim that stands where the JIT-compiled `viya_query_agg` (and, further down,
// `viya_query_select` / `viya_query_search`) stood.
//
// Reference flow being replaced (src/codegen/query/agg_query.cc:26-71 assembles it):
//   ScanVisitor   (scan.cc:168-247)  : segment loop, predicate, agg_map[key].Update(m)   -> GPU, via the C-ABI
//   PostAggVisitor(post_agg.cc:26-147): skip/limit window, HAVING, stringification       -> here, on the host
//   SortVisitor   (sort.cc:24-75)    : string sort + skip/limit                          -> here, on the host
//
// The device mirror of the table is kept behind Table::gpu_mirror; a segment is re-synced when
// its version counter moved since the last sync (upsert appends AND updates rows in place).
#include <algorithm>
#include <cstring>
#include <ctime>
#include <set>
#include <stdexcept>

#include "gpu_internal.h"

namespace viya {
namespace query {

namespace detail {

void ensure_device() {
  static std::once_flag once;
  std::call_once(once, [] {
    const char* dev = getenv("VIYA_HIP_DEVICE");
    vh_check(vh_init(dev ? atoi(dev) : 0));
  });
}

GpuMirror* ensure_mirror(db::Table& t) {
  if (!t.gpu_mirror) {
    ensure_device();
    std::vector<vh_col_desc> cols;
    for (auto* d : t.dimensions()) cols.push_back({dim_kind(d), d->num_type().vh_elem()});
    for (auto* m : t.metrics()) {
      int elem = m->num_type().vh_elem();
      if (m->agg_type() == db::Column::BITSET) elem = m->num_type().size() == 8 ? VH_BITSET64 : VH_BITSET32;
      cols.push_back({metric_kind(m), elem});
    }
    if (t.has_hidden_count()) cols.push_back({VH_METRIC_HIDDEN_COUNT, VH_U64});
    auto* mir = new GpuMirror();
    vh_check(vh_table_create(cols.data(), (int32_t)cols.size(), t.segment_size(), 1, &mir->handle));
    t.gpu_mirror = mir;
    t.gpu_mirror_free = free_mirror;
  }
  return static_cast<GpuMirror*>(t.gpu_mirror);
}

// Bring the HBM mirror up to date with the host segments; returns the per-segment size() snapshot.
std::vector<uint64_t> sync_mirror(db::Table& t, GpuMirror* mir) {
  auto& segs = t.segments();
  std::vector<uint64_t> rows(segs.size());
  mir->synced_version.resize(segs.size(), ~0ull);
  const size_t ncols = t.storage_columns();
  for (size_t s = 0; s < segs.size(); ++s) {
    db::Segment& seg = *segs[s];
    rows[s] = seg.size();
    if (mir->synced_version[s] == seg.version) continue;
    std::vector<const void*> ptrs(ncols, nullptr);
    for (size_t c = 0; c < ncols; ++c)
      if (t.storage_elem_size(c)) ptrs[c] = seg.column(c);
    // ship only the rows upsert touched since the last sync (appends at the tail, in-place metric updates)
    if (mir->synced_version[s] == ~0ull || seg.dirty_lo >= seg.dirty_hi)
      vh_check(vh_segment_sync(mir->handle, (uint32_t)s, seg.size(), ptrs.data()));
    else
      vh_check(vh_segment_sync_range(mir->handle, (uint32_t)s, seg.dirty_lo, seg.dirty_hi - seg.dirty_lo, seg.size(), ptrs.data()));
    seg.dirty_lo = SIZE_MAX; seg.dirty_hi = 0;
    for (auto* m : t.metrics()) {
      if (m->agg_type() != db::Column::BITSET) continue;
      const auto& sets = seg.bitsets(m->index());
      std::vector<uint64_t> offsets(seg.size() + 1, 0);
      for (size_t r = 0; r < seg.size(); ++r) offsets[r + 1] = offsets[r] + sets[r].size();
      const bool wide = m->num_type().size() == 8;
      std::vector<uint64_t> v64;
      std::vector<uint32_t> v32;
      for (size_t r = 0; r < seg.size(); ++r)
        for (uint64_t id : sets[r]) { if (wide) v64.push_back(id); else v32.push_back((uint32_t)id); }
      vh_check(vh_segment_sync_bitset(mir->handle, (uint32_t)s, (int32_t)m->storage_index, seg.size(), offsets.data(),
                                      wide ? (const void*)v64.data() : (const void*)v32.data()));
    }
    mir->synced_version[s] = seg.version;
  }
  return rows;
}

}  // namespace detail

using namespace detail;

namespace {

// HAVING on aggregated tuples: same ComparisonBuilder semantics, applied to key / state values
// (AVG compares the raw sum, bitsets compare the cardinality; post_agg.cc:77-83).
class HavingEval {
public:
  HavingEval(AggregateQuery& q, const Groups& g, const std::vector<db::AnyNum>& hargs) : q_(q), g_(g), hargs_(hargs) {}
  bool Eval(const Filter* f, size_t row) {
    row_ = row; next_ = 0;
    return test(*f);
  }

private:
  // every literal is consumed whether or not the outcome is already known: they are positional (no short circuit, like the
  // reference's bitwise & / | over the generated comparisons)
  bool test(const Filter& f) {
    switch (f.kind()) {
      case Filter::PASS_ALL: return true;
      case Filter::COMPARE: return cmp(q_.table().column(f.column()), f.relation(), hargs_.at(next_++));
      case Filter::MEMBER: {
        const db::Column* c = q_.table().column(f.column());
        bool r = !f.inside();
        for (size_t i = 0; i < f.literals().size(); ++i) {
          const bool e = cmp(c, f.inside() ? Filter::EQUAL : Filter::NOT_EQUAL, hargs_.at(next_++));
          r = f.inside() ? (r | e) : (r & e);
        }
        return r;
      }
      default: {
        const bool all = f.kind() == Filter::ALL_OF;
        bool r = all;
        for (const Filter& part : f.parts()) {
          const bool e = test(part);
          r = all ? (r & e) : (r | e);
        }
        return r;
      }
    }
  }
  bool cmp(const db::Column* c, Filter::Relation op, db::AnyNum lit) {
    db::AnyNum v;
    db::Num t = c->num_type().type();
    bool found = false;
    if (c->type() == db::Column::DIMENSION) {
      for (size_t k = 0; k < q_.dimension_cols().size(); ++k)
        if (q_.dimension_cols()[k].dim() == c) { memcpy(&v.bits, &g_.keys[k][row_ * c->num_type().size()], c->num_type().size()); found = true; break; }
    } else {
      for (size_t k = 0; k < q_.metric_cols().size(); ++k)
        if (q_.metric_cols()[k].metric() == c) {
          if (c->agg_type() == db::Column::BITSET) { memcpy(&v.bits, &g_.states[k][row_ * 8], 8); t = c->num_type().size() == 8 ? db::Num::ULONG : db::Num::UINT; }
          else memcpy(&v.bits, &g_.states[k][row_ * c->num_type().size()], c->num_type().size());
          found = true;
          break;
        }
    }
    if (!found) throw std::invalid_argument("Column '" + c->name() + " is not selected");
    const int r = db::compare_typed(t, v, lit);  // 2 = unordered (NaN)
    switch (op) {
      case Filter::EQUAL: return r == 0;
      case Filter::NOT_EQUAL: return r != 0;
      case Filter::LESS: return r == -1;
      case Filter::LESS_EQUAL: return r == -1 || r == 0;
      case Filter::GREATER: return r == 1;
      default: return r == 1 || r == 0;
    }
  }
  AggregateQuery& q_;
  const Groups& g_;
  const std::vector<db::AnyNum>& hargs_;
  size_t row_ = 0, next_ = 0;
};

// util::StringNumCmp (src/util/string.h:28-49)
bool str_less(db::Column::SortType st, bool asc, const std::string& a, const std::string& b) {
  if (st == db::Column::STRING) return asc ? a < b : a > b;
  if (st == db::Column::INTEGER) {
    if (a.size() != b.size()) return asc ? a.size() < b.size() : a.size() > b.size();
    return asc ? a < b : a > b;
  }
  return asc ? std::stod(a) < std::stod(b) : std::stod(a) > std::stod(b);
}

std::string format_date(const std::string& fmt, uint32_t ts) {  // Format::date(const char*, uint32_t)
  char buf[250];
  std::tm tm;
  time_t t = (time_t)ts;
  gmtime_r(&t, &tm);
  strftime(buf, sizeof(buf), fmt.c_str(), &tm);
  return buf;
}

}  // namespace

namespace detail {

// HAVING runs on the device when that cannot change which rows the reference would return: the reference cuts
// the unsorted skip/limit window BEFORE it applies HAVING (post_agg.cc:56-83), so only push down when there is
// no such window, or when the rows are sorted first (then HAVING precedes the window in the reference too).
bool HavingOnDevice(AggregateQuery& query, size_t skip, size_t limit) {
  return query.having() != nullptr && (!query.sort_cols().empty() || (skip == 0 && limit == 0)) && !getenv("VIYA_HOST_HAVING");
}

// sort + limit on a numeric first sort column: let the device keep only the groups that can make the window
// (a superset, ties included); the string sort in PostAggregate then runs on those few rows. Columns the reference
// orders as formatted strings (string / time / boolean dims, AVG = "%.15g" text compared by length) stay on the host.
void ConfigureTopN(AggregateQuery& query, size_t skip, size_t limit, bool having_on_device, vh_plan& plan) {
  if (query.sort_cols().empty() || limit == 0 || (query.having() != nullptr && !having_on_device) || getenv("VIYA_HOST_TOPN")) return;
  const SortColumn& sc = query.sort_cols()[0];
  const db::Column* c = sc.col();
  int rc = -1;
  if (c->type() == db::Column::DIMENSION) {
    if (c->dim_type() == db::Column::DIM_NUMERIC)
      for (size_t k = 0; k < query.dimension_cols().size(); ++k)
        if (query.dimension_cols()[k].dim() == c) { rc = (int)k; break; }
  } else if (c->agg_type() != db::Column::AVG) {
    for (size_t k = 0; k < query.metric_cols().size(); ++k)
      if (query.metric_cols()[k].metric() == c) { rc = (int)(query.dimension_cols().size() + k); break; }
  }
  if (rc >= 0) { plan.top_col = rc; plan.top_desc = sc.ascending() ? 0 : 1; plan.top_k = (uint64_t)skip + limit; }
}

void FetchGroups(vh_result* res, AggregateQuery& query, Groups& groups, QueryStats& stats, bool extra_count_state) {
  vh_result_info info;
  vh_check(vh_result_get_info(res, &info));
  stats.scanned_recs += info.scanned_recs;          // scan.cc:44
  stats.scanned_segments += info.scanned_segments;  // scan.cc:51
  stats.aggregated_recs = info.ngroups;             // scan.cc:246
  stats.passed_recs = info.passed_recs;
  stats.scan_kernel_ms = info.scan_kernel_ms;
  stats.device_total_ms = info.total_ms;
  stats.path = info.path;
  stats.device_flags = info.reserved; stats.retries = info.retries;

  groups.n = info.returned_groups;
  std::vector<void*> kp, sp;
  for (auto& dc : query.dimension_cols()) { groups.keys.emplace_back(groups.n * dc.dim()->num_type().size()); kp.push_back(groups.keys.back().data()); }
  for (auto& mc : query.metric_cols()) {
    const int es = mc.metric()->agg_type() == db::Column::BITSET ? 8 : mc.metric()->num_type().size();
    groups.states.emplace_back(groups.n * es);
    sp.push_back(groups.states.back().data());
  }
  if (info.has_hidden_count || extra_count_state) groups.hidden.resize(groups.n);
  if (extra_count_state) sp.push_back(groups.hidden.data());   // the plan's last metric is a u64 SUM standing in for it
  vh_check(vh_result_copy(res, kp.data(), sp.data(), info.has_hidden_count ? groups.hidden.data() : nullptr));
}

}  // namespace detail

namespace detail {
// The body of GpuAggregate on a mirror that is already up to date: plan -> vh_query_agg -> typed groups.
void AggregateOnMirror(AggregateQuery& query, vh_table* mirror, const std::vector<uint64_t>& seg_rows, bool having_on_device,
                       const std::vector<db::AnyNum>& fargs, const std::vector<db::AnyNum>& hargs, size_t skip, size_t limit,
                       int64_t now, Groups& groups, QueryStats& stats, void* node_comm) {
  db::Table& table = query.table();
  PlanFilterBuilder fb(table, fargs);
  fb.Add(*query.filter());
  std::vector<vh_group_col> gcols = PlanGroupCols(query, now);
  std::vector<int32_t> mcols;
  for (auto& mc : query.metric_cols()) mcols.push_back((int32_t)mc.metric()->storage_index);

  PlanHavingBuilder hb(query, hargs, fb.lits);
  if (having_on_device) hb.Add(*query.having());

  vh_plan plan;
  memset(&plan, 0, sizeof(plan));
  plan.filter = fb.nodes.data(); plan.nfilter = (int32_t)fb.nodes.size();
  plan.lits = fb.lits.data(); plan.nlits = (int32_t)fb.lits.size();
  plan.having = hb.nodes.empty() ? nullptr : hb.nodes.data(); plan.nhaving = (int32_t)hb.nodes.size();
  plan.groups = gcols.data(); plan.ngroups = (int32_t)gcols.size();
  plan.metrics = mcols.data(); plan.nmetrics = (int32_t)mcols.size();
  plan.seg_rows = seg_rows.data(); plan.nseg = (uint32_t)seg_rows.size();
  const char* force = getenv("VIYA_HIP_PLAN_FLAGS");
  plan.flags = force ? (uint32_t)atoi(force) : 0;
  ConfigureTopN(query, skip, limit, having_on_device, plan);

  vh_result* res = nullptr;
  if (node_comm) vh_check(vh_query_agg_sharded(mirror, &plan, static_cast<vh_comm*>(node_comm), /*root*/0, &res));   // all ranks' rows; groups land on rank 0
  else vh_check(vh_query_agg(mirror, &plan, &res));
  std::unique_ptr<vh_result, void (*)(vh_result*)> guard(res, vh_result_free);
  FetchGroups(res, query, groups, stats);
}
}  // namespace detail

void GpuAggregate(AggregateQuery& query, RowOutput& output, QueryStats& stats, std::vector<db::AnyNum> fargs,
                  size_t skip, size_t limit, std::vector<db::AnyNum> hargs, int64_t now, void* node_comm) {
  db::Table& table = query.table();
  Groups groups;
  const bool having_on_device = HavingOnDevice(query, skip, limit);
  {
    std::lock_guard<std::mutex> lk(table.mu);
    GpuMirror* mir = ensure_mirror(table);
    std::vector<uint64_t> seg_rows = sync_mirror(table, mir);  // segments_copy() + size() snapshot
    AggregateOnMirror(query, mir->handle, seg_rows, having_on_device, fargs, hargs, skip, limit, now, groups, stats, node_comm);
  }
  PostAggregate(query, groups, having_on_device, hargs, skip, limit, output, stats);
}

namespace detail {

// The plan's GROUP BY columns: storage column + query-time rollup / granularity of time dimensions
// (RollupDefs + RollupReset + TimestampRollup, src/codegen/db/rollup.cc:25-95).
std::vector<vh_group_col> PlanGroupCols(AggregateQuery& query, int64_t now) {
  std::vector<vh_group_col> gcols(query.dimension_cols().size());
  for (size_t k = 0; k < gcols.size(); ++k) {
    const DimOutputColumn& dc = query.dimension_cols()[k];
    const db::Dimension* d = dc.dim();
    vh_group_col& g = gcols[k];
    memset(&g, 0, sizeof(g));
    g.col = (int32_t)d->storage_index;
    g.granularity = VH_T_NONE;
    if (d->dim_type() == db::Column::DIM_TIME && (!d->rollup_rules().empty() || dc.has_granularity())) {
      if (d->rollup_rules().size() > VH_MAX_ROLLUP) throw std::runtime_error("too many rollup rules");
      const auto bounds = db::rollup_boundaries(*d, now);
      g.nrollup = (int32_t)bounds.size();
      for (size_t i = 0; i < bounds.size(); ++i) { g.rollup_unit[i] = d->rollup_rules()[i].granularity; g.rollup_before[i] = bounds[i]; }
      if (dc.has_granularity()) g.granularity = dc.granularity();
    }
    g.micro = d->micro_precision() ? 1 : 0;
    if (d->dim_type() == db::Column::DIM_STRING) g.cardinality = d->dict()->c2v().size();
    else if (d->dim_type() == db::Column::DIM_BOOLEAN) g.cardinality = 2;
  }
  return gcols;
}

// ---- post aggregation (post_agg.cc:26-147)
void PostAggregate(AggregateQuery& query, const Groups& groups, bool having_on_device, const std::vector<db::AnyNum>& hargs,
                   size_t skip, size_t limit, RowOutput& output, QueryStats& stats) {
  output.Start();
  typedef std::vector<std::string> Row;
  const size_t ncols = query.dimension_cols().size() + query.metric_cols().size();
  Row row(ncols);
  const bool sorted = !query.sort_cols().empty();
  const size_t total = stats.aggregated_recs;   // agg_map.size(): HAVING may already have run on the device
  skip = std::min(total, skip);
  limit = std::min(limit, total - skip);
  size_t it = 0, end = groups.n;
  if (!sorted) {  // unsorted: the window is cut BEFORE the HAVING filter (reference behaviour)
    it = skip;
    if (limit > 0) end = it + limit;
  }
  std::vector<Row> post_agg;
  if (query.header()) {
    for (auto& dc : query.dimension_cols()) row[dc.index()] = dc.dim()->name();
    for (auto& mc : query.metric_cols()) row[mc.index()] = mc.metric()->name();
    output.Send(row);
  }
  int count_k = -1;
  for (size_t k = 0; k < query.metric_cols().size(); ++k)
    if (query.metric_cols()[k].metric()->agg_type() == db::Column::COUNT) { count_k = (int)k; break; }
  HavingEval having(query, groups, hargs);
  for (; it != end; ++it) {
    if (query.having() != nullptr && !having_on_device && !having.Eval(query.having(), it)) continue;
    for (size_t k = 0; k < query.dimension_cols().size(); ++k) {
      const DimOutputColumn& dc = query.dimension_cols()[k];
      const db::Dimension* d = dc.dim();
      const int es = d->num_type().size();
      const char* p = &groups.keys[k][it * es];
      if (d->dim_type() == db::Column::DIM_STRING) {
        uint64_t code = 0;
        memcpy(&code, p, es);
        row[dc.index()] = d->dict()->c2v().at(code);
      } else if (d->dim_type() == db::Column::DIM_TIME && !dc.format().empty()) {
        uint64_t ts = 0;
        memcpy(&ts, p, es);
        row[dc.index()] = format_date(dc.format(), (uint32_t)ts);
      } else if (d->dim_type() == db::Column::DIM_BOOLEAN) {
        row[dc.index()] = *p ? "true" : "false";
      } else {
        row[dc.index()] = db::format_num(p, d->num_type().type());
      }
    }
    for (size_t k = 0; k < query.metric_cols().size(); ++k) {
      const MetricOutputColumn& mc = query.metric_cols()[k];
      const db::Metric* m = mc.metric();
      if (m->agg_type() == db::Column::BITSET) {
        uint64_t card;
        memcpy(&card, &groups.states[k][it * 8], 8);
        row[mc.index()] = std::to_string(card);
        continue;
      }
      const int es = m->num_type().size();
      const char* p = &groups.states[k][it * es];
      if (m->agg_type() == db::Column::AVG) {  // sum / (double) count   (post_agg.cc:126-128)
        double cnt;
        if (count_k >= 0) {
          const db::Metric* cm = query.metric_cols()[count_k].metric();
          cnt = db::load_as_double(&groups.states[count_k][it * cm->num_type().size()], cm->num_type().type());
        } else {
          cnt = (double)groups.hidden[it];
        }
        const double avg = db::load_as_double(p, m->num_type().type()) / cnt;
        row[mc.index()] = db::format_num(reinterpret_cast<const char*>(&avg), db::Num::DOUBLE);
      } else {
        row[mc.index()] = db::format_num(p, m->num_type().type());
      }
    }
    if (!sorted) { output.Send(row); ++stats.output_recs; }
    else post_agg.push_back(row);
  }
  if (sorted) {  // SortVisitor (sort.cc:24-75): compares the STRINGS
    const auto& sc = query.sort_cols();
    std::stable_sort(post_agg.begin(), post_agg.end(), [&sc](const Row& a, const Row& b) {
      for (size_t i = 0; i < sc.size(); ++i) {
        const db::Column::SortType st = sc[i].col()->sort_type();
        if (str_less(st, sc[i].ascending(), a[sc[i].index()], b[sc[i].index()])) return true;
        if (i + 1 < sc.size() && str_less(st, sc[i].ascending(), b[sc[i].index()], a[sc[i].index()])) return false;
        if (i + 1 == sc.size()) return false;
      }
      return false;
    });
    // the reference computes begin()+skip+limit from the PRE-having size and can overrun; clamp
    const size_t lo = std::min(skip, post_agg.size());
    const size_t hi = limit > 0 ? std::min(skip + limit, post_agg.size()) : post_agg.size();
    for (size_t i = lo; i < hi; ++i) { output.Send(post_agg[i]); ++stats.output_recs; }
  }
  output.Flush();
}

}  // namespace detail

// ------------------------------------------------------------------------------------------------
// select: ScanVisitor::Visit(SelectQuery*) (src/codegen/query/scan.cc:75-166). The scan, the ordered
// compaction and the skip/limit window (including the reference's "break leaves the tuple loop only"
// rule) run on the GPU (vh_query_select); the rows come back as typed columns and are formatted here.
void GpuSelect(SelectQuery& query, RowOutput& output, QueryStats& stats, std::vector<db::AnyNum> fargs, size_t skip,
               size_t limit) {
  db::Table& table = query.table();
  typedef std::vector<std::string> Row;
  const size_t ncols = query.dimension_cols().size() + query.metric_cols().size();
  std::vector<int32_t> cols;
  for (auto& dc : query.dimension_cols()) cols.push_back((int32_t)dc.dim()->storage_index);
  for (auto& mc : query.metric_cols()) cols.push_back((int32_t)mc.metric()->storage_index);
  // `_j[idx] / (double) _<count field>[idx]`: a selected COUNT metric, else the hidden `_count` (scan.cc:136-152)
  int count_pos = -1;
  const db::Metric* count_metric = nullptr;
  bool has_avg = false;
  for (size_t k = 0; k < query.metric_cols().size(); ++k) {
    const db::Metric* m = query.metric_cols()[k].metric();
    has_avg |= m->agg_type() == db::Column::AVG;
    if (count_pos < 0 && m->agg_type() == db::Column::COUNT) { count_pos = (int)(query.dimension_cols().size() + k); count_metric = m; }
  }
  if (has_avg && count_pos < 0) {
    if (!table.has_hidden_count()) throw std::runtime_error("AVG in a select needs a COUNT metric or the hidden count (the reference's generated code would not compile)");
    count_pos = (int)cols.size();
    cols.push_back((int32_t)table.hidden_count_storage_index());
  }

  std::vector<std::vector<char>> data(cols.size());
  std::vector<int> esize(cols.size());
  uint64_t nrows = 0;
  {
    std::lock_guard<std::mutex> lk(table.mu);
    GpuMirror* mir = ensure_mirror(table);
    std::vector<uint64_t> seg_rows = sync_mirror(table, mir);
    PlanFilterBuilder fb(table, fargs);
    fb.Add(*query.filter());
    vh_select_plan plan;
    memset(&plan, 0, sizeof(plan));
    plan.filter = fb.nodes.data(); plan.nfilter = (int32_t)fb.nodes.size();
    plan.lits = fb.lits.data(); plan.nlits = (int32_t)fb.lits.size();
    plan.cols = cols.data(); plan.ncols = (int32_t)cols.size();
    plan.seg_rows = seg_rows.data(); plan.nseg = (uint32_t)seg_rows.size();
    plan.skip = skip; plan.limit = limit;
    vh_rows* rows = nullptr;
    vh_check(vh_query_select(mir->handle, &plan, &rows));
    std::unique_ptr<vh_rows, void (*)(vh_rows*)> guard(rows, vh_rows_free);
    vh_rows_info info;
    vh_check(vh_rows_get_info(rows, &info));
    stats.scanned_recs += info.scanned_recs;
    stats.scanned_segments += info.scanned_segments;
    stats.passed_recs = info.passed_recs;
    stats.scan_kernel_ms = info.kernel_ms;
    stats.device_total_ms = info.total_ms;
    nrows = info.nrows;
    std::vector<const void*> ptrs(cols.size() + 1, nullptr);
    vh_check(vh_rows_view(rows, ptrs.data()));
    for (size_t c = 0; c < cols.size(); ++c) {
      int es = table.storage_elem_size(cols[c]);
      if (es == 0) es = 8;   // bitset column: per-row cardinality as u64
      esize[c] = es;
      data[c].resize(nrows * es);
      if (nrows) memcpy(data[c].data(), ptrs[c], nrows * es);
    }
  }

  output.Start();
  Row row(ncols);
  if (query.header()) {
    for (auto& dc : query.dimension_cols()) row[dc.index()] = dc.dim()->name();
    for (auto& mc : query.metric_cols()) row[mc.index()] = mc.metric()->name();
    output.Send(row);
  }
  for (uint64_t i = 0; i < nrows; ++i) {
    size_t c = 0;
    for (auto& dc : query.dimension_cols()) {
      const db::Dimension* d = dc.dim();
      const char* p = &data[c][i * esize[c]];
      if (d->dim_type() == db::Column::DIM_STRING) {
        uint64_t code = 0;
        memcpy(&code, p, esize[c]);
        row[dc.index()] = d->dict()->c2v().at(code);
      } else if (d->dim_type() == db::Column::DIM_TIME && !dc.format().empty()) {
        uint64_t ts = 0;
        memcpy(&ts, p, esize[c]);
        row[dc.index()] = format_date(dc.format(), (uint32_t)ts);
      } else if (d->dim_type() == db::Column::DIM_BOOLEAN) {
        row[dc.index()] = *p ? "true" : "false";
      } else {
        row[dc.index()] = db::format_num(p, d->num_type().type());
      }
      ++c;
    }
    for (auto& mc : query.metric_cols()) {
      const db::Metric* m = mc.metric();
      const char* p = &data[c][i * esize[c]];
      if (m->agg_type() == db::Column::BITSET) {
        uint64_t card;
        memcpy(&card, p, 8);
        row[mc.index()] = std::to_string(card);
      } else if (m->agg_type() == db::Column::AVG) {
        const char* cp = &data[count_pos][i * esize[count_pos]];
        const double cnt = count_metric ? db::load_as_double(cp, count_metric->num_type().type()) : (double)*reinterpret_cast<const uint64_t*>(cp);
        const double avg = db::load_as_double(p, m->num_type().type()) / cnt;
        row[mc.index()] = db::format_num(reinterpret_cast<const char*>(&avg), db::Num::DOUBLE);
      } else {
        row[mc.index()] = db::format_num(p, m->num_type().type());
      }
      ++c;
    }
    output.Send(row);
    ++stats.output_recs;
  }
  output.Flush();
}

// ------------------------------------------------------------------------------------------------
// search: ScanVisitor::Visit(SearchQuery*) + PostAggVisitor::Visit(SearchQuery*) (scan.cc:249-299,
// post_agg.cc:149-166). `codes.insert(value).second` in storage order == GROUP BY the dimension with MIN over the
// row's storage position (VH_COL_ROWID), groups taken in that order. The term match runs on the decoded values
// here. Limit: the reference's `break` leaves the tuple loop only — the rest of THAT segment is not scanned (codes
// first seen there are not inserted) and every later segment is scanned up to its first new matching value; that
// tail is reproduced with one single-segment query per later segment.
void GpuSearch(SearchQuery& query, RowOutput& output, QueryStats& stats, std::vector<db::AnyNum> fargs,
               const std::string& term, size_t limit) {
  db::Table& table = query.table();
  const db::Dimension* dim = query.dimension();
  const int es = dim->num_type().size();
  std::vector<std::string> values;
  std::set<uint64_t> codes;   // value bits (the reference's unordered_set<T>)
  auto decode = [&](uint64_t bits) -> std::string {
    if (dim->dim_type() == db::Column::DIM_STRING) return dim->dict()->c2v().at(bits);
    if (dim->dim_type() == db::Column::DIM_BOOLEAN) return (bits & 0xff) ? "true" : "false";
    return db::format_num(reinterpret_cast<const char*>(&bits), dim->num_type().type());
  };
  {
    std::lock_guard<std::mutex> lk(table.mu);
    GpuMirror* mir = ensure_mirror(table);
    std::vector<uint64_t> seg_rows = sync_mirror(table, mir);
    PlanFilterBuilder fb(table, fargs);
    fb.Add(*query.filter());
    vh_group_col g;
    memset(&g, 0, sizeof(g));
    g.col = (int32_t)dim->storage_index;
    g.granularity = VH_T_NONE;
    if (dim->dim_type() == db::Column::DIM_STRING) g.cardinality = dim->dict()->c2v().size();
    else if (dim->dim_type() == db::Column::DIM_BOOLEAN) g.cardinality = 2;
    int32_t rowid = VH_COL_ROWID;

    // -> (first storage position, value bits) of every distinct value among the passing rows, in storage order
    auto first_occurrences = [&](const std::vector<uint64_t>& rows, bool count_stats) {
      vh_plan plan;
      memset(&plan, 0, sizeof(plan));
      plan.filter = fb.nodes.data(); plan.nfilter = (int32_t)fb.nodes.size();
      plan.lits = fb.lits.data(); plan.nlits = (int32_t)fb.lits.size();
      plan.groups = &g; plan.ngroups = 1;
      plan.metrics = &rowid; plan.nmetrics = 1;
      plan.seg_rows = rows.data(); plan.nseg = (uint32_t)rows.size();
      vh_result* res = nullptr;
      vh_check(vh_query_agg(mir->handle, &plan, &res));
      std::unique_ptr<vh_result, void (*)(vh_result*)> guard(res, vh_result_free);
      vh_result_info info;
      vh_check(vh_result_get_info(res, &info));
      if (count_stats) {
        stats.scanned_recs += info.scanned_recs;
        stats.scanned_segments += info.scanned_segments;
        stats.passed_recs = info.passed_recs;
        stats.scan_kernel_ms = info.scan_kernel_ms;
        stats.device_total_ms = info.total_ms;
        stats.path = info.path;
        stats.device_flags = info.reserved; stats.retries = info.retries;
      }
      std::vector<char> keys(info.returned_groups * es);
      std::vector<uint64_t> pos(info.returned_groups);
      void* kp[1] = {keys.data()};
      void* sp[1] = {pos.data()};
      vh_check(vh_result_copy(res, kp, sp, nullptr));
      std::vector<std::pair<uint64_t, uint64_t>> out(info.returned_groups);
      for (size_t i = 0; i < out.size(); ++i) {
        uint64_t bits = 0;
        memcpy(&bits, &keys[i * es], es);
        out[i] = {pos[i], bits};
      }
      std::sort(out.begin(), out.end());
      return out;
    };

    bool limit_hit = false;
    uint64_t hit_segment = 0;
    for (auto& pv : first_occurrences(seg_rows, true)) {
      codes.insert(pv.second);
      const std::string check = decode(pv.second);
      if (check.find(term) != std::string::npos) {
        values.push_back(check);
        if (limit > 0 && values.size() >= limit) { limit_hit = true; hit_segment = pv.first >> 32; break; }
      }
    }
    if (limit_hit) {
      for (uint64_t s = hit_segment + 1; s < seg_rows.size(); ++s) {
        if (!seg_rows[s]) continue;
        std::vector<uint64_t> only(seg_rows.size(), 0);
        only[s] = seg_rows[s];
        for (auto& pv : first_occurrences(only, false)) {
          if (!codes.insert(pv.second).second) continue;
          const std::string check = decode(pv.second);
          if (check.find(term) != std::string::npos) { values.push_back(check); break; }   // size >= limit holds
        }
      }
    }
  }
  stats.aggregated_recs = codes.size();   // scan.cc:298
  output.Start();
  if (query.header()) output.Send(std::vector<std::string>{dim->name()});
  output.SendAsCol(values);
  stats.output_recs = values.size();
  output.Flush();
}

}  // namespace query
}  // namespace viya

// partial_state.cc — worker-side export and controller-side GPU merge of aggregate partial states
// (see partial_state.h for what this replaces in the reference and for the wire layout).
#include "partial_state.h"

#include <algorithm>
#include <cstring>
#include <unordered_map>

#include "gpu_internal.h"

namespace viya {
namespace cluster {
namespace query {

using namespace viya::query;
using namespace viya::query::detail;

namespace {

const char kMagic[8] = {'V', 'I', 'Y', 'A', 'P', 'S', '0', '1'};

struct WireHeader {
  char magic[8];
  uint32_t ndims, nmetrics;
  uint64_t ngroups;
  uint32_t has_hidden, reserved;
  uint64_t scanned_recs, scanned_segments, passed_recs, aggregated_recs;
};
struct WireDim {
  uint8_t dim_type, elem_size;
  uint16_t reserved;
  uint32_t name_len;
  uint64_t ndict;
};
struct WireMetric {
  uint8_t agg_type, elem_size, id_size, reserved;
  uint32_t name_len;
  uint64_t npairs;
};

class Writer {
public:
  template <typename T> void put(const T& v) { bytes(&v, sizeof(T)); }
  void bytes(const void* p, size_t n) { buf_.append(static_cast<const char*>(p), n); }
  void align8() { buf_.append((8 - buf_.size() % 8) % 8, '\0'); }
  std::string take() { return std::move(buf_); }

private:
  std::string buf_;
};

class Reader {
public:
  Reader(const std::string& s) : p_(s.data()), n_(s.size()) {}
  template <typename T> T get() { T v; memcpy(&v, take(sizeof(T)), sizeof(T)); return v; }
  const char* take(size_t n) {
    if (n > n_ - pos_) throw std::runtime_error("partial state is truncated");
    const char* r = p_ + pos_;
    pos_ += n;
    return r;
  }
  void align8() { take((8 - pos_ % 8) % 8); }
  bool done() const { return pos_ == n_; }

private:
  const char* p_;
  size_t n_, pos_ = 0;
};

size_t checked_mul(uint64_t n, size_t es) {
  if (es && n > (uint64_t)1 << 40) throw std::runtime_error("partial state declares an absurd row count");
  return (size_t)n * es;
}

struct BitsetPairs {
  uint64_t n = 0;
  int id_size = 4;
  std::vector<std::vector<char>> keys;  // per query dimension, n elements
  std::vector<char> ids;
};

int state_size(const db::Metric* m) { return m->agg_type() == db::Column::BITSET ? 0 : m->num_type().size(); }

// DimensionDict encode as upsert does it (src/codegen/db/upsert.cc:43-80): next free code, "__exceeded" (0) past the
// dimension's cardinality.
uint64_t intern(const db::Dimension* d, const std::string& value) {
  db::DimensionDict* dict = d->dict();
  auto it = dict->v2c().find(value);
  if (it != dict->v2c().end()) return it->second;
  const uint64_t code = dict->c2v().size();
  if (d->cardinality() < UINT64_MAX - 1 && code > d->cardinality()) return 0;
  dict->v2c().emplace(value, code);
  dict->c2v().emplace_back(value);
  return code;
}

uint64_t load_code(const char* p, int es) {
  uint64_t v = 0;
  memcpy(&v, p, es);
  return v;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ worker
std::string AggregatePartial(AggregateQuery& query, QueryStats& stats, std::vector<db::AnyNum> fargs, int64_t now) {
  db::Table& table = query.table();
  const size_t nd = query.dimension_cols().size(), nm = query.metric_cols().size();
  Groups groups;
  std::vector<BitsetPairs> pairs(nm);
  {
    std::lock_guard<std::mutex> lk(table.mu);
    GpuMirror* mir = ensure_mirror(table);
    std::vector<uint64_t> seg_rows = sync_mirror(table, mir);
    PlanFilterBuilder fb(table, fargs);
    fb.Add(*query.filter());
    std::vector<vh_group_col> gcols = PlanGroupCols(query, now);
    std::vector<int32_t> mcols;
    bool has_bitset = false;
    for (auto& mc : query.metric_cols()) {
      mcols.push_back((int32_t)mc.metric()->storage_index);
      has_bitset |= mc.metric()->agg_type() == db::Column::BITSET;
    }
    vh_plan plan;
    memset(&plan, 0, sizeof(plan));
    plan.filter = fb.nodes.data(); plan.nfilter = (int32_t)fb.nodes.size();
    plan.lits = fb.lits.data(); plan.nlits = (int32_t)fb.lits.size();
    plan.groups = gcols.data(); plan.ngroups = (int32_t)gcols.size();
    plan.metrics = mcols.data(); plan.nmetrics = (int32_t)mcols.size();
    plan.seg_rows = seg_rows.data(); plan.nseg = (uint32_t)seg_rows.size();
    const char* force = getenv("VIYA_HIP_PLAN_FLAGS");
    plan.flags = force ? (uint32_t)atoi(force) : 0;
    // the (group, id) pairs are read out of ONE set table: no per-XCD private copies of a dense table
    if (has_bitset) plan.flags |= VH_PLAN_NO_XCD_PRIVATE | VH_PLAN_NO_HPART;      // (the blob carries the distinct (group, id) pairs: the set table must exist)

    vh_result* res = nullptr;
    vh_check(vh_query_agg(mir->handle, &plan, &res));
    std::unique_ptr<vh_result, void (*)(vh_result*)> guard(res, vh_result_free);
    FetchGroups(res, query, groups, stats);
    for (size_t k = 0; k < nm; ++k) {
      if (query.metric_cols()[k].metric()->agg_type() != db::Column::BITSET) continue;
      uint64_t offs[2] = {0, 0};
      std::vector<vh_device_buffer> bufs(nd + 1);
      int32_t nb = 0;
      vh_check(vh_result_partition_pairs(res, (int32_t)k, 1, offs, bufs.data(), (int32_t)bufs.size(), &nb));
      if ((size_t)nb != nd + 1) throw std::runtime_error("pair export returned an unexpected column count");
      BitsetPairs& bp = pairs[k];
      bp.n = bufs[nd].count;
      bp.id_size = bufs[nd].elem == VH_U64 ? 8 : 4;
      for (size_t c = 0; c < nd; ++c) {
        const int es = query.dimension_cols()[c].dim()->num_type().size();
        bp.keys.emplace_back(bp.n * es);
        vh_check(vh_device_read(bp.keys.back().data(), bufs[c].ptr, bp.n * es));
      }
      bp.ids.resize(bp.n * bp.id_size);
      vh_check(vh_device_read(bp.ids.data(), bufs[nd].ptr, bp.ids.size()));
    }
  }

  Writer w;
  WireHeader h;
  memset(&h, 0, sizeof(h));
  memcpy(h.magic, kMagic, 8);
  h.ndims = (uint32_t)nd; h.nmetrics = (uint32_t)nm; h.ngroups = groups.n;
  h.has_hidden = groups.hidden.empty() ? 0 : 1;
  h.scanned_recs = stats.scanned_recs; h.scanned_segments = stats.scanned_segments;
  h.passed_recs = stats.passed_recs; h.aggregated_recs = stats.aggregated_recs;
  w.put(h);
  for (size_t k = 0; k < nd; ++k) {
    const db::Dimension* d = query.dimension_cols()[k].dim();
    const int es = d->num_type().size();
    // the strings behind the codes this partial uses (group keys; pair keys are a subset of them)
    std::vector<uint64_t> used;
    if (d->dim_type() == db::Column::DIM_STRING) {
      std::vector<bool> seen(d->dict()->c2v().size(), false);
      for (uint64_t i = 0; i < groups.n; ++i) {
        const uint64_t code = load_code(&groups.keys[k][i * es], es);
        if (code < seen.size() && !seen[code]) { seen[code] = true; used.push_back(code); }
      }
    }
    WireDim wd;
    memset(&wd, 0, sizeof(wd));
    wd.dim_type = (uint8_t)d->dim_type(); wd.elem_size = (uint8_t)es; wd.name_len = (uint32_t)d->name().size(); wd.ndict = used.size();
    w.put(wd);
    w.bytes(d->name().data(), d->name().size()); w.align8();
    w.bytes(groups.keys[k].data(), groups.keys[k].size()); w.align8();
    for (uint64_t code : used) {
      const std::string& v = d->dict()->c2v()[code];
      w.put(code); w.put((uint32_t)v.size()); w.bytes(v.data(), v.size()); w.align8();
    }
  }
  for (size_t k = 0; k < nm; ++k) {
    const db::Metric* m = query.metric_cols()[k].metric();
    WireMetric wm;
    memset(&wm, 0, sizeof(wm));
    wm.agg_type = (uint8_t)m->agg_type(); wm.elem_size = (uint8_t)state_size(m); wm.name_len = (uint32_t)m->name().size();
    wm.id_size = (uint8_t)pairs[k].id_size; wm.npairs = pairs[k].n;
    w.put(wm);
    w.bytes(m->name().data(), m->name().size()); w.align8();
    if (m->agg_type() != db::Column::BITSET) {
      w.bytes(groups.states[k].data(), groups.states[k].size()); w.align8();
    } else {
      for (auto& kc : pairs[k].keys) { w.bytes(kc.data(), kc.size()); w.align8(); }
      w.bytes(pairs[k].ids.data(), pairs[k].ids.size()); w.align8();
    }
  }
  if (h.has_hidden) w.bytes(groups.hidden.data(), groups.hidden.size() * 8);
  return w.take();
}

// -------------------------------------------------------------------------------------------- controller
namespace {

// One temporary-table column under construction.
struct TempCol {
  int es = 0;
  std::vector<char> data;
  void append(const char* p, size_t bytes) { data.insert(data.end(), p, p + bytes); }
  void fill(db::AnyNum v, uint64_t n) {
    const size_t at = data.size();
    data.resize(at + n * es);
    for (uint64_t i = 0; i < n; ++i) memcpy(&data[at + i * es], &v.bits, es);
  }
};

}  // namespace

void MergePartials(AggregateQuery& query, const std::vector<std::string>& partials, RowOutput& output, QueryStats& stats) {
  db::Table& table = query.table();
  const size_t nd = query.dimension_cols().size(), nm = query.metric_cols().size();
  bool need_hidden = false;   // AVG without a selected COUNT divides by the hidden count
  {
    bool has_avg = false, has_count = false;
    for (auto& mc : query.metric_cols()) {
      has_avg |= mc.metric()->agg_type() == db::Column::AVG;
      has_count |= mc.metric()->agg_type() == db::Column::COUNT;
    }
    need_hidden = has_avg && !has_count;
  }
  std::vector<TempCol> dcols(nd), mcols(nm);
  TempCol hidden;
  hidden.es = 8;
  for (size_t k = 0; k < nd; ++k) dcols[k].es = query.dimension_cols()[k].dim()->num_type().size();
  for (size_t k = 0; k < nm; ++k) mcols[k].es = state_size(query.metric_cols()[k].metric());
  // bitset columns as CSR over ALL temporary rows: group rows carry an empty set, pair rows one id
  std::vector<std::vector<uint64_t>> bs_offsets(nm);
  std::vector<std::vector<char>> bs_values(nm);
  std::vector<int> bs_id_size(nm, 0);
  for (size_t k = 0; k < nm; ++k) {
    const db::Metric* m = query.metric_cols()[k].metric();
    if (m->agg_type() == db::Column::BITSET) { bs_offsets[k].push_back(0); bs_id_size[k] = m->num_type().size() == 8 ? 8 : 4; }
  }
  uint64_t nrows = 0;
  QueryStats workers;
  std::vector<db::AnyNum> hargs;
  const bool having_on_device = HavingOnDevice(query, query.skip(), query.limit());
  Groups groups;

  std::lock_guard<std::mutex> lk(table.mu);
  auto pad_bitsets = [&](uint64_t n, size_t except) {
    for (size_t k = 0; k < nm; ++k)
      if (!bs_offsets[k].empty() && k != except) bs_offsets[k].insert(bs_offsets[k].end(), n, bs_offsets[k].back());
  };
  // rows whose only purpose is to carry a bitset id: every other state gets the value that leaves it unchanged
  auto neutral_states = [&](uint64_t n) {
    for (size_t k = 0; k < nm; ++k) {
      const db::Metric* m = query.metric_cols()[k].metric();
      if (m->agg_type() == db::Column::BITSET) continue;
      db::AnyNum v;   // SUM / COUNT / AVG sums: 0
      if (m->agg_type() == db::Column::MIN) v = m->num_type().cpp_max_value();
      else if (m->agg_type() == db::Column::MAX) v = m->num_type().cpp_min_value();
      mcols[k].fill(v, n);
    }
    if (need_hidden) hidden.fill(db::AnyNum(), n);
  };

  for (const std::string& blob : partials) {
    Reader r(blob);
    const WireHeader h = r.get<WireHeader>();
    if (memcmp(h.magic, kMagic, 8) != 0) throw std::runtime_error("not a partial state (bad magic)");
    if (h.ndims != nd || h.nmetrics != nm) throw std::runtime_error("partial state does not match the query's column list");
    if (need_hidden && !h.has_hidden && h.ngroups) throw std::runtime_error("partial state lacks the hidden count an AVG needs");
    workers.scanned_recs += h.scanned_recs; workers.scanned_segments += h.scanned_segments; workers.passed_recs += h.passed_recs;
    std::vector<std::unordered_map<uint64_t, uint64_t>> remap(nd);
    auto append_keys = [&](size_t k, const char* src, uint64_t n) {
      const db::Dimension* d = query.dimension_cols()[k].dim();
      const int es = dcols[k].es;
      if (d->dim_type() != db::Column::DIM_STRING) { dcols[k].append(src, n * es); return; }
      const size_t at = dcols[k].data.size();
      dcols[k].data.resize(at + n * es);
      for (uint64_t i = 0; i < n; ++i) {
        auto it = remap[k].find(load_code(src + i * es, es));
        if (it == remap[k].end()) throw std::runtime_error("partial state uses a dictionary code it does not define");
        memcpy(&dcols[k].data[at + i * es], &it->second, es);
      }
    };
    for (size_t k = 0; k < nd; ++k) {
      const db::Dimension* d = query.dimension_cols()[k].dim();
      const WireDim wd = r.get<WireDim>();
      const std::string name(r.take(wd.name_len), wd.name_len);
      r.align8();
      if (name != d->name() || wd.elem_size != dcols[k].es || wd.dim_type != (uint8_t)d->dim_type())
        throw std::runtime_error("partial state column '" + name + "' does not match dimension '" + d->name() + "'");
      const char* keys = r.take(checked_mul(h.ngroups, wd.elem_size));
      r.align8();
      for (uint64_t i = 0; i < wd.ndict; ++i) {
        const uint64_t code = r.get<uint64_t>();
        const uint32_t len = r.get<uint32_t>();
        const std::string v(r.take(len), len);
        r.align8();
        remap[k][code] = intern(d, v);
      }
      append_keys(k, keys, h.ngroups);
    }
    pad_bitsets(h.ngroups, SIZE_MAX);
    uint64_t blob_rows = h.ngroups;
    for (size_t k = 0; k < nm; ++k) {
      const db::Metric* m = query.metric_cols()[k].metric();
      const WireMetric wm = r.get<WireMetric>();
      const std::string name(r.take(wm.name_len), wm.name_len);
      r.align8();
      if (name != m->name() || wm.agg_type != (uint8_t)m->agg_type() || wm.elem_size != state_size(m))
        throw std::runtime_error("partial state column '" + name + "' does not match metric '" + m->name() + "'");
      if (m->agg_type() != db::Column::BITSET) {
        mcols[k].append(r.take(checked_mul(h.ngroups, wm.elem_size)), h.ngroups * wm.elem_size);
        r.align8();
        continue;
      }
      if (wm.id_size != bs_id_size[k]) throw std::runtime_error("partial state bitset '" + name + "' has the wrong id width");
      // the pairs become rows AFTER this blob's group rows and after earlier bitset metrics' pair rows; the other
      // metrics of this blob are appended below, so remember the pair sections and emit their rows at the end
      for (size_t c = 0; c < nd; ++c) {
        append_keys(c, r.take(checked_mul(wm.npairs, dcols[c].es)), wm.npairs);
        r.align8();
      }
      const char* ids = r.take(checked_mul(wm.npairs, wm.id_size));
      r.align8();
      bs_values[k].insert(bs_values[k].end(), ids, ids + wm.npairs * wm.id_size);
      const uint64_t base = bs_offsets[k].back();
      for (uint64_t i = 1; i <= wm.npairs; ++i) bs_offsets[k].push_back(base + i);
      pad_bitsets(wm.npairs, k);
      blob_rows += wm.npairs;
    }
    if (h.has_hidden) {
      const char* hp = r.take(checked_mul(h.ngroups, 8));
      if (need_hidden) hidden.append(hp, h.ngroups * 8);
    } else if (need_hidden) {
      hidden.fill(db::AnyNum(), h.ngroups);
    }
    if (!r.done()) throw std::runtime_error("partial state has trailing bytes");
    // non-bitset states of the pair rows (they sit behind the group rows of this blob in every column)
    neutral_states(blob_rows - h.ngroups);
    nrows += blob_rows;
  }
  // Column order inside one blob's rows: [group rows][pairs of bitset 0][pairs of bitset 1]... for the key columns and
  // the CSR offsets; the plain states were appended as [group rows] then [neutral x all pair rows]: same positions.

  hargs = PackFilterArgs(table, query.having());   // after interning: literals may name strings only workers had

  if (nrows) {
    ensure_device();
    std::vector<vh_col_desc> cols;
    for (size_t k = 0; k < nd; ++k) {
      const db::Dimension* d = query.dimension_cols()[k].dim();
      cols.push_back({dim_kind(d), d->num_type().vh_elem()});
    }
    for (size_t k = 0; k < nm; ++k) {
      const db::Metric* m = query.metric_cols()[k].metric();
      int kind = VH_METRIC_SUM, elem = m->num_type().vh_elem();   // sums, counts ("count" -> "long_sum" in the reference) and AVG sums add up
      if (m->agg_type() == db::Column::MIN) kind = VH_METRIC_MIN;
      else if (m->agg_type() == db::Column::MAX) kind = VH_METRIC_MAX;
      else if (m->agg_type() == db::Column::BITSET) { kind = VH_METRIC_BITSET; elem = bs_id_size[k] == 8 ? VH_BITSET64 : VH_BITSET32; }
      cols.push_back({kind, elem});
    }
    if (need_hidden) cols.push_back({VH_METRIC_SUM, VH_U64});
    vh_table* tmp = nullptr;
    vh_check(vh_table_create(cols.data(), (int32_t)cols.size(), nrows, 1, &tmp));
    std::unique_ptr<vh_table, void (*)(vh_table*)> tguard(tmp, vh_table_destroy);
    std::vector<const void*> ptrs;
    for (auto& c : dcols) ptrs.push_back(c.data.data());
    for (size_t k = 0; k < nm; ++k) ptrs.push_back(mcols[k].es ? mcols[k].data.data() : nullptr);
    if (need_hidden) ptrs.push_back(hidden.data.data());
    for (size_t k = 0; k < nd; ++k)
      if (dcols[k].data.size() != nrows * dcols[k].es) throw std::runtime_error("internal: key column length mismatch");
    for (size_t k = 0; k < nm; ++k)
      if (mcols[k].es && mcols[k].data.size() != nrows * mcols[k].es) throw std::runtime_error("internal: state column length mismatch");
    vh_check(vh_segment_sync(tmp, 0, nrows, ptrs.data()));
    for (size_t k = 0; k < nm; ++k) {
      if (bs_offsets[k].empty()) continue;
      if (bs_offsets[k].size() != nrows + 1) throw std::runtime_error("internal: bitset offsets length mismatch");
      vh_check(vh_segment_sync_bitset(tmp, 0, (int32_t)(nd + k), nrows, bs_offsets[k].data(), bs_values[k].data()));
    }

    std::vector<vh_group_col> gcols(nd);
    for (size_t k = 0; k < nd; ++k) {
      const db::Dimension* d = query.dimension_cols()[k].dim();
      memset(&gcols[k], 0, sizeof(vh_group_col));
      gcols[k].col = (int32_t)k;
      gcols[k].granularity = VH_T_NONE;   // keys arrive truncated / rolled up by the workers
      if (d->dim_type() == db::Column::DIM_STRING) gcols[k].cardinality = d->dict()->c2v().size();
      else if (d->dim_type() == db::Column::DIM_BOOLEAN) gcols[k].cardinality = 2;
    }
    std::vector<int32_t> pm;
    for (size_t k = 0; k < nm; ++k) pm.push_back((int32_t)(nd + k));
    if (need_hidden) pm.push_back((int32_t)(nd + nm));
    std::vector<vh_anynum> lits;
    PlanHavingBuilder hb(query, hargs, lits);
    if (having_on_device) hb.Add(*query.having());
    vh_filter_node all = {VH_F_TRUE, 0, 0, 0, 0, 0};
    vh_plan plan;
    memset(&plan, 0, sizeof(plan));
    plan.filter = &all; plan.nfilter = 1;
    plan.lits = lits.data(); plan.nlits = (int32_t)lits.size();
    plan.having = hb.nodes.empty() ? nullptr : hb.nodes.data(); plan.nhaving = (int32_t)hb.nodes.size();
    plan.groups = gcols.data(); plan.ngroups = (int32_t)nd;
    plan.metrics = pm.data(); plan.nmetrics = (int32_t)pm.size();
    plan.seg_rows = &nrows; plan.nseg = 1;
    const char* force = getenv("VIYA_HIP_PLAN_FLAGS");
    plan.flags = force ? (uint32_t)atoi(force) : 0;
    ConfigureTopN(query, query.skip(), query.limit(), having_on_device, plan);
    vh_result* res = nullptr;
    vh_check(vh_query_agg(tmp, &plan, &res));
    std::unique_ptr<vh_result, void (*)(vh_result*)> guard(res, vh_result_free);
    QueryStats merged;
    FetchGroups(res, query, groups, merged, need_hidden);
    stats.aggregated_recs = merged.aggregated_recs;
    stats.scan_kernel_ms = merged.scan_kernel_ms;
    stats.device_total_ms = merged.device_total_ms;
    stats.path = merged.path;
  } else {
    stats.aggregated_recs = 0;
  }
  stats.scanned_recs += workers.scanned_recs;
  stats.scanned_segments += workers.scanned_segments;
  stats.passed_recs = workers.passed_recs;
  PostAggregate(query, groups, having_on_device, hargs, query.skip(), query.limit(), output, stats);
}

}  // namespace query
}  // namespace cluster
}  // namespace viya

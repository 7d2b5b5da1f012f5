// shim_session.cc — include/viya_shim.h: the host side of a GENERATED viya_query_agg (tools/gen_shim_tu.py).
//
// The generated function lives inside the reference process and owns nothing but pointers into the reference's segments
// and dictionaries. Everything the GPU path needs beyond them is kept here, keyed by the address of the reference's
// db::Table: a descriptor-only shadow of the table (column types, rollup rules — parsed from the same JSON the reference
// built its table from), the HBM mirror, and one parsed query per query text (the reference caches one compiled
// function per generated source, src/codegen/compiler.cc:97-144).
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>

#include "../../include/viya_shim.h"
#include "gpu_internal.h"

namespace viya {
namespace shim {

namespace q = viya::query;
using q::detail::GpuMirror;

namespace {
struct Shadow {
  db::Dictionaries dicts;
  std::unique_ptr<db::Table> table;          // descriptors only: it never holds a row
  std::unique_ptr<GpuMirror> mirror;
  std::vector<uint64_t> synced_rows;         // rows of each segment already in HBM
  std::vector<uint64_t> bitset_rows;         // rows of each segment whose bitset columns are in HBM (UINT64_MAX: never; reset by Touch)
  std::vector<std::pair<uint64_t, uint64_t>> dirty;   // per segment: [first, last) rows updated in place since the last Sync
  std::vector<uint64_t> seg_rows;            // size() snapshot of the query being assembled
  std::map<std::string, std::unique_ptr<Session>> sessions;
  std::mutex mu;
};
std::mutex g_mu;
std::map<const void*, std::unique_ptr<Shadow>> g_shadows;

class CallbackOutput : public q::RowOutput {
public:
  CallbackOutput(SendFn send, void* ctx) : send_(send), ctx_(ctx) {}
  void Send(const Row& row) override { send_(ctx_, row); }
  void SendAsCol(const Row& col) override { send_(ctx_, col); }
private:
  SendFn send_; void* ctx_;
};
}  // namespace

struct Session {
  Shadow* shadow = nullptr;
  std::unique_ptr<q::AggregateQuery> query;
};

Session* Open(const void* table_key, const char* table_json, const char* query_json) {
  Shadow* sh;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto& slot = g_shadows[table_key];
    if (!slot) {
      slot.reset(new Shadow());
      slot->table.reset(new db::Table(util::Config(std::string(table_json)), slot->dicts));
      q::detail::ensure_device();
      std::vector<vh_col_desc> cols;
      for (auto* d : slot->table->dimensions()) cols.push_back({q::detail::dim_kind(d), d->num_type().vh_elem()});
      for (auto* m : slot->table->metrics()) {
        int elem = m->num_type().vh_elem();
        if (m->agg_type() == db::Column::BITSET) elem = m->num_type().size() == 8 ? VH_BITSET64 : VH_BITSET32;
        cols.push_back({q::detail::metric_kind(m), elem});
      }
      if (slot->table->has_hidden_count()) cols.push_back({VH_METRIC_HIDDEN_COUNT, VH_U64});
      slot->mirror.reset(new GpuMirror());
      q::detail::vh_check(vh_table_create(cols.data(), (int32_t)cols.size(), slot->table->segment_size(), 1, &slot->mirror->handle));
    }
    sh = slot.get();
  }
  std::lock_guard<std::mutex> lk(sh->mu);
  auto& s = sh->sessions[query_json];
  if (!s) {
    s.reset(new Session());
    s->shadow = sh;
    s->query.reset(new q::AggregateQuery(util::Config(std::string(query_json)), *sh->table));
  }
  sh->seg_rows.clear();
  return s.get();
}

void Sync(Session* s, uint32_t seg, uint64_t nrows, const void* const* col_ptrs) {
  Shadow* sh = s->shadow;
  std::lock_guard<std::mutex> lk(sh->mu);
  if (sh->synced_rows.size() <= seg) { sh->synced_rows.resize(seg + 1, 0); sh->dirty.resize(seg + 1, {UINT64_MAX, 0}); }
  if (sh->seg_rows.size() <= seg) sh->seg_rows.resize(seg + 1, 0);
  sh->seg_rows[seg] = nrows;
  uint64_t first = sh->synced_rows[seg], last = nrows;
  if (sh->dirty[seg].first < sh->dirty[seg].second) { first = std::min(first, sh->dirty[seg].first); last = std::max(last, std::min(nrows, sh->dirty[seg].second)); }
  if (first < last) q::detail::vh_check(vh_segment_sync_range(sh->mirror->handle, seg, first, last - first, nrows, col_ptrs));
  else if (nrows == 0 && sh->synced_rows[seg] == 0) q::detail::vh_check(vh_segment_sync(sh->mirror->handle, seg, 0, col_ptrs));
  sh->synced_rows[seg] = std::max(sh->synced_rows[seg], nrows);
  sh->dirty[seg] = {UINT64_MAX, 0};
}

bool BitsetStale(Session* s, uint32_t seg, uint64_t nrows) {
  Shadow* sh = s->shadow;
  std::lock_guard<std::mutex> lk(sh->mu);
  bool any = false;
  for (auto* m : sh->table->metrics()) any |= m->agg_type() == db::Column::BITSET;
  if (!any) return false;
  if (sh->bitset_rows.size() <= seg) sh->bitset_rows.resize(seg + 1, UINT64_MAX);
  return sh->bitset_rows[seg] != nrows;      // (Touch resets it: an in-place `|=` changed some row's set)
}

void SyncBitset(Session* s, uint32_t seg, size_t metric_index, uint64_t nrows, const uint64_t* offsets, const void* ids) {
  Shadow* sh = s->shadow;
  std::lock_guard<std::mutex> lk(sh->mu);
  const db::Metric* m = sh->table->metric(metric_index);
  if (m->agg_type() != db::Column::BITSET) throw std::invalid_argument("viya::shim::SyncBitset: metric " + m->name() + " is not a bitset");
  q::detail::vh_check(vh_segment_sync_bitset(sh->mirror->handle, seg, (int32_t)m->storage_index, nrows, offsets, ids));
  if (sh->bitset_rows.size() <= seg) sh->bitset_rows.resize(seg + 1, UINT64_MAX);
  // (the segment counts as seen once its LAST bitset metric has come in; the generated text syncs all of them back to back)
  bool last = true;
  for (auto* o : sh->table->metrics()) if (o->agg_type() == db::Column::BITSET && o->index() > metric_index) last = false;
  if (last) sh->bitset_rows[seg] = nrows;
}

void Touch(const void* table_key, uint32_t seg, uint64_t row_first, uint64_t row_last) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_shadows.find(table_key);
  if (it == g_shadows.end()) return;          // nothing mirrored yet: the first Sync copies everything
  Shadow* sh = it->second.get();
  std::lock_guard<std::mutex> lk2(sh->mu);
  if (sh->dirty.size() <= seg) { sh->synced_rows.resize(seg + 1, 0); sh->dirty.resize(seg + 1, {UINT64_MAX, 0}); }
  sh->dirty[seg].first = std::min(sh->dirty[seg].first, row_first);
  sh->dirty[seg].second = std::max(sh->dirty[seg].second, row_last);
  if (sh->bitset_rows.size() > seg) sh->bitset_rows[seg] = UINT64_MAX;
}

void BindDict(Session* s, size_t dim_index, const std::vector<std::string>* c2v) {
  Shadow* sh = s->shadow;
  std::lock_guard<std::mutex> lk(sh->mu);
  const db::Dimension* d = sh->table->dimension(dim_index);
  if (d->dim_type() != db::Column::DIM_STRING || !c2v) return;
  auto& mine = d->dict()->c2v();               // codes only grow (src/db/dictionary.h): append what is new
  for (size_t i = mine.size(); i < c2v->size(); ++i) { d->dict()->v2c()[(*c2v)[i]] = i; mine.push_back((*c2v)[i]); }
}

void Run(Session* s, const uint64_t* fargs, size_t nfargs, const uint64_t* hargs, size_t nhargs, size_t skip, size_t limit,
         SendFn send, void* ctx, Stats* stats) {
  Shadow* sh = s->shadow;
  std::vector<db::AnyNum> fa(nfargs), ha(nhargs);
  for (size_t i = 0; i < nfargs; ++i) fa[i].bits = fargs[i];
  for (size_t i = 0; i < nhargs; ++i) ha[i].bits = hargs[i];
  q::QueryStats qs;
  q::detail::Groups groups;
  const bool having_on_device = q::detail::HavingOnDevice(*s->query, skip, limit);
  {
    std::lock_guard<std::mutex> lk(sh->mu);
    q::detail::AggregateOnMirror(*s->query, sh->mirror->handle, sh->seg_rows, having_on_device, fa, ha, skip, limit, -1, groups, qs);
  }
  CallbackOutput out(send, ctx);
  q::detail::PostAggregate(*s->query, groups, having_on_device, ha, skip, limit, out, qs);
  if (stats) *stats = Stats{qs.scanned_segments, qs.scanned_recs, qs.aggregated_recs, qs.output_recs, qs.device_flags, qs.retries, qs.scan_kernel_ms};
}

void Close(const void* table_key) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_shadows.erase(table_key);
}

}  // namespace shim
}  // namespace viya

// shim_session.cc — include/viya_shim.h: the host side of a GENERATED viya_query_agg (tools/gen_shim_tu.py).
//
// The generated function lives inside the reference process and owns nothing but pointers into the reference's segments
// and dictionaries. Everything the GPU path needs beyond them is kept here, keyed by the address of the reference's
// db::Table: a descriptor-only shadow of the table (column types, rollup rules — parsed from the same JSON the reference
// built its table from), the HBM mirror, and one parsed query per query text (the reference caches one compiled
// function per generated source, src/codegen/compiler.cc:97-144).
//
// Threading is the reference's: `query_threads` read-pool threads enter the generated function at once, on one table and on one query
// text (src/db/database.cc:28-34, src/server/http/service.cc:115-135), while ONE writer thread upserts (service.cc:103) — appending rows
// to the last segment and updating metrics of existing rows in place (src/codegen/db/upsert.cc:384-411). So:
//   * a Session is ONE CALL of the generated function: its size() snapshot and column addresses live in it and nowhere else; Open hands
//     out a fresh one, Release ends it. Two calls never see each other's snapshot.
//   * what is shared per table (Shadow) is what really is one per table: the mirror, how many rows of every segment it holds, the rows
//     Touched since, which segments are registered with the device. One mutex, held while a call's batch of dirty ranges is assembled
//     and handed to vh_table_sync_batch (ONE kernel launch, no wait) — not while its query runs or its rows are formatted.
//   * the mirror holds the LARGEST snapshot any call brought so far; every call scans exactly the rows of ITS snapshot (vh_plan.seg_rows).
//     A row updated in place while queries run is seen old or new — the race the reference's readers have with its writer.
//   * dictionaries: BindDict appends under an exclusive lock, planning and formatting read under a shared one.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <stdexcept>

#include "../../include/viya_shim.h"
#include "gpu_internal.h"

namespace viya {
namespace shim {

namespace q = viya::query;
using q::detail::GpuMirror;

namespace {
struct Prepared {                              // one per (table, query text): the analogue of the cached compiled function
  std::unique_ptr<q::AggregateQuery> query;
};
typedef std::pair<uint64_t, uint64_t> Range;   // [first, last) rows
struct Shadow {
  db::Dictionaries dicts;
  std::unique_ptr<db::Table> table;            // descriptors only: it never holds a row
  std::unique_ptr<GpuMirror> mirror;
  size_t ncols = 0;                            // storage columns: dimensions, metrics, hidden count
  bool any_bitset = false;
  std::vector<uint64_t> synced_rows;           // rows of each segment already in HBM
  // bitset columns (CSR mirrors, replaced whole): rows of each segment's CSR in HBM — it never shrinks: a call with a smaller snapshot walks
  // the rows the mirror already has, or another call's query would read past its end —, whether it is current, and a count of the Touches
  // that hit the segment (a Touch DURING a walk leaves the segment stale). One call walks at a time (bitset_mu: decide, walk, install).
  std::vector<uint64_t> bitset_have;
  std::vector<char> bitset_clean;
  std::vector<uint64_t> bitset_touches;
  std::mutex bitset_mu;
  std::vector<std::vector<Range>> dirty;       // per segment: rows updated in place since they were last shipped
  std::vector<const void*> pinned;             // per segment: base of the registered object (NULL: not registered), or kPinRefused
  std::map<std::string, std::unique_ptr<Prepared>> prepared;
  std::mutex mu;                               // everything above
  std::shared_mutex dict_mu;                   // `dicts` (BindDict appends; plans and formatting read)
  void grow(uint32_t nseg) {
    if (synced_rows.size() < nseg) { synced_rows.resize(nseg, 0); dirty.resize(nseg); pinned.resize(nseg, nullptr); bitset_have.resize(nseg, 0); bitset_clean.resize(nseg, 0); bitset_touches.resize(nseg, 0); }
  }
};
const void* const kPinRefused = reinterpret_cast<const void*>(uintptr_t(1));
std::mutex g_mu;
std::map<const void*, std::unique_ptr<Shadow>> g_shadows;

bool pin_enabled() {
  static const bool on = [] { const char* e = getenv("VIYA_SHIM_PIN"); return !e || atoi(e) != 0; }();
  return on;
}

// Sorted, disjoint, and no two ranges closer than `gap` rows (shipping a few clean rows is cheaper than another run).
void normalise(std::vector<Range>& v, uint64_t gap) {
  if (v.size() < 2) return;
  std::sort(v.begin(), v.end());
  size_t o = 0;
  for (size_t i = 1; i < v.size(); ++i) {
    if (v[i].first <= v[o].second + gap) v[o].second = std::max(v[o].second, v[i].second);
    else v[++o] = v[i];
  }
  v.resize(o + 1);
}

class CallbackOutput : public q::RowOutput {
public:
  CallbackOutput(SendFn send, void* ctx) : send_(send), ctx_(ctx) {}
  void Send(const Row& row) override { send_(ctx_, row); }
  void SendAsCol(const Row& col) override { send_(ctx_, col); }
private:
  SendFn send_; void* ctx_;
};
}  // namespace

struct Session {                               // ONE call of the generated function
  Shadow* shadow = nullptr;
  Prepared* prepared = nullptr;
  std::vector<uint64_t> seg_rows;              // the size() snapshot this call took
  std::vector<const void*> cols;               // ncols column addresses per segment, in the order Sync saw them
  std::vector<std::pair<const void*, size_t>> objects;   // per segment: the Segment object (Pin), for registration
  std::unique_lock<std::mutex> walk;           // held from a BitsetStale that said "walk" to the SyncBitset of the segment's last bitset metric
  uint64_t walk_touches = 0;
};

Session* Open(const void* table_key, const char* table_json, const char* query_json) {
  Shadow* sh;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto& slot = g_shadows[table_key];
    if (!slot) {
      std::unique_ptr<Shadow> fresh(new Shadow());
      fresh->table.reset(new db::Table(util::Config(std::string(table_json)), fresh->dicts));
      q::detail::ensure_device();
      std::vector<vh_col_desc> cols;
      for (auto* d : fresh->table->dimensions()) cols.push_back({q::detail::dim_kind(d), d->num_type().vh_elem()});
      for (auto* m : fresh->table->metrics()) {
        int elem = m->num_type().vh_elem();
        if (m->agg_type() == db::Column::BITSET) { elem = m->num_type().size() == 8 ? VH_BITSET64 : VH_BITSET32; fresh->any_bitset = true; }
        cols.push_back({q::detail::metric_kind(m), elem});
      }
      if (fresh->table->has_hidden_count()) cols.push_back({VH_METRIC_HIDDEN_COUNT, VH_U64});
      fresh->ncols = cols.size();
      fresh->mirror.reset(new GpuMirror());
      q::detail::vh_check(vh_table_create(cols.data(), (int32_t)cols.size(), fresh->table->segment_size(), 1, &fresh->mirror->handle));
      slot = std::move(fresh);
    }
    sh = slot.get();
  }
  std::unique_ptr<Session> s(new Session());
  s->shadow = sh;
  {
    std::lock_guard<std::mutex> lk(sh->mu);
    auto& p = sh->prepared[query_json];
    if (!p) {
      std::shared_lock<std::shared_mutex> dl(sh->dict_mu);
      p.reset(new Prepared());
      p->query.reset(new q::AggregateQuery(util::Config(std::string(query_json)), *sh->table));
    }
    s->prepared = p.get();
  }
  return s.release();
}

void Release(Session* s) { delete s; }

void Pin(Session* s, uint32_t seg, const void* object, size_t bytes) {
  if (s->objects.size() <= seg) s->objects.resize(seg + 1, {nullptr, 0});
  s->objects[seg] = {object, bytes};
}

void Sync(Session* s, uint32_t seg, uint64_t nrows, const void* const* col_ptrs) {
  const size_t nc = s->shadow->ncols;
  if (s->seg_rows.size() <= seg) { s->seg_rows.resize(seg + 1, 0); s->cols.resize((size_t)(seg + 1) * nc, nullptr); }
  s->seg_rows[seg] = nrows;
  std::copy(col_ptrs, col_ptrs + nc, s->cols.begin() + (size_t)seg * nc);
}

uint64_t BitsetStale(Session* s, uint32_t seg, uint64_t nrows) {
  Shadow* sh = s->shadow;
  if (!sh->any_bitset) return 0;
  if (s->walk.owns_lock()) s->walk.unlock();   // (a walk that never reached its last SyncBitset)
  s->walk = std::unique_lock<std::mutex>(sh->bitset_mu);
  uint64_t rows = 0;
  bool stale;
  {
    std::lock_guard<std::mutex> lk(sh->mu);
    sh->grow(seg + 1);
    // current: every row of this call's snapshot is there and no row's set grew in place (Touch) since the CSR was built
    stale = !sh->bitset_clean[seg] || sh->bitset_have[seg] < nrows;
    if (stale) { rows = std::max(nrows, sh->bitset_have[seg]); s->walk_touches = sh->bitset_touches[seg]; }
  }
  if (stale && rows == 0) {                    // an empty segment seen for the first time: its (empty) CSR is created here, there is nothing to walk
    const uint64_t zero = 0;
    for (auto* m : sh->table->metrics())
      if (m->agg_type() == db::Column::BITSET) q::detail::vh_check(vh_segment_sync_bitset(sh->mirror->handle, seg, (int32_t)m->storage_index, 0, &zero, nullptr));
    std::lock_guard<std::mutex> lk(sh->mu);
    sh->bitset_clean[seg] = sh->bitset_touches[seg] == s->walk_touches;
  }
  if (!rows) s->walk.unlock();
  return rows;
}

void SyncBitset(Session* s, uint32_t seg, size_t metric_index, uint64_t nrows, const uint64_t* offsets, const void* ids) {
  Shadow* sh = s->shadow;
  const db::Metric* m = sh->table->metric(metric_index);
  if (m->agg_type() != db::Column::BITSET) throw std::invalid_argument("viya::shim::SyncBitset: metric " + m->name() + " is not a bitset");
  if (!s->walk.owns_lock()) throw std::logic_error("viya::shim::SyncBitset without a BitsetStale that asked for the walk");
  q::detail::vh_check(vh_segment_sync_bitset(sh->mirror->handle, seg, (int32_t)m->storage_index, nrows, offsets, ids));
  // (the segment counts as seen once its LAST bitset metric has come in; the generated text syncs all of them back to back)
  bool last = true;
  for (auto* o : sh->table->metrics()) if (o->agg_type() == db::Column::BITSET && o->index() > metric_index) last = false;
  if (!last) return;
  {
    std::lock_guard<std::mutex> lk(sh->mu);
    sh->grow(seg + 1);
    sh->bitset_have[seg] = std::max(sh->bitset_have[seg], nrows);
    sh->bitset_clean[seg] = sh->bitset_touches[seg] == s->walk_touches;
  }
  s->walk.unlock();
}

void Touch(const void* table_key, uint32_t seg, uint64_t row_first, uint64_t row_last) {
  Shadow* sh = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_shadows.find(table_key);
    if (it == g_shadows.end()) return;        // nothing mirrored yet: the first Sync copies everything
    sh = it->second.get();
  }
  std::lock_guard<std::mutex> lk2(sh->mu);
  sh->grow(seg + 1);
  auto& d = sh->dirty[seg];
  if (!d.empty() && row_first >= d.back().first && row_first <= d.back().second) d.back().second = std::max(d.back().second, row_last);   // the next row of a run
  else d.emplace_back(row_first, row_last);
  if (d.size() > 4096) { uint64_t gap = 64; do { normalise(d, gap); gap *= 4; } while (d.size() > 1024); }   // (a batch that sprays one segment: coarser ranges)
  sh->bitset_clean[seg] = 0; ++sh->bitset_touches[seg];
}

void BindDict(Session* s, size_t dim_index, const std::vector<std::string>* c2v) {
  Shadow* sh = s->shadow;
  const db::Dimension* d = sh->table->dimension(dim_index);
  if (d->dim_type() != db::Column::DIM_STRING || !c2v) return;
  {
    std::shared_lock<std::shared_mutex> rl(sh->dict_mu);
    if (d->dict()->c2v().size() >= c2v->size()) return;
  }
  std::unique_lock<std::shared_mutex> wl(sh->dict_mu);
  auto& mine = d->dict()->c2v();               // codes only grow (src/db/dictionary.h): append what is new
  for (size_t i = mine.size(); i < c2v->size(); ++i) { d->dict()->v2c()[(*c2v)[i]] = i; mine.push_back((*c2v)[i]); }
}

namespace {
// What this call's snapshot and the rows Touched since the last batch ask of the mirror, as ONE vh_table_sync_batch.
void ship(Shadow* sh, Session* s) {
  const uint32_t nseg = (uint32_t)s->seg_rows.size();
  const size_t nc = sh->ncols;
  std::lock_guard<std::mutex> lk(sh->mu);
  sh->grow(nseg);
  std::vector<vh_sync_item> items;
  // The shadow's claims (rows mirrored, ranges clean) change only AFTER the batch went through: a batch that fails must leave the touched
  // ranges on the books and the row counts where the mirror really is, or later queries would read stale metrics.
  std::vector<std::pair<uint32_t, uint64_t>> rows_after;                 // (segment, synced_rows once the batch is resident)
  std::vector<std::pair<uint32_t, std::vector<Range>>> taken;            // dirty ranges this batch carries, by segment
  for (uint32_t seg = 0; seg < nseg; ++seg) {
    const uint64_t nrows = s->seg_rows[seg], have = sh->synced_rows[seg];
    auto& d = sh->dirty[seg];
    const bool first_sight = have == 0 && sh->pinned[seg] == nullptr;
    if (nrows <= have && d.empty() && !(first_sight && nrows == 0)) continue;        // the common case: nothing happened to this segment
    const void* const* cols = s->cols.data() + (size_t)seg * nc;
    if (pin_enabled() && sh->pinned[seg] == nullptr && seg < s->objects.size() && s->objects[seg].first) {
      // the Segment object never moves (store.cc:203-356, SegmentStore keeps pointers): registered once, every later range of it is read in place
      sh->pinned[seg] = vh_host_register(s->objects[seg].first, s->objects[seg].second) == VH_OK ? s->objects[seg].first : kPinRefused;
    } else if (sh->pinned[seg] == nullptr) sh->pinned[seg] = kPinRefused;
    const uint64_t size_after = std::max(have, nrows);
    if (!d.empty()) {
      normalise(d, 64);
      for (const Range& r : d) {
        // rows beyond what the mirror holds travel with the append below (or with a later call's): they are copied whole then
        const uint64_t a = r.first, b = std::min(r.second, have);
        if (a < b) items.push_back(vh_sync_item{seg, VH_SYNC_METRICS_ONLY, a, b - a, have, cols});
      }
      taken.emplace_back(seg, std::move(d));
      d.clear();
    }
    if (nrows > have) items.push_back(vh_sync_item{seg, 0u, have, nrows - have, nrows, cols});
    else if (first_sight && nrows == 0) items.push_back(vh_sync_item{seg, 0u, 0, 0, 0, cols});
    rows_after.emplace_back(seg, size_after);
  }
  if (!items.empty()) {
    const int rc = vh_table_sync_batch(sh->mirror->handle, items.data(), (uint32_t)items.size());
    if (rc != VH_OK) {
      // nothing is claimed: the ranges go back in front of whatever Touch added meanwhile (it cannot have: sh->mu is held), the row counts stay
      for (auto& t : taken) { auto& d = sh->dirty[t.first]; d.insert(d.begin(), t.second.begin(), t.second.end()); }
      q::detail::vh_check(rc);        // throws with vh_last_error()
    }
  }
  for (const auto& ra : rows_after) sh->synced_rows[ra.first] = ra.second;
}
}  // namespace

void Run(Session* s, const uint64_t* fargs, size_t nfargs, const uint64_t* hargs, size_t nhargs, size_t skip, size_t limit,
         SendFn send, void* ctx, Stats* stats) {
  Shadow* sh = s->shadow;
  std::vector<db::AnyNum> fa(nfargs), ha(nhargs);
  for (size_t i = 0; i < nfargs; ++i) fa[i].bits = fargs[i];
  for (size_t i = 0; i < nhargs; ++i) ha[i].bits = hargs[i];
  q::QueryStats qs;
  q::detail::Groups groups;
  q::AggregateQuery& query = *s->prepared->query;
  const bool having_on_device = q::detail::HavingOnDevice(query, skip, limit);
  const auto t0 = std::chrono::steady_clock::now();
  ship(sh, s);
  const double sync_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  // the library plans and launches under its own per-table lock and runs queries of different threads side by side (include/viya_hip.h,
  // "Threading"); nothing of the shadow is needed from here on but the dictionaries
  std::shared_lock<std::shared_mutex> dl(sh->dict_mu);
  q::detail::AggregateOnMirror(query, sh->mirror->handle, s->seg_rows, having_on_device, fa, ha, skip, limit, -1, groups, qs);
  CallbackOutput out(send, ctx);
  q::detail::PostAggregate(query, groups, having_on_device, ha, skip, limit, out, qs);
  if (stats) *stats = Stats{qs.scanned_segments, qs.scanned_recs, qs.aggregated_recs, qs.output_recs, qs.device_flags, qs.retries, qs.scan_kernel_ms, sync_ms};
}

void Close(const void* table_key) {
  std::unique_ptr<Shadow> gone;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_shadows.find(table_key);
    if (it == g_shadows.end()) return;
    gone = std::move(it->second);
    g_shadows.erase(it);
  }
  for (const void* p : gone->pinned) if (p && p != kPinRefused) (void)vh_host_unregister(p);
}

}  // namespace shim
}  // namespace viya

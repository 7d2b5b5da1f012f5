// vhh_table.h — host side of libviya_hip, part of viya_hip.hip's translation unit (included there, in order; not a stand-alone header):
// the table mirror: column arenas, per-segment stats, execution contexts (stream, scratch, staging, events); vh_table_create / vh_table_destroy.
// -------------------------------------------------------------------- table
struct VhColumn {
  int kind = 0, elem = 0, esize = 0;
  char* base = nullptr;     // arena: cap_seg x stride bytes (+ tail pad)
  uint64_t stride = 0;      // bytes between segments
  // bitset CSR mirrors (one pair per segment)
  std::vector<uint64_t*> bs_offsets;
  std::vector<void*> bs_values;
  std::vector<uint64_t> bs_nvalues;
  std::vector<uint32_t*> bs_offsets32;  // the same offsets as 32-bit words when the segment holds < 2^32 ids (nullptr otherwise): what the compiled
                                        // scan of the hashed partitioning reads — 4 instead of 8 bytes per row of a stream every query of the set takes in full
  std::vector<uint64_t> bs_maxid;      // an upper bound of the segment's ids (what the packed tuples of the hashed partitioning are sized from)
};
struct VhSegStat {          // order keys as produced by seg_minmax_kernel
  uint64_t lo = ~0ull, hi = 0;
};
// Execution context: everything ONE in-flight query needs besides the table's columns — a stream, device scratch,
// pinned staging, events. A table keeps a pool of them; a vh_result owns one from launch until vh_result_free, so
// queries of different threads on one table overlap on the device (the reference's read_pool runs queries of one table
// concurrently: src/db/database.cc:28-34, src/server/http/service.cc:119) and a handle's device state and host view are
// never reused under it.
struct VhExec {
  hipStream_t own_stream = nullptr;
  char* scratch = nullptr; size_t scratch_bytes = 0;
  uint32_t* h_segrows = nullptr; size_t h_segrows_cap = 0;
  unsigned long long* h_counters = nullptr;     // pinned: 16 words of counters + 64 words for a big result's header
  char* d_sample = nullptr; size_t d_sample_bytes = 0;   // selectivity probe: counters + presence + seg rows
  char* h_out[2] = {nullptr, nullptr}; size_t h_out_bytes[2] = {0, 0}; int h_out_next = 0;  // pinned result staging (two alternate: a
                                                                                            // zero-copy view outlives vh_result_free until the second-next query)
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  // streamed delivery of big results (hashed partitioning, VhHpArgs::nchunks): the aggregation's chunk launches alternate between the query's
  // stream and `aux` (the tail of one chunk overlaps the start of the next), finished chunks leave on `copy`; created on first use
  hipStream_t aux = nullptr, copy = nullptr;
  hipEvent_t ev_fork = nullptr, ev_chunk[VH_HP_CHUNKS] = {};
  unsigned long long* h_chunk = nullptr;        // pinned: rows of chunk c, written by publish_count_kernel
  bool busy = false;
  // an externally owned stream (vh_set_stream) carries all work; otherwise every context has its own
  hipStream_t stream() const { return g_ctx.stream != g_ctx.own_stream ? g_ctx.stream : own_stream; }
};

// Payload projection (vh_table_pack): a row-major copy of a few columns, see pack_kernel.
struct VhPack {
  std::vector<int> cols;            // table column indices, in record order (widest first)
  std::vector<uint32_t> off;        // byte offset of each column inside a record
  std::vector<uint8_t> width;       // bytes the column's values take in a record (compressed: fewer than its element size)
  bool compressed = false;          // integer columns stored at the width their values need; only the per-query compiled kernels read these
  // BIT-FIELD records (round 4; compressed projections whose columns are all non-negative integers): every column at the BITS its values
  // need, packed into one 4- or 8-byte word — C3's (d0 10 bits, d1 7, m0 10, count 2) fits 4 bytes: 32 records per 128-byte line instead of 16
  bool bits = false;
  std::vector<uint8_t> bitoff, bitw;
  uint32_t rec_bytes = 0;           // power of two, 8..64 (bit-field records: 4 or 8)
  char* base = nullptr; uint64_t stride = 0; uint32_t cap_seg = 0;
  std::vector<uint64_t> seg_mod;    // value of vh_table::seg_mod[s] the segment was packed at (0: never)
  uint64_t applied_epoch = 0;       // every change of the table's journal up to this epoch is in the records
  bool automatic = false;
  int col_index(int col) const { for (size_t i = 0; i < cols.size(); ++i) if (cols[i] == col) return (int)i; return -1; }
};
// Narrow copy of a predicate column (vh_table_narrow): an unsigned 32-bit column whose values fit 8 or 16 bits, kept a second time
// at that width. The register-resident scan kernels stream the copy instead of the arena — a predicate column is read in full by
// every query that filters on it, so its bytes are the floor of the scan (C3: 12 of 18.75 GB per query).
struct VhNarrow {
  int col = -1, width = 0;          // bytes per element: 1 or 2
  char* base = nullptr; uint64_t stride = 0; uint32_t cap_seg = 0;
  std::vector<uint64_t> seg_mod;    // vh_table::seg_mod[s] the segment was copied at (0: never)
  uint64_t applied_epoch = 0;
  bool automatic = false;
};
// Bit-packed predicate projection (vh_table_predpack): the predicate columns of a query shape as bit fields of one word per row (each at
// the bits its recorded min / max need), kept as byte planes — what the per-query compiled scan streams instead of the columns or their
// narrow copies. C3: d2 (2 bits) + d3 (10) + d4 (10) = 22 bits -> a 2-byte and a 1-byte plane: 3 bytes per row instead of 5 (12 from the arenas).
struct VhPredPack {
  std::vector<int> cols;             // table columns, ascending
  std::vector<uint8_t> bitoff, bitw; // each column's field in the row word
  int nplanes = 0;
  int pwidth[4] = {}, ppos[4] = {};  // bytes per row of plane q, first bit of the word it holds
  char* pbase[4] = {}; uint64_t pstride[4] = {};
  uint32_t cap_seg = 0;
  std::vector<uint64_t> seg_mod; uint64_t applied_epoch = 0;
  bool automatic = false;
  // BIT-SLICED form: one plane per BIT of the word, one bit per row (32 rows = one 4-byte word of a plane); pbase[0] is the whole arena,
  // pstride[0] the bytes between segments, `pitch` the bytes between planes inside a segment; `bits` planes.
  bool sliced = false; uint32_t bits = 0; uint64_t pitch = 0;
  uint32_t bytes_per_row() const { uint32_t b = 0; for (int q = 0; q < nplanes; ++q) b += (uint32_t)pwidth[q]; return b; }
  uint32_t bits_per_row() const { return sliced ? bits : 8u * bytes_per_row(); }
};
// What a sync did to a segment's columns: rows [first, last) at sync epoch `epoch` (the table's journal; derived layouts replay it).
struct VhChange { uint64_t epoch; uint32_t seg, first, last; };
struct vh_table {
  std::vector<VhColumn> cols;
  uint64_t segment_rows = 0;
  uint64_t padded_rows = 0;
  uint32_t cap_seg = 0;
  uint32_t nseg = 0;
  std::vector<uint64_t> seg_rows;               // last synced row count
  std::vector<std::vector<VhSegStat>> stats;    // [col][seg]
  // per-query resources live in execution contexts (grow-only pool)
  std::vector<std::unique_ptr<VhExec>> execs;
  std::mutex pool_mu; std::condition_variable pool_cv;
  char* d_stats = nullptr; size_t d_stats_bytes = 0;      // vh_segment_sync*: min/max pass (its own buffer: a sync never touches a query's scratch)
  // vh_table_sync_batch: run descriptors and their result slots in ONE pinned block, a ring for small runs out of unregistered host memory,
  // and what of the last batch still has to be merged into `stats` (sync_resolve: the first planner — or sync — that comes after it waits)
  char* h_sync = nullptr; size_t h_sync_bytes = 0;
  char* h_stage = nullptr; size_t h_stage_bytes = 0;
  hipEvent_t sync_ev = nullptr;
  struct SyncPending { uint32_t col, seg, desc_first, desc_n; };
  std::vector<SyncPending> sync_pending;
  bool sync_inflight = false;
  uint64_t sync_batches = 0, sync_descs = 0, sync_bytes_pulled = 0, sync_bytes_staged = 0, sync_bytes_dma = 0;   // vh_table_sync_stats
  std::vector<uint16_t> ship_all, ship_metrics;          // vh_table_sync_batch: the columns an item ships (every fixed-width one / the metrics alone)
  std::vector<uint64_t> sync_rows_now;                   // ... and its scratch (rows mirrored per named segment while a batch is validated)
  std::map<std::string, uint64_t> groups_seen;           // group-column signature -> groups of the last query (hash sizing)
  std::set<std::string> hpart_hopeless;                  // group-column signatures whose hashed partitioning ended on the plain hash table (a group holding more ids than a range's LDS set takes in any number of passes): later queries of the shape start there
  std::set<std::string> part_clustered;                  // group-column signatures whose survivors came clustered under the piecewise writers: positional extent chunks (VhPlanDev::ext_waves) overflowed although the pool had room — later queries of the shape take their extents off the shared cursor at once
  std::map<std::string, std::pair<uint64_t, uint64_t>> sel_cache;   // filter signature + table state -> (passed, sampled) of the selectivity probe
  std::vector<std::unique_ptr<VhPack>> packs;
  std::vector<std::unique_ptr<VhNarrow>> narrows;
  std::vector<std::unique_ptr<VhPredPack>> predpacks;
  std::map<std::string, uint32_t> ppred_seen;             // predicate column set -> compiled-kernel queries that filtered on it (automatic predicate projections)
  std::map<int, uint32_t> pred_seen;                     // column -> queries that filtered on it (automatic narrow copies)
  std::vector<uint64_t> seg_mod;                          // sync_epoch of the last change to a segment's columns
  // Derived layouts follow the arenas by ROW RANGE: every sync appends what it touched; a layout that was current at epoch e re-derives the
  // ranges journalled since — one launch over a list of jobs — instead of every segment whose stamp moved (an upsert batch dirties hundreds
  // of segments by a few rows each). Entries older than `journal_floor` were dropped: a layout behind that re-derives whole segments.
  std::vector<VhChange> journal; uint64_t journal_floor = 0;
  char* h_jobs = nullptr; size_t h_jobs_bytes = 0, h_jobs_used = 0;     // pinned: the job lists of the refreshes enqueued since the stream was last waited for
  hipEvent_t derived_ev = nullptr; bool derived_pending = false;         // a refresh is enqueued on g_ctx.stream: the next query's stream waits for it
  unsigned int* d_packflag = nullptr;                      // pack_kernel's "a value outgrew its stored width" word
  std::map<std::string, uint32_t> gather_seen;            // payload column set -> low-selectivity queries seen (automatic packs)
  uint64_t sync_epoch = 0;   // bumped by every vh_segment_sync / generate: invalidates cached estimates
  std::mutex mu;             // table metadata, column arenas, projections, planner caches: held while a query is PLANNED and
                             // LAUNCHED and by every sync; not while a launched query runs or is read back
  uint64_t device_bytes = 0;
};

static const uint32_t VH_MAX_SEGMENTS = 1u << 24;   // (segment << 32 | row) positions and u32 segment loops stay far from overflow
static bool is_dim(int kind) { return kind <= VH_DIM_BOOLEAN; }
static bool is_bitset_elem(int e) { return e == VH_BITSET32 || e == VH_BITSET64; }

static void trace_alloc(const char* what, const void* p, size_t bytes) {     // VH_TRACE_ALLOC=1: where the big buffers land (placement experiments)
  if (knobs().trace_alloc) fprintf(stderr, "vh alloc %s %p %zu\n", what, p, bytes);
}

static int table_grow(vh_table* t, uint32_t need_seg) {
  if (need_seg <= t->cap_seg) return VH_OK;
  uint32_t ncap = std::max<uint32_t>(need_seg, std::max<uint32_t>(4, t->cap_seg * 2));
  for (auto& c : t->cols) {
    if (is_bitset_elem(c.elem)) {
      c.bs_offsets.resize(ncap, nullptr); c.bs_offsets32.resize(ncap, nullptr); c.bs_values.resize(ncap, nullptr); c.bs_nvalues.resize(ncap, 0); c.bs_maxid.resize(ncap, 0);
      continue;
    }
    char* nb = nullptr;
    const size_t bytes = (size_t)ncap * c.stride + 256;
    HIP_TRY(hipMalloc(&nb, bytes));
    trace_alloc("column", nb, bytes);
    if (c.base && t->nseg) {
      HIP_TRY(hipMemcpyAsync(nb, c.base, (size_t)t->nseg * c.stride, hipMemcpyDeviceToDevice, g_ctx.stream));
      HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    }
    if (c.base) { HIP_TRY(hipFree(c.base)); t->device_bytes -= (size_t)t->cap_seg * c.stride + 256; }
    c.base = nb;
    t->device_bytes += bytes;
  }
  t->cap_seg = ncap;
  t->seg_rows.resize(ncap, 0);
  t->seg_mod.resize(ncap, 0);
  for (auto& s : t->stats) s.resize(ncap);
  return VH_OK;
}

extern "C" int vh_table_create(const vh_col_desc* cols, int32_t ncols, uint64_t segment_rows,
                               uint32_t reserve_segments, vh_table** out) {
  if (!g_ctx.inited) return vh_fail(VH_E_INVALID, "vh_init has not been called");
  VH_ENTER();
  if (!cols || ncols <= 0 || !out || segment_rows == 0 || segment_rows > 0xFFFF0000ull)
    return vh_fail(VH_E_INVALID, "vh_table_create: bad arguments");
  vh_table* t = new vh_table();
  t->segment_rows = segment_rows;
  t->padded_rows = (segment_rows + 63) / 64 * 64;
  t->cols.resize(ncols);
  t->stats.resize(ncols);
  for (int i = 0; i < ncols; ++i) {
    VhColumn& c = t->cols[i];
    c.kind = cols[i].kind; c.elem = cols[i].elem;
    const bool dim_kind = c.kind >= VH_DIM_STRING && c.kind <= VH_DIM_BOOLEAN, metric_kind = c.kind >= VH_METRIC_MAX && c.kind <= VH_METRIC_HIDDEN_COUNT;
    if ((!dim_kind && !metric_kind) || (is_bitset_elem(c.elem) != (c.kind == VH_METRIC_BITSET))) {
      delete t;
      return vh_fail(VH_E_INVALID, "column %d: bad kind %d / element type %d", i, cols[i].kind, cols[i].elem);
    }
    if (is_bitset_elem(c.elem)) { c.esize = 0; continue; }
    c.esize = vh_elem_size(c.elem);
    if (!c.esize) { delete t; return vh_fail(VH_E_INVALID, "column %d: bad element type %d", i, c.elem); }
    c.stride = t->padded_rows * c.esize;
  }
  int rc = table_grow(t, std::max<uint32_t>(1, reserve_segments));
  if (rc) { vh_table_destroy(t); return rc; }
  *out = t;
  return VH_OK;
}

// ------------------------------------------------------------------ execution contexts
static void exec_free(VhExec* x) {
  if (x->scratch) (void)hipFree(x->scratch);
  if (x->d_sample) (void)hipFree(x->d_sample);
  for (auto& hp : x->h_out) if (hp) (void)hipHostFree(hp);
  if (x->h_segrows) (void)hipHostFree(x->h_segrows);
  if (x->h_counters) (void)hipHostFree(x->h_counters);
  for (auto& e : x->ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : x->ev_chunk) if (e) (void)hipEventDestroy(e);
  if (x->ev_fork) (void)hipEventDestroy(x->ev_fork);
  if (x->h_chunk) (void)hipHostFree(x->h_chunk);
  if (x->aux) (void)hipStreamDestroy(x->aux);
  if (x->copy) (void)hipStreamDestroy(x->copy);
  if (x->own_stream) (void)hipStreamDestroy(x->own_stream);
}
// A free context of the table's pool, a new one while the pool may grow, else wait for one to come back.
// (Round 2 timed partitioned plans on three contexts and kept the one whose scratch "landed best": the tuple pool's placement decided
// 10 % of phase 1 when tuples left as partial lines. Whole-line tuple writes removed the sensitivity — eight processes, trials 1 vs 3:
// 2.50-2.54 vs 2.42-2.53 ms, profiles/r03/NOTES.md — and with it the three scratch buffers per table.)
static int exec_acquire(vh_table* t, VhExec** out, bool wait = true) {      // wait = false: a context that is free or can be made now, or VH_E_NOMEM at once (no message)
  const size_t max_exec = (size_t)knobs().max_exec;
  std::unique_lock<std::mutex> lk(t->pool_mu);
  for (;;) {
    for (auto& x : t->execs) if (!x->busy) { x->busy = true; *out = x.get(); return VH_OK; }
    if (t->execs.size() < max_exec) break;
    if (!wait) return VH_E_NOMEM;
    if (t->pool_cv.wait_for(lk, std::chrono::seconds(60)) == std::cv_status::timeout)
      return vh_fail(VH_E_NOMEM, "all %zu execution contexts of this table are held by live vh_result / running queries (vh_result_free them)", max_exec);
  }
  std::unique_ptr<VhExec> x(new VhExec());
  hipError_t he = hipStreamCreateWithFlags(&x->own_stream, hipStreamNonBlocking);
  if (he == hipSuccess) he = hipHostMalloc((void**)&x->h_counters, (16 + 64) * sizeof(unsigned long long), hipHostMallocDefault);
  for (auto& e : x->ev) if (he == hipSuccess) he = hipEventCreate(&e);
  if (he != hipSuccess) { exec_free(x.get()); return vh_fail(VH_E_DEVICE, "execution context: stream / pinned staging / events: %s", hipGetErrorString(he)); }
  x->busy = true;
  *out = x.get();
  t->execs.push_back(std::move(x));
  return VH_OK;
}
static int exec_streaming(VhExec* x) {       // what a streamed result needs on top of a context's stream; once per context
  if (x->copy) return VH_OK;
  HIP_TRY(hipStreamCreateWithFlags(&x->aux, hipStreamNonBlocking));
  { int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi); HIP_TRY(hipStreamCreateWithPriority(&x->copy, hipStreamNonBlocking, 0)); }
  HIP_TRY(hipEventCreateWithFlags(&x->ev_fork, hipEventDisableTiming));
  for (auto& e : x->ev_chunk) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIP_TRY(hipHostMalloc((void**)&x->h_chunk, VH_HP_CHUNKS * sizeof(unsigned long long), hipHostMallocCoherent));
  return VH_OK;
}
static void exec_release(vh_table* t, VhExec* x) {
  if (!x) return;
  { std::lock_guard<std::mutex> lk(t->pool_mu); x->busy = false; }
  t->pool_cv.notify_one();
}
// Before column arenas, CSR mirrors or projections are replaced: wait for every launched query that may still read them.
// Called with t->mu held (no new launch can start).
static void table_quiesce(vh_table* t) {
  std::lock_guard<std::mutex> lk(t->pool_mu);
  for (auto& x : t->execs) if (x->busy) (void)hipStreamSynchronize(x->stream());
}

// A sync changed rows [first, last) of a segment's columns: stamp the segment and tell the journal.
static void table_note_change(vh_table* t, uint32_t seg, uint64_t first, uint64_t last, bool new_epoch = true) {
  if (new_epoch) ++t->sync_epoch;
  t->seg_mod[seg] = t->sync_epoch;
  if (t->journal.size() >= (1u << 18)) {          // keep the newer half; layouts older than the floor fall back to whole segments
    const size_t drop = t->journal.size() / 2;
    t->journal_floor = t->journal[drop - 1].epoch;
    t->journal.erase(t->journal.begin(), t->journal.begin() + (long)drop);
  }
  t->journal.push_back(VhChange{t->sync_epoch, seg, (uint32_t)first, (uint32_t)last});
}

extern "C" void vh_table_destroy(vh_table* t) {
  if (!t) return;
  VH_ENTER();
  (void)hipStreamSynchronize(g_ctx.stream);
  for (auto& x : t->execs) { (void)hipStreamSynchronize(x->stream()); exec_free(x.get()); }
  for (auto& c : t->cols) {
    if (c.base) (void)hipFree(c.base);
    for (auto p : c.bs_offsets) if (p) (void)hipFree(p);
    for (auto p : c.bs_offsets32) if (p) (void)hipFree(p);
    for (auto p : c.bs_values) if (p) (void)hipFree(p);
  }
  if (t->d_stats) (void)hipFree(t->d_stats);
  if (t->sync_ev) { (void)hipEventSynchronize(t->sync_ev); (void)hipEventDestroy(t->sync_ev); }
  if (t->h_sync) (void)hipHostFree(t->h_sync);
  if (t->h_stage) (void)hipHostFree(t->h_stage);
  if (t->d_packflag) (void)hipFree(t->d_packflag);
  for (auto& pk : t->packs) if (pk->base) (void)hipFree(pk->base);
  for (auto& nw : t->narrows) if (nw->base) (void)hipFree(nw->base);
  for (auto& pp : t->predpacks) for (char* b : pp->pbase) if (b) (void)hipFree(b);
  if (t->h_jobs) (void)hipHostFree(t->h_jobs);
  if (t->derived_ev) (void)hipEventDestroy(t->derived_ev);
  delete t;
}

// What a partitioning query reads while it appends tuples, and where in its scratch the tuple pool will lie: enough to try a scratch

// vhh_sync.h — host side of libviya_hip, part of viya_hip.hip's translation unit (included there, in order; not a stand-alone header):
// data in: vh_segment_sync*, the per-segment min / max pass (refresh_stats), vh_segment_generate, vh_segment_read.
static int sync_resolve(vh_table* t);      // (batched sync, below: what the last batch left to merge into the stats)
static hipError_t wait_event_spinning(hipEvent_t ev);      // (vhh_finalize.h: poll, then block — a blocking wait alone costs tens of microseconds of wake-up)
static int ensure_segrows(VhExec* x, size_t n) {
  if (n <= x->h_segrows_cap) return VH_OK;
  if (x->h_segrows) (void)hipHostFree(x->h_segrows);
  size_t cap = std::max<size_t>(n * 2, 1024);
  // coherent (fine-grained): init_regions_kernel reads the plan words straight out of this buffer, rewritten by the host before every query
  HIP_TRY(hipHostMalloc((void**)&x->h_segrows, cap * sizeof(uint32_t), hipHostMallocCoherent));
  x->h_segrows_cap = cap;
  return VH_OK;
}

#define VH_ELEM_SWITCH(elem, CALL)                       \
  switch (elem) {                                        \
    case VH_U8: { typedef uint8_t T; CALL; } break;      \
    case VH_U16: { typedef uint16_t T; CALL; } break;    \
    case VH_U32: { typedef uint32_t T; CALL; } break;    \
    case VH_U64: { typedef uint64_t T; CALL; } break;    \
    case VH_I8: { typedef int8_t T; CALL; } break;       \
    case VH_I16: { typedef int16_t T; CALL; } break;     \
    case VH_I32: { typedef int32_t T; CALL; } break;     \
    case VH_I64: { typedef int64_t T; CALL; } break;     \
    case VH_F32: { typedef float T; CALL; } break;       \
    default: { typedef double T; CALL; } break;          \
  }

// Refresh the per-segment min / max of every fixed-width column for segments [first, first+n). For NUMERIC / TIME dimensions these are the
// reference's SegmentStats (store.cc:171-201: segment skipping, dense digit ranges); for the other columns — metrics included, which the
// reference keeps no stats for — they tell the planner how many BITS the values really use: compressed records (vh_table_pack), narrow
// predicate copies and the packed tuples of the hashed partitioning (vh_hpart.h) are sized from them. One pass over the segment in HBM.
static int refresh_stats(vh_table* t, uint32_t first, uint32_t n) {
  int ndim = 0;
  for (auto& c : t->cols) ndim += !is_bitset_elem(c.elem);
  if (!ndim || !n) return VH_OK;
  const size_t stat_bytes = (size_t)ndim * n * 2 * sizeof(unsigned long long);
  const size_t rows_bytes = (size_t)n * sizeof(uint32_t);
  if (stat_bytes + rows_bytes + 256 > t->d_stats_bytes) {
    if (t->d_stats) { HIP_TRY(hipFree(t->d_stats)); t->d_stats = nullptr; t->d_stats_bytes = 0; }
    const size_t nb = std::max<size_t>((stat_bytes + rows_bytes + 256) * 2, 1 << 16);
    HIP_TRY(hipMalloc(&t->d_stats, nb));
    t->d_stats_bytes = nb;
  }
  unsigned long long* d_stats = reinterpret_cast<unsigned long long*>(t->d_stats);
  uint32_t* d_rows = reinterpret_cast<uint32_t*>(t->d_stats + stat_bytes);
  std::vector<unsigned long long> init((size_t)ndim * n * 2);
  for (size_t i = 0; i < init.size(); i += 2) { init[i] = ~0ull; init[i + 1] = 0; }
  std::vector<uint32_t> hrows(n);
  for (uint32_t s = 0; s < n; ++s) hrows[s] = (uint32_t)t->seg_rows[first + s];
  HIP_TRY(hipMemcpyAsync(d_stats, init.data(), stat_bytes, hipMemcpyHostToDevice, g_ctx.stream));
  HIP_TRY(hipMemcpyAsync(d_rows, hrows.data(), rows_bytes, hipMemcpyHostToDevice, g_ctx.stream));
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));  // init is a stack/heap buffer
  int di = 0;
  for (auto& c : t->cols) {
    if (is_bitset_elem(c.elem)) continue;
    dim3 grid((unsigned)std::min<uint64_t>(64, (t->segment_rows + 4095) / 4096), n);
    unsigned long long* st = d_stats + (size_t)di * n * 2;
    VH_ELEM_SWITCH(c.elem, (seg_minmax_kernel<T><<<grid, dim3(256), 0, g_ctx.stream>>>(
                               reinterpret_cast<const T*>(c.base), c.stride / c.esize, d_rows, first, st)));
    ++di;
  }
  HIP_TRY(hipGetLastError());
  std::vector<unsigned long long> host((size_t)ndim * n * 2);
  HIP_TRY(hipMemcpyAsync(host.data(), d_stats, stat_bytes, hipMemcpyDeviceToHost, g_ctx.stream));
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  di = 0;
  for (size_t ci = 0; ci < t->cols.size(); ++ci) {
    auto& c = t->cols[ci];
    if (is_bitset_elem(c.elem)) continue;
    for (uint32_t s = 0; s < n; ++s) {
      t->stats[ci][first + s].lo = host[((size_t)di * n + s) * 2];
      t->stats[ci][first + s].hi = host[((size_t)di * n + s) * 2 + 1];
    }
    ++di;
  }
  return VH_OK;
}

extern "C" int vh_segment_sync(vh_table* t, uint32_t seg, uint64_t nrows, const void* const* col_ptrs) {
  if (!t || !col_ptrs) return vh_fail(VH_E_INVALID, "vh_segment_sync: null argument");
  if (nrows > t->segment_rows) return vh_fail(VH_E_INVALID, "vh_segment_sync: nrows %llu > segment_rows", (unsigned long long)nrows);
  if (seg >= VH_MAX_SEGMENTS) return vh_fail(VH_E_INVALID, "vh_segment_sync: segment index %u out of range", seg);
  VH_ENTER();
  std::lock_guard<std::mutex> lk(t->mu);
  if (int src = sync_resolve(t)) return src;
  table_quiesce(t);
  int rc = table_grow(t, seg + 1);
  if (rc) return rc;
  for (size_t i = 0; i < t->cols.size(); ++i) {
    auto& c = t->cols[i];
    if (is_bitset_elem(c.elem) || !col_ptrs[i] || !nrows) continue;
    HIP_TRY(hipMemcpyAsync(c.base + (size_t)seg * c.stride, col_ptrs[i], (size_t)nrows * c.esize,
                           hipMemcpyDefault, g_ctx.stream));   // host or device source (unified addressing)
  }
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  const uint64_t was = t->seg_rows[seg];
  t->seg_rows[seg] = nrows;
  t->nseg = std::max(t->nseg, seg + 1);
  table_note_change(t, seg, 0, std::max(was, nrows));
  return refresh_stats(t, seg, 1);
}

// ------------------------------------------------------------------ registered host memory (vh_host_register)
struct VhHostReg { uint64_t bytes; char* dev; };
static std::map<uintptr_t, VhHostReg> g_hostreg;
static std::mutex g_hostreg_mu;

extern "C" int vh_host_register(const void* base, uint64_t bytes) {
  if (!base || !bytes) return vh_fail(VH_E_INVALID, "vh_host_register: null range");
  if (!g_ctx.inited) return vh_fail(VH_E_INVALID, "vh_init has not been called");
  VH_ENTER();
  std::lock_guard<std::mutex> lk(g_hostreg_mu);
  auto it = g_hostreg.find(reinterpret_cast<uintptr_t>(base));
  if (it != g_hostreg.end()) {
    if (it->second.bytes >= bytes) return VH_OK;
    (void)hipHostUnregister(const_cast<void*>(base));
    g_hostreg.erase(it);
  }
  hipError_t he = hipHostRegister(const_cast<void*>(base), (size_t)bytes, hipHostRegisterDefault);
  void* dev = nullptr;
  if (he == hipSuccess) { he = hipHostGetDevicePointer(&dev, const_cast<void*>(base), 0); if (he != hipSuccess) (void)hipHostUnregister(const_cast<void*>(base)); }
  if (he != hipSuccess) { (void)hipGetLastError(); return vh_fail(VH_E_DEVICE, "vh_host_register: %llu bytes at %p: %s", (unsigned long long)bytes, base, hipGetErrorString(he)); }
  g_hostreg[reinterpret_cast<uintptr_t>(base)] = VhHostReg{bytes, static_cast<char*>(dev)};
  return VH_OK;
}
extern "C" int vh_host_unregister(const void* base) {
  if (!base) return VH_OK;
  VH_ENTER();
  std::lock_guard<std::mutex> lk(g_hostreg_mu);
  auto it = g_hostreg.find(reinterpret_cast<uintptr_t>(base));
  if (it == g_hostreg.end()) return VH_OK;
  g_hostreg.erase(it);
  // (a batch that still reads the range was ordered on g_ctx.stream: wait for it before the pages are let go)
  (void)hipStreamSynchronize(g_ctx.stream);
  HIP_TRY(hipHostUnregister(const_cast<void*>(base)));
  return VH_OK;
}
// The device's address of [p, p + bytes) when that lies inside ONE registered range, else NULL.
static const char* hostreg_lookup(const void* p, uint64_t bytes, uintptr_t* lo, uintptr_t* hi, const char** dev) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  std::lock_guard<std::mutex> lk(g_hostreg_mu);
  auto it = g_hostreg.upper_bound(a);
  if (it == g_hostreg.begin()) return nullptr;
  --it;
  if (a + bytes > it->first + it->second.bytes) return nullptr;
  *lo = it->first; *hi = it->first + it->second.bytes; *dev = it->second.dev;
  return it->second.dev + (a - it->first);
}

// ------------------------------------------------------------------ batched sync
// What the last vh_table_sync_batch left to do on the host: wait for its kernel, widen the per-segment stats by the runs' min / max.
// Called with t->mu held by everything that reads stats or arenas from the host side, plans a query, or starts another batch.
static int sync_resolve(vh_table* t) {
  if (!t->sync_inflight) return VH_OK;
  HIP_TRY(wait_event_spinning(t->sync_ev));
  t->sync_inflight = false;
  const unsigned long long* slots = reinterpret_cast<const unsigned long long*>(t->h_sync);
  for (const auto& p : t->sync_pending) {
    VhSegStat& st = t->stats[p.col][p.seg];
    for (uint32_t d = p.desc_first; d < p.desc_first + p.desc_n; ++d) {
      const uint64_t lo = slots[2ull * d], hi = slots[2ull * d + 1];
      if (lo > hi) continue;                    // (an empty run)
      st.lo = std::min(st.lo, lo);
      st.hi = std::max(st.hi, hi);
    }
  }
  t->sync_pending.clear();
  return VH_OK;
}

static const size_t VH_SYNC_STAGE_RUN = 64u << 10;       // runs of unregistered memory up to this size go through the pinned ring ...
static const size_t VH_SYNC_STAGE_BYTES = 16u << 20;     // ... while it has room; the rest is copied by the DMA engine

// What a batch has done so far, for the failure guard of sync_batch_locked.
struct VhSyncProgress {
  uint32_t launched = 0;                                   // descriptors handed to sync_pull_kernel
  bool queued = false;                                     // anything at all enqueued on the stream (kernels or DMA copies)
  std::vector<std::pair<uint32_t, uint64_t>> rows_before;  // (segment, seg_rows) as they stood when the batch first changed them
  uint32_t nseg_before = 0;
};

static int sync_batch_body(vh_table* t, const vh_sync_item* items, uint32_t n, VhSyncProgress& prog) {
  // ---- validate against the rows the mirror will hold as the items are applied in order
  uint32_t max_seg = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const vh_sync_item& it = items[i];
    if (it.seg >= VH_MAX_SEGMENTS) return vh_fail(VH_E_INVALID, "vh_table_sync_batch: item %u: segment index %u out of range", i, it.seg);
    if (it.new_size > t->segment_rows || it.row_first + it.nrows > it.new_size)
      return vh_fail(VH_E_INVALID, "vh_table_sync_batch: item %u: rows [%llu, %llu) do not fit a segment of %llu rows", i,
                     (unsigned long long)it.row_first, (unsigned long long)(it.row_first + it.nrows), (unsigned long long)it.new_size);
    if (it.nrows && !it.col_ptrs) return vh_fail(VH_E_INVALID, "vh_table_sync_batch: item %u: null col_ptrs", i);
    max_seg = std::max(max_seg, it.seg);
  }
  if (max_seg + 1 > t->cap_seg) {                          // arenas move: nothing may still read the old ones
    table_quiesce(t);
    if (int rc = table_grow(t, max_seg + 1)) return rc;
  }
  // the columns an item ships: every fixed-width one, or the metrics alone (lists made once per table: a batch over a thousand segments walks
  // them a thousand times, and the host's share of such a batch is what separates "resident" from the link's own time)
  if (t->ship_all.empty() && !t->cols.empty()) {
    for (size_t c = 0; c < t->cols.size(); ++c) {
      if (is_bitset_elem(t->cols[c].elem)) continue;
      t->ship_all.push_back((uint16_t)c);
      if (!is_dim(t->cols[c].kind)) t->ship_metrics.push_back((uint16_t)c);
    }
  }
  std::vector<uint64_t>& rows_now = t->sync_rows_now;      // segments named by this batch -> rows mirrored so far (UINT64_MAX: not named yet)
  rows_now.assign((size_t)max_seg + 1, ~0ull);
  // ---- one descriptor per run of at most `run_max` bytes: at most ~64 K of them however big the batch (counted for 256 KB runs: an upper bound)
  uint64_t total_bytes = 0, ndesc_max = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const vh_sync_item& it = items[i];
    uint64_t& have = rows_now[it.seg];
    if (have == ~0ull) have = t->seg_rows[it.seg];
    if (it.row_first > have)
      return vh_fail(VH_E_INVALID, "vh_table_sync_batch: item %u: segment %u has %llu mirrored rows, range starts at %llu (gap)", i, it.seg,
                     (unsigned long long)have, (unsigned long long)it.row_first);
    have = it.new_size;
    if (!it.nrows) continue;
    for (const uint16_t c : (it.flags & VH_SYNC_METRICS_ONLY) ? t->ship_metrics : t->ship_all) {
      if (!it.col_ptrs[c]) continue;
      const uint64_t bytes = it.nrows * (uint64_t)t->cols[c].esize;
      total_bytes += bytes;
      ndesc_max += (bytes + (256u << 10) - 1) >> 18;
    }
  }
  size_t run_max = 256u << 10;
  while (total_bytes / run_max > (1u << 16) && run_max < (4u << 20)) run_max *= 2;
  if (ndesc_max > 0x7FFFFFFFull) return vh_fail(VH_E_INVALID, "vh_table_sync_batch: too many runs");
  if (!t->sync_ev) HIP_TRY(hipEventCreateWithFlags(&t->sync_ev, hipEventDisableTiming));
  const size_t need = (size_t)ndesc_max * (2 * sizeof(unsigned long long) + sizeof(VhSyncDesc));
  if (need > t->h_sync_bytes) {
    if (t->h_sync) { HIP_TRY(hipHostFree(t->h_sync)); t->h_sync = nullptr; t->h_sync_bytes = 0; }
    const size_t nb = std::max<size_t>(need * 2, 1u << 16);
    HIP_TRY(hipHostMalloc((void**)&t->h_sync, nb, hipHostMallocCoherent));
    t->h_sync_bytes = nb;
  }
  unsigned long long* slots = reinterpret_cast<unsigned long long*>(t->h_sync);
  VhSyncDesc* descs = reinterpret_cast<VhSyncDesc*>(t->h_sync + (size_t)ndesc_max * 2 * sizeof(unsigned long long));
  uint32_t nd = 0;
  uint32_t& launched = prog.launched;
  // (the kernel of the first few thousand runs pulls while the host is still writing the descriptors of the next)
  uint32_t launch_at = 1024;                               // first launch early, then twice as many runs each time: a handful of launches however big the batch
  auto launch_some = [&](bool all) -> int {
    if (nd == launched || (!all && nd - launched < launch_at)) return VH_OK;
    launch_at *= 2;
    hipLaunchKernelGGL(sync_pull_kernel, dim3(nd - launched), dim3(256), 0, g_ctx.stream, descs + launched, slots + 2ull * launched);
    HIP_TRY(hipGetLastError());
    launched = nd; prog.queued = true;
    return VH_OK;
  };
  size_t stage_used = 0;
  bool dma_from_pageable = false;
  uintptr_t reg_lo = 0, reg_hi = 0; const char* reg_dev = nullptr;     // the registered range the last run lay in (a segment's columns share one)
  t->sync_pending.reserve((size_t)ndesc_max);
  for (uint32_t i = 0; i < n; ++i) {
    const vh_sync_item& it = items[i];
    // a range that holds every row the segment will have replaces the stats of the columns it ships; any other range widens them
    const bool replace = it.row_first == 0 && it.nrows == it.new_size;
    for (const uint16_t c : (it.flags & VH_SYNC_METRICS_ONLY) ? t->ship_metrics : t->ship_all) {
      const VhColumn& col = t->cols[c];
      if (replace && (it.col_ptrs && it.col_ptrs[c])) t->stats[c][it.seg] = VhSegStat();
      if (!it.nrows || !it.col_ptrs[c]) continue;
      const uint64_t bytes = it.nrows * (uint64_t)col.esize;
      const char* host = static_cast<const char*>(it.col_ptrs[c]) + it.row_first * (uint64_t)col.esize;
      char* dst = col.base + (size_t)it.seg * col.stride + it.row_first * (uint64_t)col.esize;
      const char* src = nullptr;
      uint8_t copy = 1;
      if (it.flags & VH_SYNC_DEVICE_SRC) src = host;
      else if (reinterpret_cast<uintptr_t>(host) >= reg_lo && reinterpret_cast<uintptr_t>(host) + bytes <= reg_hi) { src = reg_dev + (reinterpret_cast<uintptr_t>(host) - reg_lo); t->sync_bytes_pulled += bytes; }
      else if (const char* dev = hostreg_lookup(host, bytes, &reg_lo, &reg_hi, &reg_dev)) { src = dev; t->sync_bytes_pulled += bytes; }
      else if (bytes <= VH_SYNC_STAGE_RUN && stage_used + bytes + 16 <= VH_SYNC_STAGE_BYTES) {
        if (!t->h_stage) { HIP_TRY(hipHostMalloc((void**)&t->h_stage, VH_SYNC_STAGE_BYTES, hipHostMallocDefault)); t->h_stage_bytes = VH_SYNC_STAGE_BYTES; }
        char* at = t->h_stage + stage_used;
        memcpy(at, host, bytes);
        stage_used += (bytes + 15) / 16 * 16;
        src = at;
        t->sync_bytes_staged += bytes;
      } else {
        prog.queued = true;
        HIP_TRY(hipMemcpyAsync(dst, host, bytes, hipMemcpyHostToDevice, g_ctx.stream));
        dma_from_pageable = true;
        src = dst; copy = 0;
        t->sync_bytes_dma += bytes;
      }
      const uint32_t first = nd;
      for (uint64_t off = 0; off < bytes; off += run_max) {
        const uint64_t b = std::min<uint64_t>(run_max, bytes - off);
        VhSyncDesc& d = descs[nd++];
        d.src = src + off; d.dst = dst + off; d.nelem = (uint32_t)(b / col.esize); d.elem = (uint8_t)col.elem; d.copy = copy; d.pad0 = 0; d.pad1 = d.pad2 = 0;
      }
      t->sync_pending.push_back(vh_table::SyncPending{(uint32_t)c, it.seg, first, nd - first});
    }
    if (int lrc = launch_some(false)) return lrc;
    if (const char* e = test_env("VH_TEST_SYNC_FAIL_AT")) if ((uint32_t)atoi(e) == i) { (void)launch_some(true); return vh_fail(VH_E_DEVICE, "vh_table_sync_batch: failure injected at item %u", i); }
    const uint64_t was = t->seg_rows[it.seg];
    prog.rows_before.emplace_back(it.seg, was);
    t->seg_rows[it.seg] = it.new_size;
    t->nseg = std::max(t->nseg, it.seg + 1);
    // (rows that fell off the end of a shrunk segment count as changed: layouts zero what lies beyond size())
    table_note_change(t, it.seg, it.new_size < was ? std::min<uint64_t>(it.row_first, it.new_size) : it.row_first, it.new_size < was ? was : it.row_first + it.nrows);
  }
  if (nd) {
    if (int lrc = launch_some(true)) return lrc;
    HIP_TRY(hipEventRecord(t->sync_ev, g_ctx.stream));
    t->sync_inflight = true;
    ++t->sync_batches; t->sync_descs += nd;
  }
  // sources in unregistered memory behind a DMA copy: the call promises they may be reused when it returns
  if (dma_from_pageable) HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  return VH_OK;
}

// A batch either goes through or leaves the table as it found it. A failure half way — an allocation, a launch, a copy — has
// by then advanced seg_rows and the journal, maybe reset a column's stats for a whole-segment range, maybe launched some of its runs, and
// holds sync_pending entries that the NEXT batch would pair with ITS slots. The guard: wait for whatever was enqueued (its sources and the
// descriptors must not be reused under it), merge the min / max of the runs that did execute, widen to "anything" the stats of every column
// whose runs did not all execute (stats too narrow would skip segments and size bit fields wrongly; too wide only costs time), put seg_rows
// back, and leave nothing pending. The caller's next batch re-sends the same ranges; rows already pulled are simply pulled again.
static int sync_batch_locked(vh_table* t, const vh_sync_item* items, uint32_t n) {
  if (int rc = sync_resolve(t)) return rc;                // the previous batch's slots, descriptors and ring are free again
  VhSyncProgress prog;
  prog.nseg_before = t->nseg;
  const int rc = sync_batch_body(t, items, n, prog);
  if (rc == VH_OK) return VH_OK;
  char own[sizeof(g_err)];
  snprintf(own, sizeof(own), "%s", g_err);                  // (the waits below may overwrite the message)
  if (prog.queued) (void)hipStreamSynchronize(g_ctx.stream);
  const unsigned long long* slots = reinterpret_cast<const unsigned long long*>(t->h_sync);
  for (const auto& p : t->sync_pending) {
    VhSegStat& st = t->stats[p.col][p.seg];
    if (slots && p.desc_first + p.desc_n <= prog.launched) {
      for (uint32_t d = p.desc_first; d < p.desc_first + p.desc_n; ++d) {
        const uint64_t lo = slots[2ull * d], hi = slots[2ull * d + 1];
        if (lo <= hi) { st.lo = std::min(st.lo, lo); st.hi = std::max(st.hi, hi); }
      }
    } else { st.lo = 0; st.hi = ~0ull; }
  }
  t->sync_pending.clear();
  t->sync_inflight = false;
  for (auto it = prog.rows_before.rbegin(); it != prog.rows_before.rend(); ++it) t->seg_rows[it->first] = it->second;   // (oldest value of a segment wins)
  t->nseg = prog.nseg_before;
  return vh_fail(rc, "%s", own);
}

extern "C" int vh_table_sync_batch(vh_table* t, const vh_sync_item* items, uint32_t nitems) {
  if (!t || (!items && nitems)) return vh_fail(VH_E_INVALID, "vh_table_sync_batch: null argument");
  if (!nitems) return VH_OK;
  VH_ENTER();
  std::lock_guard<std::mutex> lk(t->mu);
  return sync_batch_locked(t, items, nitems);
}

extern "C" int vh_table_sync_stats(vh_table* t, uint64_t* batches, uint64_t* runs, uint64_t* bytes_pulled, uint64_t* bytes_staged, uint64_t* bytes_dma) {
  if (!t) return vh_fail(VH_E_INVALID, "null table");
  std::lock_guard<std::mutex> lk(t->mu);
  if (batches) *batches = t->sync_batches;
  if (runs) *runs = t->sync_descs;
  if (bytes_pulled) *bytes_pulled = t->sync_bytes_pulled;
  if (bytes_staged) *bytes_staged = t->sync_bytes_staged;
  if (bytes_dma) *bytes_dma = t->sync_bytes_dma;
  return VH_OK;
}

// The one-range form: a batch of one item.
extern "C" int vh_segment_sync_range(vh_table* t, uint32_t seg, uint64_t row_first, uint64_t nrows, uint64_t new_size,
                                     const void* const* col_ptrs) {
  if (!t || !col_ptrs) return vh_fail(VH_E_INVALID, "vh_segment_sync_range: null argument");
  const vh_sync_item it{seg, 0u, row_first, nrows, new_size, col_ptrs};
  VH_ENTER();
  std::lock_guard<std::mutex> lk(t->mu);
  return sync_batch_locked(t, &it, 1);
}

// The 32-bit copy of a segment's CSR offsets (VhColumn::bs_offsets32): rebuilt whenever the 64-bit ones are.
static int bitset_offsets32(VhColumn& c, uint32_t seg, uint64_t nrows, uint64_t nvals) {
  if (c.bs_offsets32[seg]) { HIP_TRY(hipFree(c.bs_offsets32[seg])); c.bs_offsets32[seg] = nullptr; }
  if (nvals > 0xFFFFFFFFull || !c.bs_offsets[seg]) return VH_OK;
  HIP_TRY(hipMalloc((void**)&c.bs_offsets32[seg], (nrows + 4) * sizeof(uint32_t)));      // (+ what an 8-byte load at the last row reads)
  hipLaunchKernelGGL(narrow_offsets_kernel, dim3((unsigned)std::min<uint64_t>((nrows + 256) / 256, 4096)), dim3(256), 0, g_ctx.stream, c.bs_offsets[seg], c.bs_offsets32[seg], nrows + 1);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  return VH_OK;
}

extern "C" int vh_segment_sync_bitset(vh_table* t, uint32_t seg, int32_t col, uint64_t nrows,
                                      const uint64_t* offsets, const void* values) {
  if (!t || col < 0 || (size_t)col >= t->cols.size() || !offsets) return vh_fail(VH_E_INVALID, "vh_segment_sync_bitset: bad argument");
  auto& c = t->cols[col];
  if (!is_bitset_elem(c.elem)) return vh_fail(VH_E_INVALID, "column %d is not a bitset column", col);
  if (nrows > t->segment_rows || seg >= VH_MAX_SEGMENTS) return vh_fail(VH_E_INVALID, "vh_segment_sync_bitset: segment %u / %llu rows out of range", seg, (unsigned long long)nrows);
  if (offsets[0] != 0 || (offsets[nrows] && !values)) return vh_fail(VH_E_INVALID, "vh_segment_sync_bitset: offsets must start at 0 and values must be given");
  VH_ENTER();
  std::lock_guard<std::mutex> lk(t->mu);
  table_quiesce(t);
  int rc = table_grow(t, seg + 1);
  if (rc) return rc;
  if (c.bs_offsets[seg]) { HIP_TRY(hipFree(c.bs_offsets[seg])); c.bs_offsets[seg] = nullptr; }
  if (c.bs_values[seg]) { HIP_TRY(hipFree(c.bs_values[seg])); c.bs_values[seg] = nullptr; }
  const uint64_t nvals = offsets[nrows];
  const size_t vsz = c.elem == VH_BITSET32 ? 4 : 8;
  HIP_TRY(hipMalloc((void**)&c.bs_offsets[seg], (nrows + 1) * sizeof(uint64_t)));
  HIP_TRY(hipMalloc((void**)&c.bs_values[seg], nvals * vsz + VH_BS_PAD));
  HIP_TRY(hipMemcpy(c.bs_offsets[seg], offsets, (nrows + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
  if (nvals) HIP_TRY(hipMemcpy(c.bs_values[seg], values, nvals * vsz, hipMemcpyHostToDevice));
  c.bs_nvalues[seg] = nvals;
  if (int orc = bitset_offsets32(c, seg, nrows, nvals)) return orc;
  {
    uint64_t mx = 0;
    if (c.elem == VH_BITSET32) { const uint32_t* v = static_cast<const uint32_t*>(values); for (uint64_t i = 0; i < nvals; ++i) mx = std::max<uint64_t>(mx, v[i]); }
    else { const uint64_t* v = static_cast<const uint64_t*>(values); for (uint64_t i = 0; i < nvals; ++i) mx = std::max(mx, v[i]); }
    c.bs_maxid[seg] = mx;
  }
  t->nseg = std::max(t->nseg, seg + 1);
  return VH_OK;
}

// One id per row, ids already in HBM (exchanged (group, id) pairs on their owner): offsets are 0, 1, 2, ... n.
extern "C" int vh_segment_sync_ids_device(vh_table* t, uint32_t seg, int32_t col, uint64_t nrows, const void* d_ids) {
  if (!t || col < 0 || (size_t)col >= t->cols.size()) return vh_fail(VH_E_INVALID, "vh_segment_sync_ids_device: bad argument");
  auto& c = t->cols[col];
  if (!is_bitset_elem(c.elem)) return vh_fail(VH_E_INVALID, "column %d is not a bitset column", col);
  if (nrows > t->segment_rows || seg >= VH_MAX_SEGMENTS) return vh_fail(VH_E_INVALID, "vh_segment_sync_ids_device: segment %u / %llu rows out of range", seg, (unsigned long long)nrows);
  if (nrows && !d_ids) return vh_fail(VH_E_INVALID, "vh_segment_sync_ids_device: null ids");
  VH_ENTER();
  std::lock_guard<std::mutex> lk(t->mu);
  table_quiesce(t);
  int rc = table_grow(t, seg + 1);
  if (rc) return rc;
  if (c.bs_offsets[seg]) { HIP_TRY(hipFree(c.bs_offsets[seg])); c.bs_offsets[seg] = nullptr; }
  if (c.bs_values[seg]) { HIP_TRY(hipFree(c.bs_values[seg])); c.bs_values[seg] = nullptr; }
  const size_t vsz = c.elem == VH_BITSET32 ? 4 : 8;
  HIP_TRY(hipMalloc((void**)&c.bs_offsets[seg], (nrows + 1) * sizeof(uint64_t)));
  HIP_TRY(hipMalloc((void**)&c.bs_values[seg], nrows * vsz + VH_BS_PAD));
  hipLaunchKernelGGL(iota_kernel, dim3((unsigned)std::min<uint64_t>((nrows + 256) / 256, 65535)), dim3(256), 0, g_ctx.stream,
                     c.bs_offsets[seg], nrows + 1);
  HIP_TRY(hipGetLastError());
  if (nrows) HIP_TRY(hipMemcpyAsync(c.bs_values[seg], d_ids, nrows * vsz, hipMemcpyDefault, g_ctx.stream));
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  c.bs_nvalues[seg] = nrows;
  if (int orc = bitset_offsets32(c, seg, nrows, nrows)) return orc;
  c.bs_maxid[seg] = c.elem == VH_BITSET32 ? 0xFFFFFFFFull : ~0ull;      // (exchanged ids, never looked at on this side: the type's range)
  t->nseg = std::max(t->nseg, seg + 1);
  return VH_OK;
}

extern "C" int vh_segment_generate(vh_table* t, uint32_t seg_first, uint32_t nseg, uint64_t rows_per_seg,
                                   uint64_t row_base, const vh_gen_spec* specs, uint64_t seed) {
  if (!t || !specs || !nseg) return vh_fail(VH_E_INVALID, "vh_segment_generate: bad argument");
  if (rows_per_seg > t->segment_rows) return vh_fail(VH_E_INVALID, "rows_per_seg exceeds segment_rows");
  if (seg_first >= VH_MAX_SEGMENTS || nseg > VH_MAX_SEGMENTS - seg_first) return vh_fail(VH_E_INVALID, "vh_segment_generate: segments [%u, +%u) out of range", seg_first, nseg);
  VH_ENTER();
  std::lock_guard<std::mutex> lk(t->mu);
  if (int src = sync_resolve(t)) return src;
  table_quiesce(t);
  int rc = table_grow(t, seg_first + nseg);
  if (rc) return rc;
  for (size_t i = 0; i < t->cols.size(); ++i) {
    auto& c = t->cols[i];
    if (specs[i].mode != VH_GEN_ROWID && specs[i].mode != VH_GEN_CONST && specs[i].mod == 0) return vh_fail(VH_E_INVALID, "column %zu: mod == 0", i);
    const uint64_t colseed = seed ^ ((uint64_t)i * 0x9E3779B97F4A7C15ull);
    if (is_bitset_elem(c.elem)) {   // CSR per segment: `add` ids per row drawn from [0, mod)
      const uint32_t k = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(specs[i].add, 8));
      const size_t vsz = c.elem == VH_BITSET32 ? 4 : 8;
      for (uint32_t sgi = 0; sgi < nseg; ++sgi) {
        const uint32_t seg = seg_first + sgi;
        if (c.bs_offsets[seg]) { HIP_TRY(hipFree(c.bs_offsets[seg])); c.bs_offsets[seg] = nullptr; }
        if (c.bs_values[seg]) { HIP_TRY(hipFree(c.bs_values[seg])); c.bs_values[seg] = nullptr; }
        HIP_TRY(hipMalloc((void**)&c.bs_offsets[seg], (rows_per_seg + 1) * sizeof(uint64_t)));
        HIP_TRY(hipMalloc((void**)&c.bs_values[seg], rows_per_seg * k * vsz + VH_BS_PAD));
        const unsigned grid = (unsigned)std::min<uint64_t>(512, (rows_per_seg + 256) / 256);
        const uint64_t rb = row_base + (uint64_t)sgi * rows_per_seg;
        if (vsz == 4) gen_csr_kernel<uint32_t><<<grid, 256, 0, g_ctx.stream>>>(c.bs_offsets[seg], (uint32_t*)c.bs_values[seg], rows_per_seg, k, rb, specs[i].mod, colseed);
        else gen_csr_kernel<uint64_t><<<grid, 256, 0, g_ctx.stream>>>(c.bs_offsets[seg], (uint64_t*)c.bs_values[seg], rows_per_seg, k, rb, specs[i].mod, colseed);
        c.bs_nvalues[seg] = rows_per_seg * k;
        if (int orc = bitset_offsets32(c, seg, rows_per_seg, rows_per_seg * k)) return orc;
        t->device_bytes += (rows_per_seg + 4) * 4;
        c.bs_maxid[seg] = specs[i].mod - 1;      // (ids are drawn from [0, mod))
        t->device_bytes += (rows_per_seg + 1) * 8 + rows_per_seg * k * vsz;
      }
      continue;
    }
    dim3 grid((unsigned)std::min<uint64_t>(256, (rows_per_seg + 255) / 256), nseg);
    VH_ELEM_SWITCH(c.elem, (gen_kernel<T><<<grid, dim3(256), 0, g_ctx.stream>>>(
                               reinterpret_cast<T*>(c.base + (size_t)seg_first * c.stride), c.stride / c.esize,
                               rows_per_seg, row_base, specs[i], colseed, seed)));
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  ++t->sync_epoch;
  for (uint32_t s = 0; s < nseg; ++s) { const uint64_t was = t->seg_rows[seg_first + s]; t->seg_rows[seg_first + s] = rows_per_seg; table_note_change(t, seg_first + s, 0, std::max(was, rows_per_seg), false); }
  t->nseg = std::max(t->nseg, seg_first + nseg);
  // stats in batches so the staging buffers stay small
  for (uint32_t s = 0; s < nseg; s += 256) {
    rc = refresh_stats(t, seg_first + s, std::min<uint32_t>(256, nseg - s));
    if (rc) return rc;
  }
  return VH_OK;
}

extern "C" int vh_segment_read(vh_table* t, uint32_t seg, int32_t col, uint64_t nrows, void* dst) {
  if (!t || col < 0 || (size_t)col >= t->cols.size() || seg >= t->nseg || !dst) return vh_fail(VH_E_INVALID, "vh_segment_read: bad argument");
  auto& c = t->cols[col];
  if (is_bitset_elem(c.elem)) return vh_fail(VH_E_UNSUPPORTED, "vh_segment_read: bitset column");
  VH_ENTER();
  std::lock_guard<std::mutex> lk(t->mu);
  if (int src = sync_resolve(t)) return src;
  HIP_TRY(hipMemcpy(dst, c.base + (size_t)seg * c.stride, (size_t)nrows * c.esize, hipMemcpyDeviceToHost));
  return VH_OK;
}

// Host copy of a vh_device_buffer (the exchange buffers of vh_result_partition[_pairs]); ordered after the library's stream.
extern "C" int vh_device_read(void* dst, const void* device_src, uint64_t bytes) {
  if (!bytes) return VH_OK;
  if (!dst || !device_src) return vh_fail(VH_E_INVALID, "vh_device_read: null argument");
  VH_ENTER();
  HIP_TRY(hipMemcpy(dst, device_src, (size_t)bytes, hipMemcpyDeviceToHost));   // the buffers were produced by calls that completed on their own stream
  return VH_OK;
}

extern "C" int vh_table_info(vh_table* t, uint32_t* nseg, uint64_t* segment_rows, uint64_t* device_bytes) {
  if (!t) return vh_fail(VH_E_INVALID, "null table");
  if (nseg) *nseg = t->nseg;
  if (segment_rows) *segment_rows = t->segment_rows;
  if (device_bytes) *device_bytes = t->device_bytes;
  return VH_OK;
}


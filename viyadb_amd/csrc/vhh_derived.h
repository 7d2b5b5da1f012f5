// vhh_derived.h — host side of libviya_hip, part of viya_hip.hip's translation unit (included there, in order; not a stand-alone header):
// derived layouts: payload projections (vh_table_pack) and narrow predicate copies (vh_table_narrow).
// ------------------------------------------------------- payload projections (vh_table_pack)
static uint64_t order_key_of_bits(int elem, uint64_t bits);      // (typed host helpers: vhh_result.h)
static uint64_t bits_of_order_key(int elem, uint64_t k);
#define VH_PACK_STALE 9001      // (internal) pack_refresh: a value no longer fits its stored width, the projection must go
// What a derived layout that was current at `applied_epoch` (per segment: at `seg_mod[s]`) has to re-derive, as jobs for its kernel: the
// row ranges journalled since (vh_table::journal), cut at 256-row boundaries, merged, and split into pieces of VH_JOB_ROWS; whole segments
// for a layout that is new, or so far behind that the journal no longer reaches back to it. `row_limit`: rows a segment of the layout has
// room for (a projection's stride is padded to 256 rows, a narrow copy's to 64).
static void derived_jobs(const vh_table* t, uint64_t applied_epoch, const std::vector<uint64_t>& seg_mod, uint64_t row_limit, std::vector<VhJob>* jobs) {
  jobs->clear();
  std::vector<std::pair<uint64_t, uint64_t>> ranges;      // (seg << 32 | first, last)
  auto whole = [&](uint32_t s) { if (s < seg_mod.size() && seg_mod[s] != t->seg_mod[s]) ranges.emplace_back((uint64_t)s << 32, row_limit); };
  if (applied_epoch == 0 || applied_epoch < t->journal_floor) {
    for (uint32_t s = 0; s < t->nseg; ++s) whole(s);
  } else {
    auto it = std::upper_bound(t->journal.begin(), t->journal.end(), applied_epoch, [](uint64_t e, const VhChange& c) { return e < c.epoch; });
    for (; it != t->journal.end(); ++it) {
      if (it->seg >= t->nseg || it->seg >= seg_mod.size()) continue;
      if (seg_mod[it->seg] == 0) { whole(it->seg); continue; }          // a segment this layout never held (the table grew)
      const uint64_t a = it->first & ~255ull, b = std::min<uint64_t>(((uint64_t)it->last + 255) & ~255ull, row_limit);
      if (a < b) ranges.emplace_back(((uint64_t)it->seg << 32) | a, b);
    }
  }
  if (ranges.empty()) return;
  std::sort(ranges.begin(), ranges.end());
  size_t o = 0;
  for (size_t i = 1; i < ranges.size(); ++i) {
    if ((ranges[i].first >> 32) == (ranges[o].first >> 32) && (ranges[i].first & 0xFFFFFFFFull) <= ranges[o].second) ranges[o].second = std::max(ranges[o].second, ranges[i].second);
    else ranges[++o] = ranges[i];
  }
  ranges.resize(o + 1);
  for (const auto& r : ranges) {
    const uint32_t seg = (uint32_t)(r.first >> 32);
    const uint64_t a = r.first & 0xFFFFFFFFull, b = std::min(r.second, row_limit);
    for (uint64_t f = a; f < b; f += VH_JOB_ROWS) jobs->push_back(VhJob{seg, (uint32_t)f, (uint32_t)std::min<uint64_t>(VH_JOB_ROWS, b - f), (uint32_t)t->seg_rows[seg]});
  }
}
// The jobs in pinned memory the kernels read them from (they are 16 bytes each; a list lives until the stream has been waited for).
static int derived_upload(vh_table* t, const std::vector<VhJob>& jobs, const VhJob** out) {
  const size_t bytes = jobs.size() * sizeof(VhJob);
  if (t->h_jobs_used + bytes > t->h_jobs_bytes) {
    if (t->derived_pending) { HIP_TRY(hipStreamSynchronize(g_ctx.stream)); t->derived_pending = false; }      // (earlier lists are still being read)
    t->h_jobs_used = 0;
    if (bytes > t->h_jobs_bytes) {
      if (t->h_jobs) { HIP_TRY(hipHostFree(t->h_jobs)); t->h_jobs = nullptr; t->h_jobs_bytes = 0; }
      const size_t nb = std::max<size_t>(bytes * 2, 1u << 16);
      HIP_TRY(hipHostMalloc((void**)&t->h_jobs, nb, hipHostMallocCoherent));
      t->h_jobs_bytes = nb;
    }
  }
  memcpy(t->h_jobs + t->h_jobs_used, jobs.data(), bytes);
  *out = reinterpret_cast<const VhJob*>(t->h_jobs + t->h_jobs_used);
  t->h_jobs_used += (bytes + 255) / 256 * 256;
  return VH_OK;
}
static int derived_enqueued(vh_table* t) {       // a refresh kernel went onto g_ctx.stream: queries launched from now on wait for it (QueryBuild::launch)
  if (!t->derived_ev) HIP_TRY(hipEventCreateWithFlags(&t->derived_ev, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(t->derived_ev, g_ctx.stream));
  t->derived_pending = true;
  return VH_OK;
}
static int derived_waited(vh_table* t);
// Work about to go onto a query's own stream reads derived layouts (and arenas): it is ordered behind the refreshes enqueued on g_ctx.stream.
static int derived_fence(vh_table* t, hipStream_t st) {
  if (!t->derived_pending) return VH_OK;
  if (hipEventQuery(t->derived_ev) == hipSuccess) return derived_waited(t);
  HIP_TRY(hipStreamWaitEvent(st, t->derived_ev, 0));
  return VH_OK;
}
static int derived_waited(vh_table* t) {         // the host waited for g_ctx.stream: nothing pending, the job lists are free
  t->derived_pending = false; t->h_jobs_used = 0;
  return VH_OK;
}

// Bring the projection up to date with the arenas: re-pack what changed since it was last packed (derived_jobs) in ONE launch.
static int pack_refresh(vh_table* t, VhPack* pk, uint32_t first, uint32_t n) {
  (void)first; (void)n;
  if (!t->nseg) return VH_OK;
  if (pk->cap_seg < t->cap_seg) {                      // the table grew: move the arena
    table_quiesce(t);
    HIP_TRY(hipStreamSynchronize(g_ctx.stream)); derived_waited(t);
    char* nb = nullptr;
    const size_t bytes = (size_t)t->cap_seg * pk->stride + 256;
    HIP_TRY(hipMalloc(&nb, bytes));
    trace_alloc("projection", nb, bytes);
    if (pk->base) {
      HIP_TRY(hipMemcpyAsync(nb, pk->base, (size_t)pk->cap_seg * pk->stride, hipMemcpyDeviceToDevice, g_ctx.stream));
      HIP_TRY(hipStreamSynchronize(g_ctx.stream));
      HIP_TRY(hipFree(pk->base));
      t->device_bytes -= (size_t)pk->cap_seg * pk->stride + 256;
    }
    pk->base = nb; pk->cap_seg = t->cap_seg;
    pk->seg_mod.resize(t->cap_seg, 0);
    t->device_bytes += bytes;
  }
  if (pk->applied_epoch == t->sync_epoch) return VH_OK;
  std::vector<VhJob> jobs;
  derived_jobs(t, pk->applied_epoch, pk->seg_mod, (t->segment_rows + 255) / 256 * 256, &jobs);
  if (!jobs.empty()) {
    const VhJob* d_jobs = nullptr;
    if (int rc = derived_upload(t, jobs, &d_jobs)) return rc;
    if (!t->d_packflag) { HIP_TRY(hipMalloc((void**)&t->d_packflag, 256)); HIP_TRY(hipMemsetAsync(t->d_packflag, 0, 256, g_ctx.stream)); }
    if (pk->bits) {
      VhPackBitsArgs B{};
      B.ncols = (int32_t)pk->cols.size(); B.rec_bytes = pk->rec_bytes;
      for (size_t c = 0; c < pk->cols.size(); ++c) {
        const VhColumn& col = t->cols[pk->cols[c]];
        B.src[c] = col.base; B.src_stride[c] = col.stride; B.esize[c] = (uint32_t)col.esize; B.bitoff[c] = pk->bitoff[c]; B.bitw[c] = pk->bitw[c];
      }
      B.overflow = t->d_packflag; B.dst = pk->base; B.dst_stride = pk->stride; B.jobs = d_jobs;
      hipLaunchKernelGGL(pack_bits_kernel, dim3((unsigned)jobs.size()), dim3(256), 0, g_ctx.stream, B);
    } else {
      VhPackArgs A{};
      A.ncols = (int32_t)pk->cols.size(); A.rec_bytes = pk->rec_bytes;
      for (size_t c = 0; c < pk->cols.size(); ++c) {
        const VhColumn& col = t->cols[pk->cols[c]];
        A.src[c] = col.base; A.src_stride[c] = col.stride; A.esize[c] = (uint32_t)col.esize; A.off[c] = pk->off[c];
        A.wbytes[c] = pk->width[c];
        if (col.elem == VH_I8 || col.elem == VH_I16 || col.elem == VH_I32 || col.elem == VH_I64) A.sgn_mask |= 1u << c;
      }
      A.overflow = t->d_packflag; A.dst = pk->base; A.dst_stride = pk->stride; A.jobs = d_jobs;
      hipLaunchKernelGGL(pack_kernel, dim3((unsigned)jobs.size()), dim3(256), 256 * pk->rec_bytes, g_ctx.stream, A);
    }
    HIP_TRY(hipGetLastError());
    if (pk->compressed) {                               // did every value survive its stored width? (the one host wait of a refresh)
      unsigned int ovf = 0;
      HIP_TRY(hipMemcpyAsync(&ovf, t->d_packflag, sizeof(ovf), hipMemcpyDeviceToHost, g_ctx.stream));
      HIP_TRY(hipStreamSynchronize(g_ctx.stream));
      derived_waited(t);
      if (ovf) {                                        // a synced value outgrew its stored width: the projection is void (the caller drops it)
        HIP_TRY(hipMemsetAsync(t->d_packflag, 0, 256, g_ctx.stream));
        return VH_PACK_STALE;
      }
    } else if (int rc = derived_enqueued(t)) return rc;
  }
  for (uint32_t s = 0; s < t->nseg; ++s) pk->seg_mod[s] = t->seg_mod[s];
  pk->applied_epoch = t->sync_epoch;
  return VH_OK;
}
static void pack_drop(vh_table* t, VhPack* pk) {
  table_quiesce(t);
  (void)hipStreamSynchronize(g_ctx.stream);
  for (size_t k = 0; k < t->packs.size(); ++k) {
    if (t->packs[k].get() != pk) continue;
    if (pk->base) { (void)hipFree(pk->base); t->device_bytes -= (size_t)pk->cap_seg * pk->stride + 256; }
    t->packs.erase(t->packs.begin() + (long)k);
    return;
  }
}

// Bytes the values of an integer column need over every mirrored segment (1, 2, 4 or 8; the element size for floating point): dimensions
// from their SegmentStats, metrics from a min / max pass of their own (they keep no stats).
static int column_stored_width(vh_table* t, int col, int* width_out) {
  const VhColumn& c = t->cols[col];
  *width_out = (int)c.esize;
  if (c.elem == VH_F32 || c.elem == VH_F64 || c.esize == 1 || !t->nseg) return VH_OK;
  uint64_t lo = ~0ull, hi = 0;
  if ((size_t)col < t->stats.size() && t->stats[col].size() >= t->nseg) {      // (refresh_stats keeps min / max of every fixed-width column, metrics included)
    for (uint32_t s = 0; s < t->nseg; ++s) { const VhSegStat& st = t->stats[col][s]; if (st.lo > st.hi) continue; lo = std::min(lo, st.lo); hi = std::max(hi, st.hi); }
  } else {
    const uint32_t n = t->nseg;
    char* tmp = nullptr;
    HIP_TRY(hipMalloc(&tmp, (size_t)n * 16 + (size_t)n * 4 + 256));
    unsigned long long* d_st = reinterpret_cast<unsigned long long*>(tmp);
    uint32_t* d_rows = reinterpret_cast<uint32_t*>(tmp + (size_t)n * 16);
    std::vector<unsigned long long> init((size_t)n * 2);
    for (size_t i = 0; i < init.size(); i += 2) { init[i] = ~0ull; init[i + 1] = 0; }
    std::vector<uint32_t> hrows(n);
    for (uint32_t s = 0; s < n; ++s) hrows[s] = (uint32_t)t->seg_rows[s];
    hipError_t he = hipMemcpyAsync(d_st, init.data(), init.size() * 8, hipMemcpyHostToDevice, g_ctx.stream);
    if (he == hipSuccess) he = hipMemcpyAsync(d_rows, hrows.data(), hrows.size() * 4, hipMemcpyHostToDevice, g_ctx.stream);
    if (he == hipSuccess) {
      for (uint32_t first = 0; first < n; first += 32768) {       // (grid.y)
        const uint32_t cnt = std::min<uint32_t>(32768, n - first);
        dim3 grid((unsigned)std::min<uint64_t>(64, (t->segment_rows + 4095) / 4096), cnt);
        VH_ELEM_SWITCH(c.elem, (seg_minmax_kernel<T><<<grid, dim3(256), 0, g_ctx.stream>>>(reinterpret_cast<const T*>(c.base), c.stride / c.esize, d_rows + first, first, d_st + 2ull * first)));
      }
      he = hipGetLastError();
    }
    if (he == hipSuccess) he = hipMemcpyAsync(init.data(), d_st, init.size() * 8, hipMemcpyDeviceToHost, g_ctx.stream);
    if (he == hipSuccess) he = hipStreamSynchronize(g_ctx.stream);
    (void)hipFree(tmp);
    if (he != hipSuccess) return vh_fail(VH_E_DEVICE, "min / max pass over column %d: %s", col, hipGetErrorString(he));
    for (uint32_t s = 0; s < n; ++s) { if (init[2 * s] > init[2 * s + 1]) continue; lo = std::min<uint64_t>(lo, init[2 * s]); hi = std::max<uint64_t>(hi, init[2 * s + 1]); }
  }
  if (lo > hi) { *width_out = 1; return VH_OK; }       // no rows yet: anything fits (a later value that does not voids the projection)
  int w = (int)c.esize;
  if (c.elem == VH_I16 || c.elem == VH_I32 || c.elem == VH_I64) {
    const int64_t a = (int64_t)(lo ^ (1ull << 63)), b = (int64_t)(hi ^ (1ull << 63));       // (order key of a signed integer: the value with its sign bit flipped)
    w = (a >= INT8_MIN && b <= INT8_MAX) ? 1 : (a >= INT16_MIN && b <= INT16_MAX) ? 2 : (a >= INT32_MIN && b <= INT32_MAX) ? 4 : 8;
  } else {
    w = hi < 256 ? 1 : hi < 65536 ? 2 : hi <= 0xFFFFFFFFull ? 4 : 8;
  }
  *width_out = std::min(w, (int)c.esize);
  return VH_OK;
}

static int table_pack_locked(vh_table* t, const int32_t* cols, int32_t ncols, bool automatic, VhPack** out, bool compress) {
  if (!cols || ncols <= 0 || ncols > VH_PACK_MAX_COLS) return vh_fail(VH_E_INVALID, "vh_table_pack: 1..%d columns", VH_PACK_MAX_COLS);
  std::vector<int> order;
  for (int i = 0; i < ncols; ++i) {
    const int c = cols[i];
    if (c < 0 || (size_t)c >= t->cols.size() || is_bitset_elem(t->cols[c].elem)) return vh_fail(VH_E_INVALID, "vh_table_pack: column %d cannot be packed", c);
    if (std::find(order.begin(), order.end(), c) == order.end()) order.push_back(c);
  }
  std::vector<int> sorted_cols = order;
  std::sort(sorted_cols.begin(), sorted_cols.end());
  for (auto& pk : t->packs) {
    std::vector<int> have = pk->cols;
    std::sort(have.begin(), have.end());
    if (have != sorted_cols || pk->compressed != compress) continue;
    const int rc = pack_refresh(t, pk.get(), 0, t->nseg);
    if (rc == VH_PACK_STALE) { pack_drop(t, pk.get()); break; }      // built again below, at the widths the values need now
    if (out) *out = pk.get();
    return rc;
  }
  for (int attempt = 0; attempt < 2; ++attempt) {
    std::map<int, int> wof;
    for (int c : order) {
      int w = (int)t->cols[c].esize;
      if (compress) if (int rc = column_stored_width(t, c, &w)) return rc;
      wof[c] = w;
    }
    std::vector<int> ord = order;
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return wof[a] > wof[b]; });   // widest first: every field naturally aligned
    uint32_t bytes = 0;
    std::vector<uint32_t> off;
    std::vector<uint8_t> width;
    for (int c : ord) { off.push_back(bytes); width.push_back((uint8_t)wof[c]); bytes += (uint32_t)wof[c]; }
    if (bytes > 64) return vh_fail(VH_E_UNSUPPORTED, "vh_table_pack: %u payload bytes per row (max 64)", bytes);
    uint32_t rec = 8;
    while (rec < bytes) rec <<= 1;
    // bit fields instead of bytes when every column is a non-negative integer (by its recorded min / max) and the word comes out smaller
    std::vector<uint8_t> bitoff, bitw;
    bool bits = compress && !test_env("VH_NO_PACK_BITS") && t->nseg > 0;
    uint32_t used = 0;
    for (int c : ord) {
      if (!bits) break;
      const VhColumn& col = t->cols[c];
      if (col.elem == VH_F32 || col.elem == VH_F64 || (size_t)c >= t->stats.size() || t->stats[c].size() < t->nseg) { bits = false; break; }
      uint64_t lo = ~0ull, hi = 0;
      for (uint32_t sg = 0; sg < t->nseg; ++sg) { const VhSegStat& st = t->stats[c][sg]; if (st.lo > st.hi) continue; lo = std::min(lo, st.lo); hi = std::max(hi, st.hi); }
      if (lo > hi) lo = hi = order_key_of_bits(col.elem, 0);
      const bool sgn = col.elem == VH_I8 || col.elem == VH_I16 || col.elem == VH_I32 || col.elem == VH_I64;
      if (sgn && (int64_t)(lo ^ (1ull << 63)) < 0) { bits = false; break; }
      const uint64_t vmax = sgn ? (hi ^ (1ull << 63)) : bits_of_order_key(col.elem, hi);
      int b = 1; while (b < 64 && (vmax >> b)) ++b;
      bitoff.push_back((uint8_t)used); bitw.push_back((uint8_t)b); used += (uint32_t)b;
      if (used > 64) { bits = false; break; }
    }
    const uint32_t rec_bits = used <= 32 ? 4u : 8u;
    if (bits && rec_bits >= rec) bits = false;
    std::unique_ptr<VhPack> pk(new VhPack());
    pk->cols = ord; pk->off = off; pk->width = width; pk->rec_bytes = rec; pk->automatic = automatic; pk->compressed = compress;
    if (bits) { pk->bits = true; pk->bitoff = bitoff; pk->bitw = bitw; pk->rec_bytes = rec = rec_bits; for (auto& o : pk->off) o = 0; }
    pk->stride = (t->segment_rows + 255) / 256 * 256 * (uint64_t)rec;
    VhPack* raw = pk.get();
    t->packs.push_back(std::move(pk));
    const int rc = pack_refresh(t, raw, 0, t->nseg);
    if (rc == VH_PACK_STALE && attempt == 0) { pack_drop(t, raw); continue; }      // (a metric changed between the min / max pass and the copy)
    if (rc) { pack_drop(t, raw); return rc == VH_PACK_STALE ? vh_fail(VH_E_DEVICE, "vh_table_pack: values keep outgrowing their stored widths") : rc; }
    if (out) *out = raw;
    return VH_OK;
  }
  return VH_OK;
}

// ------------------------------------------------------- narrow predicate copies (vh_table_narrow)
// Width the column's values fit over segments [0, nseg): 1, 2, or 0 (not an unsigned 32-bit column, or its values need all 32 bits).
static int narrow_width_for(const vh_table* t, int col, uint32_t nseg) {
  const VhColumn& c = t->cols[col];
  if (c.elem != VH_U32 || (size_t)col >= t->stats.size()) return 0;
  uint64_t hi = 0;
  bool any = false;
  for (uint32_t s = 0; s < nseg && s < t->stats[col].size(); ++s) {
    const VhSegStat& st = t->stats[col][s];
    if (st.lo > st.hi) continue;          // empty segment
    hi = std::max(hi, st.hi); any = true;
  }
  if (!any) return 0;
  return hi < 256 ? 1 : hi < 65536 ? 2 : 0;
}
// Bring the narrow copy up to date with its column: the ranges that changed since it was last copied, ONE launch, no host wait.
static int narrow_refresh(vh_table* t, VhNarrow* nw, uint32_t first, uint32_t n) {
  (void)first; (void)n;
  if (!t->nseg) return VH_OK;
  if (nw->cap_seg < t->cap_seg) {
    table_quiesce(t);
    HIP_TRY(hipStreamSynchronize(g_ctx.stream)); derived_waited(t);
    char* nb = nullptr;
    const size_t bytes = (size_t)t->cap_seg * nw->stride + 256;
    HIP_TRY(hipMalloc(&nb, bytes));
    trace_alloc("narrow", nb, bytes);
    if (nw->base) {
      HIP_TRY(hipMemcpyAsync(nb, nw->base, (size_t)nw->cap_seg * nw->stride, hipMemcpyDeviceToDevice, g_ctx.stream));
      HIP_TRY(hipStreamSynchronize(g_ctx.stream));
      HIP_TRY(hipFree(nw->base));
      t->device_bytes -= (size_t)nw->cap_seg * nw->stride + 256;
    }
    nw->base = nb; nw->cap_seg = t->cap_seg;
    nw->seg_mod.resize(t->cap_seg, 0);
    t->device_bytes += bytes;
  }
  if (nw->applied_epoch == t->sync_epoch) return VH_OK;
  const VhColumn& c = t->cols[nw->col];
  std::vector<VhJob> jobs;
  derived_jobs(t, nw->applied_epoch, nw->seg_mod, t->padded_rows, &jobs);
  if (!jobs.empty()) {
    const VhJob* d_jobs = nullptr;
    if (int rc = derived_upload(t, jobs, &d_jobs)) return rc;
    if (nw->width == 1)
      hipLaunchKernelGGL((narrow_kernel<uint8_t>), dim3((unsigned)jobs.size()), dim3(256), 0, g_ctx.stream, reinterpret_cast<const uint32_t*>(c.base), c.stride / 4,
                         reinterpret_cast<uint8_t*>(nw->base), nw->stride, d_jobs);
    else
      hipLaunchKernelGGL((narrow_kernel<uint16_t>), dim3((unsigned)jobs.size()), dim3(256), 0, g_ctx.stream, reinterpret_cast<const uint32_t*>(c.base), c.stride / 4,
                         reinterpret_cast<uint16_t*>(nw->base), nw->stride / 2, d_jobs);
    HIP_TRY(hipGetLastError());
    if (int rc = derived_enqueued(t)) return rc;
  }
  for (uint32_t s = 0; s < t->nseg; ++s) nw->seg_mod[s] = t->seg_mod[s];
  nw->applied_epoch = t->sync_epoch;
  return VH_OK;
}
static void narrow_drop(vh_table* t, size_t k) {
  table_quiesce(t);
  VhNarrow* nw = t->narrows[k].get();
  if (nw->base) { (void)hipFree(nw->base); t->device_bytes -= (size_t)nw->cap_seg * nw->stride + 256; }
  t->narrows.erase(t->narrows.begin() + (long)k);
}
// The narrow copy of `col`, fresh for segments [0, nseg), or nullptr (none, or the values no longer fit: the copy is dropped).
static VhNarrow* narrow_usable(vh_table* t, int col, uint32_t nseg) {
  for (size_t k = 0; k < t->narrows.size(); ++k) {
    VhNarrow* nw = t->narrows[k].get();
    if (nw->col != col) continue;
    const int w = narrow_width_for(t, col, t->nseg);
    if (w == 0 || w > nw->width) { narrow_drop(t, k); return nullptr; }
    if (narrow_refresh(t, nw, 0, nseg) != VH_OK) return nullptr;
    return nw;
  }
  return nullptr;
}
static int table_narrow_locked(vh_table* t, int col, bool automatic) {
  if (col < 0 || (size_t)col >= t->cols.size()) return vh_fail(VH_E_INVALID, "vh_table_narrow: column %d", col);
  for (auto& nw : t->narrows) if (nw->col == col) return narrow_usable(t, col, t->nseg) ? VH_OK : VH_OK;
  const int w = narrow_width_for(t, col, t->nseg);
  if (!w) return VH_OK;                          // nothing to gain: not an unsigned 32-bit column, or it uses its bits
  std::unique_ptr<VhNarrow> nw(new VhNarrow());
  nw->col = col; nw->width = w; nw->automatic = automatic;
  nw->stride = t->padded_rows * (uint64_t)w;
  VhNarrow* raw = nw.get();
  t->narrows.push_back(std::move(nw));
  const int rc = narrow_refresh(t, raw, 0, t->nseg);
  if (rc) { narrow_drop(t, t->narrows.size() - 1); return rc; }
  return VH_OK;
}

// ------------------------------------------------------- bit-packed predicate projections (vh_table_predpack)
// Bits the values of a column need over every mirrored segment, from its recorded min / max (0: not an integer column, negative values, or no stats).
static int predpack_bits_for(const vh_table* t, int col) {
  const VhColumn& c = t->cols[col];
  if (c.elem == VH_F32 || c.elem == VH_F64 || is_bitset_elem(c.elem) || (size_t)col >= t->stats.size() || t->stats[col].size() < t->nseg) return 0;
  uint64_t lo = ~0ull, hi = 0;
  for (uint32_t s = 0; s < t->nseg; ++s) { const VhSegStat& st = t->stats[col][s]; if (st.lo > st.hi) continue; lo = std::min(lo, st.lo); hi = std::max(hi, st.hi); }
  if (lo > hi) return 0;
  const bool sgn = c.elem == VH_I8 || c.elem == VH_I16 || c.elem == VH_I32 || c.elem == VH_I64;
  if (sgn && (int64_t)(lo ^ (1ull << 63)) < 0) return 0;
  const uint64_t vmax = sgn ? (hi ^ (1ull << 63)) : hi;
  int b = 1;
  while (b < 64 && (vmax >> b)) ++b;
  return b;
}
static void predpack_drop(vh_table* t, size_t k) {
  table_quiesce(t);
  (void)hipStreamSynchronize(g_ctx.stream); derived_waited(t);
  VhPredPack* pp = t->predpacks[k].get();
  for (int q = 0; q < pp->nplanes; ++q) if (pp->pbase[q]) { (void)hipFree(pp->pbase[q]); t->device_bytes -= (size_t)pp->cap_seg * pp->pstride[q] + 256; }
  t->predpacks.erase(t->predpacks.begin() + (long)k);
}
static int predpack_refresh(vh_table* t, VhPredPack* pp) {
  if (!t->nseg) return VH_OK;
  if (pp->cap_seg < t->cap_seg) {
    table_quiesce(t);
    HIP_TRY(hipStreamSynchronize(g_ctx.stream)); derived_waited(t);
    for (int q = 0; q < pp->nplanes; ++q) {
      char* nb = nullptr;
      const size_t bytes = (size_t)t->cap_seg * pp->pstride[q] + 256;
      HIP_TRY(hipMalloc(&nb, bytes));
      trace_alloc("predicate plane", nb, bytes);
      if (pp->pbase[q]) {
        HIP_TRY(hipMemcpyAsync(nb, pp->pbase[q], (size_t)pp->cap_seg * pp->pstride[q], hipMemcpyDeviceToDevice, g_ctx.stream));
        HIP_TRY(hipStreamSynchronize(g_ctx.stream));
        HIP_TRY(hipFree(pp->pbase[q]));
        t->device_bytes -= (size_t)pp->cap_seg * pp->pstride[q] + 256;
      }
      pp->pbase[q] = nb;
      t->device_bytes += bytes;
    }
    pp->cap_seg = t->cap_seg;
    pp->seg_mod.resize(t->cap_seg, 0);
  }
  if (pp->applied_epoch == t->sync_epoch) return VH_OK;
  std::vector<VhJob> jobs;
  derived_jobs(t, pp->applied_epoch, pp->seg_mod, t->padded_rows, &jobs);
  if (!jobs.empty()) {
    const VhJob* d_jobs = nullptr;
    if (int rc = derived_upload(t, jobs, &d_jobs)) return rc;
    VhPredPackArgs A{};
    A.ncols = (int32_t)pp->cols.size(); A.nplanes = pp->nplanes;
    for (size_t c = 0; c < pp->cols.size(); ++c) {
      const VhColumn& col = t->cols[pp->cols[c]];
      A.src[c] = col.base; A.src_stride[c] = col.stride; A.esize[c] = (uint32_t)col.esize; A.bitoff[c] = pp->bitoff[c];
    }
    for (int q = 0; q < pp->nplanes; ++q) { A.plane[q] = pp->pbase[q]; A.plane_stride[q] = pp->pstride[q]; A.plane_width[q] = (uint32_t)pp->pwidth[q]; A.plane_pos[q] = (uint32_t)pp->ppos[q]; }
    A.jobs = d_jobs;
    if (pp->sliced) hipLaunchKernelGGL(predslice_kernel, dim3((unsigned)jobs.size()), dim3(256), 0, g_ctx.stream, A, pp->bits, pp->pitch);
    else hipLaunchKernelGGL(predpack_kernel, dim3((unsigned)jobs.size()), dim3(256), 0, g_ctx.stream, A);
    HIP_TRY(hipGetLastError());
    if (int rc = derived_enqueued(t)) return rc;
  }
  for (uint32_t s = 0; s < t->nseg; ++s) pp->seg_mod[s] = t->seg_mod[s];
  pp->applied_epoch = t->sync_epoch;
  return VH_OK;
}
// The projection that holds every column of `cols` (ascending), fresh, or nullptr. One whose fields no longer hold the recorded values is dropped.
static VhPredPack* predpack_usable(vh_table* t, const std::vector<int>& cols, int want_sliced = -1) {
  for (size_t k = 0; k < t->predpacks.size(); ++k) {
    VhPredPack* pp = t->predpacks[k].get();
    if (!std::includes(pp->cols.begin(), pp->cols.end(), cols.begin(), cols.end())) continue;
    if (want_sliced >= 0 && (int)pp->sliced != want_sliced) continue;
    bool fits = true;
    for (size_t c = 0; c < pp->cols.size(); ++c) { const int b = predpack_bits_for(t, pp->cols[c]); fits &= b > 0 && b <= (int)pp->bitw[c]; }
    if (!fits) { predpack_drop(t, k); return nullptr; }
    if (predpack_refresh(t, pp) != VH_OK) return nullptr;
    return pp;
  }
  return nullptr;
}
// Build one for `cols` (ascending, distinct). *built = nullptr when there is nothing to gain: a column that is no non-negative integer,
// more than 32 bits in all, or no fewer bytes per row than the columns' narrowest copies would take.
static int table_predpack_locked(vh_table* t, const std::vector<int>& cols, bool automatic, VhPredPack** built, bool sliced) {
  if (built) *built = nullptr;
  if (cols.empty() || cols.size() > VH_PACK_MAX_COLS || cols.size() > VJ_MAX_PRED) return VH_OK;
  for (auto& pp : t->predpacks) if (pp->cols == cols && pp->sliced == sliced) { if (built) *built = predpack_usable(t, cols, sliced ? 1 : 0); return VH_OK; }
  std::unique_ptr<VhPredPack> pp(new VhPredPack());
  uint32_t used = 0, plain = 0;
  for (int c : cols) {
    if (c < 0 || (size_t)c >= t->cols.size()) return vh_fail(VH_E_INVALID, "vh_table_predpack: column %d", c);
    const int b = predpack_bits_for(t, c);
    if (!b) return VH_OK;
    pp->cols.push_back(c); pp->bitoff.push_back((uint8_t)used); pp->bitw.push_back((uint8_t)b);
    used += (uint32_t)b;
    const int nwid = narrow_width_for(t, c, t->nseg);
    plain += nwid ? (uint32_t)nwid : (uint32_t)t->cols[c].esize;
  }
  if (used > 32) return VH_OK;
  if (sliced) {          // `used` planes of one bit per row; a plane's share of a segment padded to whole 256-byte blocks
    pp->sliced = true; pp->bits = used;
    pp->pitch = (t->padded_rows / 8 + 255) / 256 * 256;
    pp->nplanes = 1; pp->pwidth[0] = 0; pp->ppos[0] = 0; pp->pstride[0] = pp->pitch * used;
    if (used >= plain * 8u) return VH_OK;
  } else {
    for (uint32_t left = used, pos = 0; left > 0;) {
      const int w = left > 8 ? 2 : 1;
      pp->pwidth[pp->nplanes] = w; pp->ppos[pp->nplanes] = (int)pos; pp->pstride[pp->nplanes] = t->padded_rows * (uint64_t)w;
      ++pp->nplanes;
      pos += 8u * w; left = left > 8u * w ? left - 8u * w : 0;
    }
    if (pp->bytes_per_row() >= plain) return VH_OK;
  }
  pp->automatic = automatic;
  VhPredPack* raw = pp.get();
  t->predpacks.push_back(std::move(pp));
  const int rc = predpack_refresh(t, raw);
  if (rc) { predpack_drop(t, t->predpacks.size() - 1); return rc; }
  if (built) *built = raw;
  return VH_OK;
}

extern "C" int vh_table_predpack(vh_table* t, const int32_t* cols, int32_t ncols) { return vh_table_predpack_ex(t, cols, ncols, VH_PREDPACK_AUTO); }
extern "C" int vh_table_predpack_ex(vh_table* t, const int32_t* cols, int32_t ncols, uint32_t form) {
  if (!t || !cols || ncols <= 0) return vh_fail(VH_E_INVALID, "vh_table_predpack: null argument");
  if (form > VH_PREDPACK_SLICED) return vh_fail(VH_E_INVALID, "vh_table_predpack_ex: form %u", form);
  VH_ENTER();
  std::lock_guard<std::mutex> lk(t->mu);
  if (int src = sync_resolve(t)) return src;
  std::vector<int> set(cols, cols + ncols);
  std::sort(set.begin(), set.end());
  set.erase(std::unique(set.begin(), set.end()), set.end());
  return table_predpack_locked(t, set, false, nullptr, form == VH_PREDPACK_AUTO ? !knobs().predpack_bytes : form == VH_PREDPACK_SLICED);
}

extern "C" int vh_table_narrow(vh_table* t, const int32_t* cols, int32_t ncols) {
  if (!t || (!cols && ncols)) return vh_fail(VH_E_INVALID, "null argument");
  VH_ENTER();
  std::lock_guard<std::mutex> lk(t->mu);
  if (int src = sync_resolve(t)) return src;
  for (int i = 0; i < ncols; ++i)
    if (int rc = table_narrow_locked(t, cols[i], false)) return rc;
  return VH_OK;
}

extern "C" int vh_table_pack_ex(vh_table* t, const int32_t* cols, int32_t ncols, uint32_t form) {
  if (!t) return vh_fail(VH_E_INVALID, "null table");
  if (form > VH_PACK_COMPRESSED) return vh_fail(VH_E_INVALID, "vh_table_pack_ex: form %u", form);
  VH_ENTER();
  std::lock_guard<std::mutex> lk(t->mu);
  if (int src = sync_resolve(t)) return src;
  // VH_PACK_AUTO: compressed where the per-query compiled kernels — the only readers of compressed records — would run a scan of the
  // whole table (VH_JIT=force, or auto and the table holds VH_JIT_MIN_ROWS rows); plain where the pre-built kernels answer
  bool compress = form == VH_PACK_COMPRESSED;
  if (form == VH_PACK_AUTO) {
    uint64_t rows = 0;
    for (uint32_t s = 0; s < t->nseg; ++s) rows += t->seg_rows[s];
    compress = !knobs().pack_plain && (vh_jit_policy() == VH_JIT_FORCE || (vh_jit_policy() == VH_JIT_AUTO && rows >= vh_jit_min_rows()));
  }
  return table_pack_locked(t, cols, ncols, false, nullptr, compress);
}
extern "C" int vh_table_pack(vh_table* t, const int32_t* cols, int32_t ncols) { return vh_table_pack_ex(t, cols, ncols, VH_PACK_AUTO); }

// WHERE the derived layouts lie. The same records and planes read by the same kernel take 1.07 or 1.23 ms per 1 B rows depending on the physical
// pages they were given (profiles/r06/NOTES.md, "Placement": six execution contexts with six scratch allocations agree within 0.5 %, the
// projection alone moved twenty-three times changes nothing, projection AND planes re-built behind 3 GB spacers spread over 15 % — all at
// 2 MB-aligned virtual addresses, so it is nothing a process can compute). What a process can do is try: derived_move copies every layout `which`
// names (1: projections, 2: predicate planes) to FRESH allocations while the old ones are still held — so that the new ones are other pages —
// and swaps the pointers (kernels take addresses as arguments); the caller measures and keeps or gives back (vh_table_prepare, vh_table_relocate).
struct VhMoved { int kind; void* owner; int plane; char* old_ptr; char* new_ptr; size_t bytes; };      // kind 1: VhPack* owner, 2: VhPredPack* owner
static size_t derived_bytes(const vh_table* t, uint32_t which) {
  size_t b = 0;
  if (which & 1u) for (auto& pk : t->packs) if (pk->base) b += (size_t)pk->cap_seg * pk->stride + 256;
  if (which & 2u) for (auto& pp : t->predpacks) for (int q = 0; q < pp->nplanes; ++q) if (pp->pbase[q]) b += (size_t)pp->cap_seg * pp->pstride[q] + 256;
  return b;
}
static int derived_move(vh_table* t, uint32_t which, std::vector<VhMoved>* moved) {      // (t->mu held)
  table_quiesce(t);
  HIP_TRY(hipStreamSynchronize(g_ctx.stream)); derived_waited(t);
  auto move = [&](int kind, void* owner, int plane, char*& base, size_t bytes, const char* what) -> int {
    char* nb = nullptr;
    if (hipMalloc(&nb, bytes) != hipSuccess) { (void)hipGetLastError(); return VH_E_NOMEM; }
    trace_alloc(what, nb, bytes);
    if (hipMemcpyAsync(nb, base, bytes, hipMemcpyDeviceToDevice, g_ctx.stream) != hipSuccess) { (void)hipFree(nb); return vh_fail(VH_E_DEVICE, "moving a derived layout"); }
    moved->push_back(VhMoved{kind, owner, plane, base, nb, bytes});
    base = nb;
    return VH_OK;
  };
  int rc = VH_OK;
  if (which & 1u) for (auto& pk : t->packs) if (pk->base && !rc) rc = move(1, pk.get(), 0, pk->base, (size_t)pk->cap_seg * pk->stride + 256, "projection");
  if (which & 2u) for (auto& pp : t->predpacks) for (int q = 0; q < pp->nplanes && !rc; ++q) if (pp->pbase[q]) rc = move(2, pp.get(), q, pp->pbase[q], (size_t)pp->cap_seg * pp->pstride[q] + 256, "predicate plane");
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  return rc == VH_E_NOMEM ? VH_OK : rc;          // (out of memory: what could be moved was moved)
}
// The buffers a move left behind (keep = true), or the ones it made after the layouts were pointed back at the old ones (keep = false), into
// `out` — still allocated: whoever tries several places frees them all at the end, so that no candidate lands on a place already tried.
static void derived_settle(vh_table* t, std::vector<VhMoved>& moved, bool keep, std::vector<char*>* out) {      // (t->mu held)
  table_quiesce(t);
  (void)hipStreamSynchronize(g_ctx.stream);
  for (const VhMoved& m : moved) {
    char** slot = nullptr;      // the layout may be gone by now (a sync voided it): then both buffers are somebody else's or nobody's — only ours is freed
    if (m.kind == 1) { for (auto& pk : t->packs) if (pk.get() == m.owner && pk->base == m.new_ptr) slot = &pk->base; }
    else { for (auto& pp : t->predpacks) if (pp.get() == m.owner && pp->pbase[m.plane] == m.new_ptr) slot = &pp->pbase[m.plane]; }
    if (!slot) { out->push_back(m.old_ptr); continue; }      // (the layout was dropped or moved again with new_ptr freed by its owner: the old buffer is ours to free)
    if (keep) out->push_back(m.old_ptr);
    else { *slot = m.old_ptr; out->push_back(m.new_ptr); }
  }
  moved.clear();
}
extern "C" int vh_table_relocate(vh_table* t, uint32_t which) {
  if (!t) return vh_fail(VH_E_INVALID, "null table");
  VH_ENTER();
  std::vector<char*> drop;
  {
    std::lock_guard<std::mutex> lk(t->mu);
    if (int src = sync_resolve(t)) return src;
    std::vector<VhMoved> moved;
    if (int rc = derived_move(t, which ? which : 3u, &moved)) { derived_settle(t, moved, false, &drop); for (char* p : drop) (void)hipFree(p); return rc; }
    derived_settle(t, moved, true, &drop);
  }
  for (char* p : drop) (void)hipFree(p);
  return VH_OK;
}

extern "C" int vh_table_unpack(vh_table* t) {
  if (!t) return vh_fail(VH_E_INVALID, "null table");
  VH_ENTER();
  std::lock_guard<std::mutex> lk(t->mu);
  if (int src = sync_resolve(t)) return src;
  table_quiesce(t);
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  for (auto& pk : t->packs) if (pk->base) { (void)hipFree(pk->base); t->device_bytes -= (size_t)pk->cap_seg * pk->stride + 256; }
  t->packs.clear();
  t->gather_seen.clear();
  for (auto& nw : t->narrows) if (nw->base) { (void)hipFree(nw->base); t->device_bytes -= (size_t)nw->cap_seg * nw->stride + 256; }
  t->narrows.clear();
  t->pred_seen.clear();
  for (auto& pp : t->predpacks) for (int q = 0; q < pp->nplanes; ++q) if (pp->pbase[q]) { (void)hipFree(pp->pbase[q]); t->device_bytes -= (size_t)pp->cap_seg * pp->pstride[q] + 256; }
  t->predpacks.clear();
  t->ppred_seen.clear();
  return VH_OK;
}


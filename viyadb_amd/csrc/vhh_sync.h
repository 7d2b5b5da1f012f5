// vhh_sync.h — host side of libviya_hip, part of viya_hip.hip's translation unit (included there, in order; not a stand-alone header):
// data in: vh_segment_sync*, the per-segment min / max pass (refresh_stats), vh_segment_generate, vh_segment_read.
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
  t->seg_rows[seg] = nrows;
  t->nseg = std::max(t->nseg, seg + 1);
  t->seg_mod[seg] = ++t->sync_epoch;
  return refresh_stats(t, seg, 1);
}

extern "C" int vh_segment_sync_range(vh_table* t, uint32_t seg, uint64_t row_first, uint64_t nrows, uint64_t new_size,
                                     const void* const* col_ptrs) {
  if (!t || !col_ptrs) return vh_fail(VH_E_INVALID, "vh_segment_sync_range: null argument");
  if (new_size > t->segment_rows || row_first + nrows > new_size)
    return vh_fail(VH_E_INVALID, "vh_segment_sync_range: rows [%llu, %llu) do not fit a segment of %llu rows",
                   (unsigned long long)row_first, (unsigned long long)(row_first + nrows), (unsigned long long)new_size);
  if (seg >= VH_MAX_SEGMENTS) return vh_fail(VH_E_INVALID, "vh_segment_sync_range: segment index %u out of range", seg);
  VH_ENTER();
  std::lock_guard<std::mutex> lk(t->mu);
  table_quiesce(t);
  int rc = table_grow(t, seg + 1);
  if (rc) return rc;
  if (row_first > t->seg_rows[seg])
    return vh_fail(VH_E_INVALID, "vh_segment_sync_range: segment %u has %llu mirrored rows, range starts at %llu (gap)",
                   seg, (unsigned long long)t->seg_rows[seg], (unsigned long long)row_first);
  for (size_t i = 0; i < t->cols.size(); ++i) {
    auto& c = t->cols[i];
    if (is_bitset_elem(c.elem) || !col_ptrs[i] || !nrows) continue;
    HIP_TRY(hipMemcpyAsync(c.base + (size_t)seg * c.stride + row_first * c.esize,
                           static_cast<const char*>(col_ptrs[i]) + row_first * c.esize, (size_t)nrows * c.esize,
                           hipMemcpyHostToDevice, g_ctx.stream));
  }
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  t->seg_rows[seg] = new_size;
  t->nseg = std::max(t->nseg, seg + 1);
  t->seg_mod[seg] = ++t->sync_epoch;
  return refresh_stats(t, seg, 1);   // one pass over the segment's dimension columns in HBM
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
  table_quiesce(t);
  int rc = table_grow(t, seg_first + nseg);
  if (rc) return rc;
  for (size_t i = 0; i < t->cols.size(); ++i) {
    auto& c = t->cols[i];
    if (specs[i].mode == VH_GEN_UNIFORM && specs[i].mod == 0) return vh_fail(VH_E_INVALID, "column %zu: mod == 0", i);
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
                               rows_per_seg, row_base, specs[i], colseed)));
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  ++t->sync_epoch;
  for (uint32_t s = 0; s < nseg; ++s) { t->seg_rows[seg_first + s] = rows_per_seg; t->seg_mod[seg_first + s] = t->sync_epoch; }
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


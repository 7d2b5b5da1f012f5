// vhh_exchange.h — host side of libviya_hip, part of viya_hip.hip's translation unit (included there, in order; not a stand-alone header):
// partials for the multi-GPU exchange: vh_result_device_buffers, vh_result_partition, vh_result_partition_pairs.
extern "C" int vh_result_device_buffers(vh_result* r, vh_device_buffer* bufs, int32_t max_bufs, int32_t* nbufs) {
  if (!r || !bufs || !nbufs) return vh_fail(VH_E_INVALID, "null argument");
  if (r->plan.nbitset) return vh_fail(VH_E_UNSUPPORTED, "count-distinct partials are cardinalities: they cannot be reduced across GPUs");
  if (r->mode == VH_MODE_HASH) return vh_fail(VH_E_UNSUPPORTED, "hash-path partials are exchanged by key, not reduced in place");
  const VhPlanDev& P = r->plan;
  int n = 0;
  if (max_bufs < P.nmetric + 1) return vh_fail(VH_E_INVALID, "need %d buffers", P.nmetric + 1);
  if (r->unmerged) {        // the caller reduces copy 0 across GPUs: the private copies go into it first (a small result leaves them to its one tail launch)
    if (!r->exec) return vh_fail(VH_E_INVALID, "a launched result without its context");
    VH_ENTER();
    if (int mrc = merge_copies_now(r, r->stream_for_work())) return mrc;
  }
  // presence bytes are only written when no SUM state carries the flag (SOP_ADD32P): one collective less
  if (!((r->mode == VH_MODE_DENSE_GLOBAL || r->mode == VH_MODE_DENSE_PART) && P.present_carrier >= 0)) bufs[n++] = vh_device_buffer{P.present, P.G, VH_U8, VH_RED_MAX};
  for (int j = 0; j < P.nmetric; ++j) {
    vh_device_buffer b{P.m[j].state, P.G, 0, VH_RED_SUM};
    switch (P.m[j].sop()) {
      case SOP_ADD32: b.elem = VH_U32; break;
      case SOP_ADD64: case SOP_ADD32P: b.elem = VH_U64; break;
      case SOP_ADDF32: b.elem = VH_F32; break;
      case SOP_ADDF64: b.elem = VH_F64; break;
      case SOP_MIN_I32: b.elem = VH_I32; b.reduce = VH_RED_MIN; break;
      case SOP_MAX_I32: b.elem = VH_I32; b.reduce = VH_RED_MAX; break;
      case SOP_MIN_U32: b.elem = VH_U32; b.reduce = VH_RED_MIN; break;
      case SOP_MAX_U32: b.elem = VH_U32; b.reduce = VH_RED_MAX; break;
      case SOP_MIN_I64: b.elem = VH_I64; b.reduce = VH_RED_MIN; break;
      case SOP_MAX_I64: b.elem = VH_I64; b.reduce = VH_RED_MAX; break;
      case SOP_MIN_U64: b.elem = VH_U64; b.reduce = VH_RED_MIN; break;
      case SOP_MAX_U64: b.elem = VH_U64; b.reduce = VH_RED_MAX; break;
      case SOP_MIN_F32: b.elem = VH_F32; b.reduce = VH_RED_MIN; break;
      case SOP_MAX_F32: b.elem = VH_F32; b.reduce = VH_RED_MAX; break;
      case SOP_MIN_F64: b.elem = VH_F64; b.reduce = VH_RED_MIN; break;
      default: b.elem = VH_F64; b.reduce = VH_RED_MAX; break;
    }
    // Integer SUM states of the same width that sit back to back in scratch (they do: all zero-identity states are
    // laid out contiguously and cleared by one memset, alignment gaps included) merge into ONE buffer — the
    // collective is latency-bound at this size (C4: 2 x 800 KB), so fewer, larger calls is the whole game.
    if (n > 0 && b.reduce == VH_RED_SUM && bufs[n - 1].reduce == VH_RED_SUM && (b.elem == VH_U64 || b.elem == VH_U32) &&
        bufs[n - 1].elem == b.elem && P.m[j].ident == 0 && r->nxcd == 1) {
      const size_t es = vh_elem_size(b.elem);
      char* prev_end = static_cast<char*>(bufs[n - 1].ptr) + bufs[n - 1].count * es;
      char* cur = static_cast<char*>(b.ptr);
      if (cur >= prev_end && (size_t)(cur - prev_end) < 4096 && (size_t)(cur - prev_end) % es == 0 && r->zero_begin <= bufs[n - 1].ptr &&
          cur + b.count * es <= r->zero_end) {
        bufs[n - 1].count = (uint64_t)((cur + b.count * es) - static_cast<char*>(bufs[n - 1].ptr)) / es;
        continue;
      }
    }
    bufs[n++] = b;
  }
  *nbufs = n;
  return VH_OK;
}

// SURVEY 8(e), hash path: "each GPU radix-partitions its partial table by hash(key) mod nGPU -> all-to-all ->
// local merge on the owned partition". This is the first step, on a finalised result: its emitted rows are
// regrouped by owner in HBM so that every column is one contiguous send buffer per destination.
extern "C" int vh_result_partition(vh_result* r, uint32_t nparts, uint64_t* part_offsets, vh_device_buffer* bufs,
                                   int32_t max_bufs, int32_t* nbufs) {
  if (!r || !part_offsets || !bufs || !nbufs) return vh_fail(VH_E_INVALID, "null argument");
  if (!r->finalized) return vh_fail(VH_E_INVALID, "result is not finalised");
  if (nparts == 0 || nparts > 64) return vh_fail(VH_E_INVALID, "nparts must be 1..64");
  if (r->nhaving) return vh_fail(VH_E_UNSUPPORTED, "HAVING applies to merged groups: run the partial query without it");
  if (r->topk) return vh_fail(VH_E_UNSUPPORTED, "top-N applies to merged groups: run the partial query without it");
  VH_ENTER();
  const VhPlanDev& P = r->plan;
  hipStream_t st = r->stream_for_work();
  const uint64_t ng = r->ngroups_host;
  const int ncols = P.ngroup + P.nmetric;
  if (max_bufs < ncols) return vh_fail(VH_E_INVALID, "need %d buffers", ncols);
  VhPartitionArgs A{};
  A.n = ng; A.nparts = nparts; A.nkeys = P.ngroup; A.ncols = ncols;
  // output order: key columns, then the plan's metrics in plan order, then the hidden count (if any)
  std::vector<int> order;
  for (size_t j = 0; j < r->user_metric.size(); ++j) order.push_back(r->user_metric[j]);
  if (r->info.has_hidden_count) order.push_back(P.nmetric - 1);
  if ((int)order.size() != P.nmetric) return vh_fail(VH_E_DEVICE, "metric bookkeeping is inconsistent");
  size_t bytes = 0;
  std::vector<size_t> off(ncols);
  for (int c = 0; c < ncols; ++c) {
    const int elem = c < P.ngroup ? P.g[c].type() : r->metric_elem[order[c - P.ngroup]];
    A.esize[c] = (uint32_t)vh_elem_size(elem);
    A.src[c] = c < P.ngroup ? r->d_out_key[c] : r->d_out_state[order[c - P.ngroup]];
    off[c] = bytes;
    bytes += ((size_t)std::max<uint64_t>(ng, 1) * A.esize[c] + 255) / 256 * 256;
  }
  const size_t ctr_off = bytes;
  bytes += 3 * 64 * sizeof(unsigned long long) + 8;
  if (r->d_xchg) { (void)hipFree(r->d_xchg); r->d_xchg = nullptr; }
  HIP_TRY(hipMalloc((void**)&r->d_xchg, bytes));
  unsigned long long* ctr = reinterpret_cast<unsigned long long*>(r->d_xchg + ctr_off);
  HIP_TRY(hipMemsetAsync(ctr, 0, 3 * 64 * sizeof(unsigned long long) + 8, st));
  for (int c = 0; c < ncols; ++c) A.dst[c] = r->d_xchg + off[c];
  A.counts = ctr; A.cursors = ctr + 64;
  std::vector<unsigned long long> counts(nparts, 0), offs(nparts + 1, 0);
  if (ng) {
    const unsigned grid = (unsigned)((ng + 256 * VH_XCHG_SPAN - 1) / (256 * VH_XCHG_SPAN));
    A.pass = 0; A.offsets = nullptr;
    hipLaunchKernelGGL(partition_groups_kernel, dim3(grid), dim3(256), 0, st, A);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(counts.data(), ctr, nparts * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (uint32_t p = 0; p < nparts; ++p) offs[p + 1] = offs[p] + counts[p];
    if (offs[nparts] != ng) return vh_fail(VH_E_DEVICE, "partition counted %llu of %llu rows", offs[nparts], (unsigned long long)ng);
    HIP_TRY(hipMemcpyAsync(ctr + 128, offs.data(), (nparts + 1) * sizeof(unsigned long long), hipMemcpyHostToDevice, st));
    A.pass = 1; A.offsets = ctr + 128;
    hipLaunchKernelGGL(partition_groups_kernel, dim3(grid), dim3(256), 0, st, A);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));   // offs lives on this frame
  }
  for (uint32_t p = 0; p <= nparts; ++p) part_offsets[p] = offs[p];
  for (int c = 0; c < ncols; ++c) {
    vh_device_buffer b{A.dst[c], ng, 0, -1};
    if (c < P.ngroup) b.elem = P.g[c].type();
    else {
      const int u = order[c - P.ngroup];
      b.elem = r->metric_elem[u];
      switch (P.m[u].sop()) {
        case SOP_MIN_I32: case SOP_MIN_U32: case SOP_MIN_I64: case SOP_MIN_U64: case SOP_MIN_F32: case SOP_MIN_F64: b.reduce = VH_RED_MIN; break;
        case SOP_MAX_I32: case SOP_MAX_U32: case SOP_MAX_I64: case SOP_MAX_U64: case SOP_MAX_F32: case SOP_MAX_F64: b.reduce = VH_RED_MAX; break;
        case SOP_BITSET: b.reduce = -2; break;   // cardinalities do not merge: exchange the pairs (vh_result_partition_pairs)
        default: b.reduce = VH_RED_SUM; break;
      }
    }
    bufs[c] = b;
  }
  *nbufs = ncols;
  return VH_OK;
}

// Count-distinct partials for the exchange: the distinct (group, id) pairs of bitset metric `metric` (index into the
// plan's metrics), as key columns + an id column, regrouped by the owner of the GROUP (same function as
// vh_result_partition). See partition_pairs_kernel.
extern "C" int vh_result_partition_pairs(vh_result* r, int32_t metric, uint32_t nparts, uint64_t* part_offsets,
                                         vh_device_buffer* bufs, int32_t max_bufs, int32_t* nbufs) {
  if (!r || !part_offsets || !bufs || !nbufs) return vh_fail(VH_E_INVALID, "null argument");
  if (!r->finalized) return vh_fail(VH_E_INVALID, "result is not finalised");
  if (nparts == 0 || nparts > 64) return vh_fail(VH_E_INVALID, "nparts must be 1..64");
  if (metric < 0 || metric >= (int)r->user_metric.size()) return vh_fail(VH_E_INVALID, "metric %d is not in the plan", metric);
  const VhPlanDev& P = r->plan;
  const int dj = r->user_metric[metric];
  if (P.m[dj].sop() != SOP_BITSET) return vh_fail(VH_E_INVALID, "metric %d is not a bitset (count-distinct) metric", metric);
  const int b = (int)P.m[dj].slot();
  if (max_bufs < P.ngroup + 1) return vh_fail(VH_E_INVALID, "need %d buffers", P.ngroup + 1);
  if (r->hpart) {
    // hashed partitioning: no device-wide set was built; the ids are read out of the last tuple pool (hp_partition_pairs_kernel), every
    // one a rank saw — the count is only known after the counting pass, so the buffers are allocated between the passes
    if (r->hp_args.units != 2 && !r->hp_args.pk) return vh_fail(VH_E_INVALID, "the hashed partitioning carried no ids for metric %d", metric);
    VH_ENTER();
    hipStream_t st = r->stream_for_work();
    const VhHpPool& B = r->hp_args.k[0].b;
    VhHpPairArgs A{};
    A.tuples = B.tuples; A.fill = B.fill; A.max_extents = B.max_extents; A.stride = B.stride; A.et = (uint32_t)(HP_ET / r->hp_args.units);
    A.pk = r->hp_args.pk; A.pk_pbits = r->hp_args.pk_pbits; A.pk_idbits = r->hp_args.pk_idbits;
    A.ngroup = P.ngroup; A.nparts = nparts;
    for (int c = 0; c < P.ngroup; ++c) { A.gkey_shift[c] = P.g[c].key_shift(); A.gesize[c] = (uint32_t)vh_elem_size(P.g[c].type()); }
    char* ctrbuf = nullptr;
    HIP_TRY(hipMalloc((void**)&ctrbuf, 3 * 64 * sizeof(unsigned long long) + 8));
    r->d_pairs.push_back(ctrbuf);
    unsigned long long* ctr = reinterpret_cast<unsigned long long*>(ctrbuf);
    HIP_TRY(hipMemsetAsync(ctr, 0, 3 * 64 * sizeof(unsigned long long) + 8, st));
    A.counts = ctr; A.cursors = ctr + 64;
    const uint64_t items = (uint64_t)A.max_extents * A.et * 2;
    const unsigned grid = (unsigned)std::max<uint64_t>(1, (items + 256 * VH_XCHG_SPAN - 1) / (256 * VH_XCHG_SPAN));
    std::vector<unsigned long long> counts(nparts, 0), offs(nparts + 1, 0);
    A.pass = 0; A.offsets = nullptr;
    hipLaunchKernelGGL(hp_partition_pairs_kernel, dim3(grid), dim3(256), 0, st, A);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(counts.data(), ctr, nparts * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (uint32_t q = 0; q < nparts; ++q) offs[q + 1] = offs[q] + counts[q];
    const uint64_t np = offs[nparts];
    size_t bytes = 0;
    std::vector<size_t> off(P.ngroup + 1);
    for (int c = 0; c <= P.ngroup; ++c) {
      const uint32_t es = c < P.ngroup ? A.gesize[c] : 4u;
      off[c] = bytes;
      bytes += ((size_t)std::max<uint64_t>(np, 1) * es + 255) / 256 * 256;
    }
    char* buf = nullptr;
    HIP_TRY(hipMalloc((void**)&buf, bytes));
    r->d_pairs.push_back(buf);
    for (int c = 0; c <= P.ngroup; ++c) A.dst[c] = buf + off[c];
    if (np) {
      HIP_TRY(hipMemcpyAsync(ctr + 128, offs.data(), (nparts + 1) * sizeof(unsigned long long), hipMemcpyHostToDevice, st));
      A.pass = 1; A.offsets = ctr + 128;
      hipLaunchKernelGGL(hp_partition_pairs_kernel, dim3(grid), dim3(256), 0, st, A);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipStreamSynchronize(st));
    }
    for (uint32_t q = 0; q <= nparts; ++q) part_offsets[q] = offs[q];
    for (int c = 0; c <= P.ngroup; ++c) bufs[c] = vh_device_buffer{A.dst[c], np, c < P.ngroup ? (int32_t)P.g[c].type() : VH_U32, -1};
    *nbufs = P.ngroup + 1;
    return VH_OK;
  }
  if (r->mode != VH_MODE_HASH && r->nxcd != 1) return vh_fail(VH_E_UNSUPPORTED, "pairs of an XCD-private dense table");
  VH_ENTER();
  hipStream_t st = r->stream_for_work();
  // number of pairs = sum of the emitted cardinalities would need a reduction; the set's fill count is counters[4],
  // read back with the result header (h_base): every pair bumps it exactly once
  const uint64_t npairs = reinterpret_cast<const unsigned long long*>(r->h_base)[4];
  VhPairArgs A{};
  A.mode = r->mode == VH_MODE_DENSE_PART ? VH_MODE_DENSE_GLOBAL : r->mode; A.ngroup = P.ngroup; A.key_words = P.key_words; A.wide = P.bs_wide[b];
  A.nslots = P.dset_mask[b] + 1; A.hcap = P.hmask + 1; A.hkeys = P.hkeys; A.hstride = P.hrec_bytes ? P.hrec_bytes / 8u : (uint64_t)P.key_words;
  A.dkeys = P.dset_keys[b]; A.dtags = P.dset_tags[b];
  size_t bytes = 0;
  std::vector<size_t> off(P.ngroup + 1);
  for (int c = 0; c <= P.ngroup; ++c) {
    const uint32_t es = c < P.ngroup ? (uint32_t)vh_elem_size(P.g[c].type()) : (A.wide ? 8u : 4u);
    if (c < P.ngroup) {
      A.glo[c] = P.g[c].lo; A.gextent[c] = P.g[c].extent; A.gstride[c] = P.g[c].stride;
      A.gkey_word[c] = P.g[c].key_word(); A.gkey_shift[c] = P.g[c].key_shift(); A.gesize[c] = es;
    }
    off[c] = bytes;
    bytes += ((size_t)std::max<uint64_t>(npairs, 1) * es + 255) / 256 * 256;
  }
  const size_t ctr_off = bytes;
  bytes += 3 * 64 * sizeof(unsigned long long) + 8;
  char* buf = nullptr;
  HIP_TRY(hipMalloc((void**)&buf, bytes));
  r->d_pairs.push_back(buf);
  unsigned long long* ctr = reinterpret_cast<unsigned long long*>(buf + ctr_off);
  HIP_TRY(hipMemsetAsync(ctr, 0, 3 * 64 * sizeof(unsigned long long) + 8, st));
  for (int c = 0; c <= P.ngroup; ++c) A.dst[c] = buf + off[c];
  A.nparts = nparts; A.counts = ctr; A.cursors = ctr + 64;
  std::vector<unsigned long long> counts(nparts, 0), offs(nparts + 1, 0);
  if (npairs) {
    const unsigned grid = (unsigned)((A.nslots + 256 * VH_XCHG_SPAN - 1) / (256 * VH_XCHG_SPAN));
    A.pass = 0; A.offsets = nullptr;
    hipLaunchKernelGGL(partition_pairs_kernel, dim3(grid), dim3(256), 0, st, A);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(counts.data(), ctr, nparts * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (uint32_t p = 0; p < nparts; ++p) offs[p + 1] = offs[p] + counts[p];
    if (offs[nparts] != npairs) return vh_fail(VH_E_DEVICE, "pair partition counted %llu of %llu pairs", offs[nparts], (unsigned long long)npairs);
    HIP_TRY(hipMemcpyAsync(ctr + 128, offs.data(), (nparts + 1) * sizeof(unsigned long long), hipMemcpyHostToDevice, st));
    A.pass = 1; A.offsets = ctr + 128;
    hipLaunchKernelGGL(partition_pairs_kernel, dim3(grid), dim3(256), 0, st, A);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
  }
  for (uint32_t p = 0; p <= nparts; ++p) part_offsets[p] = offs[p];
  for (int c = 0; c <= P.ngroup; ++c)
    bufs[c] = vh_device_buffer{A.dst[c], npairs, c < P.ngroup ? (int32_t)P.g[c].type() : (A.wide ? VH_U64 : VH_U32), -1};
  *nbufs = P.ngroup + 1;
  return VH_OK;
}


// vhh_finalize.h — host side of libviya_hip, part of viya_hip.hip's translation unit (included there, in order; not a stand-alone header):
// the end of a query: emission, delivery to pinned host memory, verdict and re-plan loop; vh_query_agg, vh_table_prepare.
// The end of a query is a host wait for a few hundred microseconds to a few milliseconds of device work: poll the event
// (a blocking hipStreamSynchronize adds tens of microseconds of wake-up latency to every query), fall back to a
// blocking wait when the work turns out to be long.
static hipError_t wait_event_spinning(hipEvent_t ev) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t e = hipEventQuery(ev);
    if (e != hipErrorNotReady) return e;
    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) return hipEventSynchronize(ev);
  }
}

// The second pass of a hashed partitioning that marked heavy ranges (VhPlanDev::heavy_mark): the caller's plan once more, through the plain hash
// organisation, over the rows of exactly those ranges (the generic scan drops every other survivor behind its key: VhPlanDev::heavy_only) — a group
// with a tenth of the table's rows, or with more ids than a range's LDS set takes, costs its query that, not the whole organisation. In two steps,
// so that a big result's rows cross PCIe WHILE the pass scans (C5 with a hot key: 10 ms of copy over 8.5 ms of scan): heavy_pass_launch enqueues
// the pass on a context of its own, heavy_pass_finish waits for it (and re-runs it with a bigger table when that one filled). H.r2: the
// finalised result (nullptr: it could not be had — the caller falls back as before).
static int result_finalize(vh_result* r, int* retry, const vh_plan* plan = nullptr);
struct VhHeavyRun { VhExec* x2 = nullptr; vh_result* r2 = nullptr; vh_plan p2{}; uint64_t cap = 0, ids_bound = 0; bool from_tuples = false; };
static int heavy_pass_enqueue(vh_result* r, VhHeavyRun& H) {
  vh_table* t = r->table;
  std::lock_guard<std::mutex> lk(t->mu);
  g_heavy.only = r->plan.heavy_mark; g_heavy.ids_bound = H.ids_bound; g_heavy.allow_mark = false;
  g_heavy.tuples_of = H.from_tuples ? r : nullptr; g_heavy.epoch = r->launch_epoch;
  const int rc = query_launch_locked(t, H.x2, &H.p2, &H.r2, H.cap, true, 0, false, false, nullptr, nullptr, false, 0, true);
  g_heavy = VhHeavyCtx{};
  if (rc) H.r2 = nullptr; else H.r2->exec = H.x2;
  return rc;
}
static void heavy_pass_drop(vh_result* r, VhHeavyRun& H) {
  if (H.r2) { H.r2->exec = nullptr; delete H.r2; H.r2 = nullptr; }
  if (H.x2) { (void)hipStreamSynchronize(H.x2->stream()); exec_release(r->table, H.x2); H.x2 = nullptr; }
  (void)hipGetLastError();
}
static void heavy_pass_launch(vh_result* r, const vh_plan* plan, uint64_t ranges, uint64_t tuples_bound, uint64_t whole_partitions, VhHeavyRun& H) {
  H.from_tuples = ranges == whole_partitions * 256ull && !test_env("VH_HEAVY_RESCAN");      // (every marked range lies in a partition that was left out whole: its tuples are in pool a)
  if (exec_acquire(r->table, &H.x2, false) != VH_OK) { H.x2 = nullptr; return; }      // (never waits for a context while holding one: two such queries would wait for each other — the caller re-plans instead)
  H.p2 = *plan;
  H.p2.flags = (H.p2.flags | VH_PLAN_FORCE_HASH | VH_PLAN_NO_JIT | VH_PLAN_NO_FAST | VH_PLAN_NO_LANES | VH_PLAN_NO_HPART) & ~(uint32_t)(VH_PLAN_FORCE_JIT | VH_PLAN_FORCE_HPART | VH_PLAN_FORCE_LANES);
  H.p2.groups_hint = 0;
  H.cap = 1ull << 16;
  while (H.cap < ranges * 8192ull && H.cap < (1ull << 30)) H.cap <<= 1;      // (a range holds ~500 groups when the keys are spread evenly; heavy_pass_finish regrows)
  H.ids_bound = 2 * tuples_bound + 1024;
  if (heavy_pass_enqueue(r, H)) heavy_pass_drop(r, H);
}
static void heavy_pass_finish(vh_result* r, VhHeavyRun& H) {
  for (int attempt = 0; H.r2; ++attempt) {
    int retry2 = 0;
    const int rc = result_finalize(H.r2, &retry2);
    if (!rc && !retry2) { H.r2->stream_quiet = true; H.x2 = nullptr; return; }      // (the result owns the context from here)
    H.r2->exec = nullptr; delete H.r2; H.r2 = nullptr;
    if (rc || retry2 != 1 || attempt >= 5) break;                               // anything but a full table: give up, the caller falls back
    H.cap <<= 2;
    if (heavy_pass_enqueue(r, H)) break;
  }
  heavy_pass_drop(r, H);
}

// returns VH_OK, or a positive "retry" request: 1 = grow hash table, 2 = fall back to hash
static int result_finalize(vh_result* r, int* retry, const vh_plan* plan) {
  VhExec* x = r->exec;   // staging buffers, scratch and events of this query's context
  const VhPlanDev& P = r->plan;
  hipStream_t st = x->stream();
  *retry = 0;
  // pinned staging buffer (two alternate per context: a zero-copy view stays readable after vh_result_free until the
  // second-next query); a re-planned attempt of the same query reuses its slot. Small results take the output region as it lies in the
  // scratch (one copy, or none: direct emission below); big ones are PACKED on their way out — a staging buffer for the rows that
  // exist, not for the rows the tables could hold (C5: 0.7 GB instead of 2.7 GB per slot; what does not fit the GPU's own NUMA node
  // is copied to at half the rate).
  const int slot = r->h_slot >= 0 ? r->h_slot : (x->h_out_next ^= 1);
  r->h_slot = slot;
  auto stage = [&](size_t bytes) -> int {
    if (x->h_out_bytes[slot] >= bytes) return VH_OK;
    if (x->h_out[slot]) HIP_TRY(hipHostFree(x->h_out[slot]));
    x->h_out[slot] = nullptr; x->h_out_bytes[slot] = 0;
    const size_t nb = std::max<size_t>(bytes + bytes / 8, 1 << 20);
    // coherent (fine-grained): the emission kernel writes small results straight into this buffer, and the host must see
    // them when the event behind the kernel has completed, whatever HIP_HOST_COHERENT says
    HIP_TRY(host_alloc_near_device((void**)&x->h_out[slot], nb, hipHostMallocCoherent));
    x->h_out_bytes[slot] = nb;
    return VH_OK;
  };
  const bool env_no_direct = knobs().no_direct_emit;
  const bool one_shot = r->out_region_bytes <= (8u << 20) && !r->topk_active && !r->hp_chunks;      // (a streamed result's rows are packed on their way out: never the region as a whole)
  const bool direct = one_shot && r->mode != VH_MODE_HASH && !env_no_direct && !r->device_rows;
  if (one_shot) { if (int src = stage(r->out_region_bytes)) return src; }
  // Small results of the dense paths are written by the emission kernel straight into that pinned host buffer
  // (posted PCIe writes, coalesced per column) and a one-wave kernel publishes the 512-byte header behind them: no
  // DMA-engine copy at the end of the query (its start-up costs 20-100 us, more than the 2 MB it moves).
  if (direct) {
    for (int i = 0; i < P.ngroup; ++i) r->d_out_key[i] = x->h_out[slot] + r->off_key[i];
    for (int j = 0; j < P.nmetric; ++j) r->d_out_state[j] = x->h_out[slot] + r->off_state[j];
  }
  VhEmitArgs A{};
  A.mode = r->mode == VH_MODE_DENSE_PART ? VH_MODE_DENSE_GLOBAL : r->mode; A.ngroup = P.ngroup; A.nmetric = P.nmetric; A.key_words = P.key_words;
  A.hstride = P.hrec_bytes ? P.hrec_bytes / 8u : (uint32_t)P.key_words;
  A.n = r->out_cap; A.present = P.present; A.present_carrier = (r->mode == VH_MODE_DENSE_GLOBAL || r->mode == VH_MODE_DENSE_PART) ? P.present_carrier : -1; A.hkeys = P.hkeys; A.htags = P.htags; A.counters = P.counters;
  A.out_count = r->d_out_count;
  A.n_dev = r->hpart ? P.counters + 1 : nullptr;         // hashed partitioning: entries [0, *n_dev) of the table are a compact list of group records
  for (int i = 0; i < P.ngroup; ++i) {
    A.glo[i] = P.g[i].lo; A.gextent[i] = P.g[i].extent; A.gstride[i] = P.g[i].stride;
    A.gtype[i] = P.g[i].type(); A.gkey_word[i] = P.g[i].key_word(); A.gkey_shift[i] = P.g[i].key_shift();
    A.out_key[i] = r->d_out_key[i];
  }
  for (int j = 0; j < P.nmetric; ++j) {
    A.state[j] = P.m[j].state; A.out_state[j] = r->d_out_state[j]; A.sop[j] = P.m[j].sop(); A.mtype[j] = (uint8_t)r->metric_elem[j];
    A.state_stride[j] = r->mode == VH_MODE_HASH && P.hrec_bytes ? P.hrec_bytes : (uint32_t)vh_sop_bytes(P.m[j].sop());
  }
  A.nhaving = r->nhaving;
  A.total_groups = P.counters + 6;
  for (int i = 0; i < r->nhaving; ++i) { A.hprog[i] = r->hprog[i]; A.htype[i] = r->htype[i]; }
  for (int i = 0; i < VH_MAX_HAVING_LITS; ++i) A.hlits[i] = r->hlits[i];
  // a small dense result: private copies, emission and header in ONE launch (small_tail_kernel); anything else merges first if nobody has
  const bool fused_tail = r->small_tail && direct && A.n <= VH_SMALL_TAIL_MAX;
  if (r->unmerged && !fused_tail) { if (int mrc = merge_copies_now(r, st)) return mrc; }
  if (fused_tail) {
    VhMergeArgs M;
    merge_args_of(r, &M);
    if (!r->unmerged) M.nxcd = 1;
    hipLaunchKernelGGL(small_tail_kernel, dim3(1), dim3(1024), 0, st, M, A, reinterpret_cast<unsigned long long*>(x->h_out[slot]),
                       reinterpret_cast<const unsigned long long*>(x->scratch + r->out_region_off));
    r->unmerged = false;
  }
  else if (r->hp_direct && r->nhaving == 0 && !r->topk_active) { /* hp_aggregate_kernel wrote the output columns and counted the rows */ }
  else if (A.n <= (4u << 20)) hipLaunchKernelGGL(emit_groups_kernel<2>, dim3((unsigned)((A.n + 256 * 2 - 1) / (256 * 2))), dim3(256), 0, st, A);
  else hipLaunchKernelGGL(emit_groups_kernel<16>, dim3((unsigned)((A.n + 256 * 16 - 1) / (256 * 16))), dim3(256), 0, st, A);
  HIP_TRY(hipGetLastError());
  if (r->topk_active) {
    // radix select of the top_k-th best sort key among the emitted rows (8 x 8 bits, no host round trip), then keep
    // every row that ties with or beats it. Row count is only known on the device: grids are sized by out_cap.
    VhTopkState init{};
    init.k_remaining = r->topk;
    HIP_TRY(hipMemcpyAsync(r->d_topk_state, &init, sizeof(init), hipMemcpyHostToDevice, st));
    const unsigned g1 = (unsigned)std::min<uint64_t>((r->out_cap + 255) / 256, (uint64_t)g_ctx.num_cu * 8);
    const void* src = r->topk_src_is_key ? r->d_out_key[r->topk_src] : r->d_out_state[r->topk_src];
    hipLaunchKernelGGL(topk_keys_kernel, dim3(g1), dim3(256), 0, st, src, r->topk_elem, (uint32_t)vh_elem_size(r->topk_elem),
                       r->topk_cls, r->topk_desc, (const unsigned long long*)r->d_out_count, r->d_topk_keys);
    for (int shift = 56; shift >= 0; shift -= 8) {
      hipLaunchKernelGGL(topk_hist_kernel, dim3(g1), dim3(256), 0, st, (const uint64_t*)r->d_topk_keys,
                         (const unsigned long long*)r->d_out_count, shift, r->d_topk_state);
      hipLaunchKernelGGL(topk_pick_kernel, dim3(1), dim3(64), 0, st, shift, r->d_topk_state);
    }
    VhTopkCompact C{};
    C.ncols = P.ngroup + P.nmetric;
    // formatter rounding ("%.15g" / "%g") can make nearby values compare equal in the reference: keep a margin
    C.slack = r->topk_cls == VH_TOPK_FLOAT ? (r->topk_elem == VH_F32 ? (256ull << 32) : 64ull) : 0ull;
    for (int i = 0; i < P.ngroup; ++i) { C.src[i] = r->d_out_key[i]; C.dst[i] = r->d_out_key2[i]; C.esize[i] = (uint32_t)vh_elem_size(P.g[i].type()); }
    for (int j = 0; j < P.nmetric; ++j) {
      C.src[P.ngroup + j] = r->d_out_state[j]; C.dst[P.ngroup + j] = r->d_out_state2[j];
      C.esize[P.ngroup + j] = (uint32_t)vh_elem_size(r->metric_elem[j]);
    }
    hipLaunchKernelGGL(topk_compact_kernel, dim3((unsigned)((r->out_cap + 255) / 256)), dim3(256), 0, st, C,
                       (const uint64_t*)r->d_topk_keys, (const unsigned long long*)r->d_out_count, (unsigned long long)r->topk, r->d_topk_state);
    HIP_TRY(hipGetLastError());
  }
  const char* D = x->scratch + r->out_region_off;
  // packed layout of a big result in the staging buffer: [512-byte header | key columns | state columns], each column `rows` long
  struct Packed { size_t key[VH_MAX_GROUP], state[VH_MAX_METRIC], bytes; };
  auto packed_for = [&](uint64_t rows) {
    Packed L{};
    size_t o = 512;
    for (int i = 0; i < P.ngroup; ++i) { L.key[i] = o; o += (std::max<uint64_t>(rows, 1) * vh_elem_size(P.g[i].type()) + 255) / 256 * 256; }
    for (int j = 0; j < P.nmetric; ++j) { L.state[j] = o; o += (std::max<uint64_t>(rows, 1) * vh_elem_size(r->metric_elem[j]) + 255) / 256 * 256; }
    L.bytes = o;
    return L;
  };
  auto copy_rows = [&](const Packed& L, uint64_t dst_row, uint64_t src_row, uint64_t n, bool second, hipStream_t cs) -> int {
    char* H = x->h_out[slot];
    uint64_t row_bytes = 0;
    for (int i = 0; i < P.ngroup; ++i) row_bytes += vh_elem_size(P.g[i].type());
    for (int j = 0; j < P.nmetric; ++j) row_bytes += vh_elem_size(r->metric_elem[j]);
    if (n * row_bytes >= ((uint64_t)1 << 20) && P.ngroup + P.nmetric <= VH_DELIVER_COLS && knobs().deliver_blocks > 0) {      // (deliver_kernel: why not the DMA engine)
      VhDeliverArgs A{};
      for (int i = 0; i < P.ngroup; ++i) {
        const size_t es = vh_elem_size(P.g[i].type());
        A.src[A.ncols] = (second ? (const char*)r->d_out_key2[i] : (const char*)r->d_out_key[i]) + src_row * es; A.dst[A.ncols] = H + L.key[i] + dst_row * es; A.bytes[A.ncols++] = n * es;
      }
      for (int j = 0; j < P.nmetric; ++j) {
        const size_t es = vh_elem_size(r->metric_elem[j]);
        A.src[A.ncols] = (second ? (const char*)r->d_out_state2[j] : (const char*)r->d_out_state[j]) + src_row * es; A.dst[A.ncols] = H + L.state[j] + dst_row * es; A.bytes[A.ncols++] = n * es;
      }
      hipLaunchKernelGGL(deliver_kernel, dim3((unsigned)knobs().deliver_blocks), dim3(256), 0, cs, A);
      HIP_TRY(hipGetLastError());
      return VH_OK;
    }
    for (int i = 0; i < P.ngroup; ++i) {
      const size_t es = vh_elem_size(P.g[i].type());
      HIP_TRY(hipMemcpyAsync(H + L.key[i] + dst_row * es, (second ? (const char*)r->d_out_key2[i] : (const char*)r->d_out_key[i]) + src_row * es, n * es, hipMemcpyDeviceToHost, cs));
    }
    for (int j = 0; j < P.nmetric; ++j) {
      const size_t es = vh_elem_size(r->metric_elem[j]);
      HIP_TRY(hipMemcpyAsync(H + L.state[j] + dst_row * es, (second ? (const char*)r->d_out_state2[j] : (const char*)r->d_out_state[j]) + src_row * es, n * es, hipMemcpyDeviceToHost, cs));
    }
    return VH_OK;
  };
  // streamed result: every finished chunk's rows go out on the copy stream, packed one chunk behind the other, while the next chunks run.
  // The staging buffer is sized when the first chunk's count is in (the mixed key deals the groups evenly: eight times that, and a bit);
  // should the rest not fit after all, everything is copied once more when all counts are known.
  uint64_t streamed = 0;
  Packed L{};
  if (r->hp_chunks) {
    uint64_t cnt[VH_HP_CHUNKS] = {}, rows_cap = 0;
    bool redo = false;
    for (int c = 0; c < r->hp_chunks; ++c) {
      HIP_TRY(wait_event_spinning(x->ev_chunk[c]));
      cnt[c] = std::min<uint64_t>(reinterpret_cast<volatile unsigned long long*>(x->h_chunk)[c], r->hp_chunk_rows);      // (more: the region overflowed, the attempt is void — flagged in the header)
      if (c == 0) {
        rows_cap = cnt[0] * (uint64_t)r->hp_chunks + cnt[0] / 4 + 65536;
        L = packed_for(rows_cap);
        if (int src = stage(L.bytes)) return src;
      }
      if (streamed + cnt[c] > rows_cap) redo = true;
      if (cnt[c] && !redo) { if (int crc = copy_rows(L, streamed, (uint64_t)c * r->hp_chunk_rows, cnt[c], false, x->copy)) return crc; }
      streamed += cnt[c];
    }
    if (redo) {
      HIP_TRY(hipStreamSynchronize(x->copy));
      L = packed_for(streamed);
      if (int src = stage(L.bytes)) return src;
      uint64_t at = 0;
      for (int c = 0; c < r->hp_chunks; ++c) { if (cnt[c]) { if (int crc = copy_rows(L, at, (uint64_t)c * r->hp_chunk_rows, cnt[c], false, x->copy)) return crc; } at += cnt[c]; }
    }
  }
  VhTopkState tk{};
  if (r->topk_active) HIP_TRY(hipMemcpyAsync(&tk, r->d_topk_state, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  // small results: counters, group count and every output array come back in ONE copy + ONE sync; big ones: the header first
  unsigned long long* const head = one_shot ? reinterpret_cast<unsigned long long*>(x->h_out[slot]) : x->h_counters + 16;
  if (direct) {
    if (!fused_tail) hipLaunchKernelGGL(publish_header_kernel, dim3(1), dim3(64), 0, st, head, reinterpret_cast<const unsigned long long*>(D));
    HIP_TRY(hipGetLastError());
  } else {
    HIP_TRY(hipMemcpyAsync(head, D, one_shot ? r->out_region_bytes : 512, hipMemcpyDeviceToHost, st));
  }
  HIP_TRY(hipEventRecord(x->ev[3], st));
  HIP_TRY(wait_event_spinning(x->ev[3]));
  const unsigned long long* hc = head;
  const unsigned long long err = hc[2];
  if (r->hp_chunks) HIP_TRY(hipStreamSynchronize(x->copy));      // (also when the attempt is void: the next one rewrites what the copies read)
  if (err & VH_ERR_HP_WIDE) { *retry = 6; return VH_OK; }        // packed tuples met a value beyond the recorded min / max: the plain hash table
  if (err & VH_ERR_HPART_FULL) { *retry = 4; return VH_OK; }
  if (err & VH_ERR_HASH_FULL) { *retry = 1; return VH_OK; }
  if (err & VH_ERR_PART_FULL) { r->info.passed_recs = hc[0]; *retry = 3; return VH_OK; }   // phase 1 ran to the end: the survivors are counted
  if (err & VH_ERR_RANGE) { *retry = 2; return VH_OK; }
  uint64_t ng = r->hp_chunks ? streamed : hc[32];                                 // rows emitted (after HAVING): the word at byte 256
  r->info.ngroups = r->nhaving ? hc[6] : ng;                                     // agg_map.size()
  if (r->topk_active) ng = tk.out_count;                                          // rows kept by the top-N superset
  r->info.returned_groups = ng;
  r->ngroups_host = ng;
  r->info.passed_recs = hc[0];
  // heavy ranges of a hashed partitioning (hc[11] of them, at most hc[12] tuples): their groups come from a second pass through the plain hash
  // organisation and are appended behind the rows the ranges' kernel wrote
  std::unique_ptr<vh_result> heavy;
  uint64_t n2 = 0;
  VhHeavyRun HR;
  struct HeavyGuard { vh_result* r; VhHeavyRun& H; ~HeavyGuard() { if (H.x2) heavy_pass_drop(r, H); } } heavy_guard{r, HR};      // (an early return leaves no context behind)
  const bool heavy_wanted = r->hpart && P.heavy_mark && hc[11];
  // a big result in one piece: its rows cross PCIe WHILE the pass scans — the staging buffer then has room for the rows the pass can add at most
  // (a group per tuple of the marked ranges); beyond a quarter of a gigabyte of such slack the pass is waited for first
  uint64_t row_bytes = 0;
  for (int i = 0; i < P.ngroup; ++i) row_bytes += vh_elem_size(P.g[i].type());
  for (int j = 0; j < P.nmetric; ++j) row_bytes += vh_elem_size(r->metric_elem[j]);
  const bool overlap = heavy_wanted && plan && !one_shot && !r->hp_chunks && hc[12] * row_bytes <= (256ull << 20) && !test_env("VH_NO_HEAVY_OVERLAP");
  if (heavy_wanted && plan && !overlap) heavy_pass_launch(r, plan, hc[11], hc[12], hc[13], HR);
  auto heavy_join = [&]() -> bool {             // false: no second pass to be had — the plain table for everything, as before
    heavy_pass_finish(r, HR);
    if (!HR.r2 || (one_shot && ng + HR.r2->ngroups_host > r->out_cap)) { delete HR.r2; HR.r2 = nullptr; *retry = 4; r->plan.hp_passes = 64; return false; }
    heavy.reset(HR.r2);
    n2 = HR.r2->ngroups_host;
    r->info.retries += 1;       // (counted like an attempt: the caller sees that the query took more than one pass)
    if (r->kernel.find(HR.r2->kernel) == std::string::npos) r->kernel += " + " + HR.r2->kernel;      // (vh_result_kernel: what the second pass ran)
    return true;
  };
  auto append_heavy = [&](const size_t* key_off, const size_t* state_off) {      // the second pass's rows behind row ng of this result's host columns
    char* H = x->h_out[slot];
    for (int i = 0; i < P.ngroup; ++i) { const size_t es = vh_elem_size(P.g[i].type()); memcpy(H + key_off[i] + ng * es, heavy->h_base + heavy->off_key[i], n2 * es); }
    for (int j = 0; j < P.nmetric; ++j) { const size_t es = vh_elem_size(r->metric_elem[j]); memcpy(H + state_off[j] + ng * es, heavy->h_base + heavy->off_state[j], n2 * es); }
  };
  if (!one_shot) {
    if (!r->hp_chunks) {           // a big result in one piece: the staging buffer is sized for the rows there are
      if (heavy_wanted && !overlap) { if (!heavy_join()) return VH_OK; }
      L = packed_for(ng + (overlap ? hc[12] : n2));
      if (int src = stage(L.bytes)) return src;
      // (the pass is enqueued BEFORE the copy's kernel: its clears and its scan are under way when the 64 blocks that push rows over PCIe start)
      const auto h0 = std::chrono::steady_clock::now();
      if (overlap) heavy_pass_launch(r, plan, hc[11], hc[12], hc[13], HR);
      if (ng) { if (int crc = copy_rows(L, 0, 0, ng, r->topk_active, st)) return crc; }
      HIP_TRY(hipEventRecord(x->ev[3], st));
      if (overlap) {
        const bool joined = heavy_join();
        const auto h1 = std::chrono::steady_clock::now();
        if (!joined) { (void)wait_event_spinning(x->ev[3]); return VH_OK; }
        HIP_TRY(wait_event_spinning(x->ev[3]));
        if (knobs().times) {
          const auto ms = [](auto a, auto b) { return (double)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count() / 1e3; };
          fprintf(stderr, "vh heavy pass: done %.3f ms after its launch, the copy of the first pass's rows %.3f ms after that (ranges %llu, tuples <= %llu, rows %llu)\n", ms(h0, h1),
                  ms(h1, std::chrono::steady_clock::now()), hc[11], hc[12], (unsigned long long)n2);
        }
      }
      HIP_TRY(wait_event_spinning(x->ev[3]));
      if (n2) append_heavy(L.key, L.state);
    } else if (heavy_wanted) { if (!heavy_join()) return VH_OK; }
    memcpy(x->h_out[slot], head, 512);
    for (int i = 0; i < P.ngroup; ++i) r->off_key[i] = L.key[i];      // (the host view: where vh_result_view finds the columns)
    for (int j = 0; j < P.nmetric; ++j) r->off_state[j] = L.state[j];
  } else if (heavy_wanted) {
    if (!heavy_join()) return VH_OK;
    if (n2) append_heavy(r->off_key, r->off_state);
  }
  if (n2) { ng += n2; r->info.ngroups += n2; r->info.returned_groups = ng; r->ngroups_host = ng; }
  heavy.reset();
  r->h_base = x->h_out[slot];
  float ms = 0;
  (void)hipEventElapsedTime(&ms, x->ev[1], x->ev[2]); r->info.scan_kernel_ms = ms;
  (void)hipEventElapsedTime(&ms, x->ev[0], x->ev[3]); r->info.total_ms = ms;
  if (knobs().times) {   // where a query's device time goes: setup (clears, uploads) | scan | emission + read-back
    float a = 0, b = 0;
    (void)hipEventElapsedTime(&a, x->ev[0], x->ev[1]); (void)hipEventElapsedTime(&b, x->ev[2], x->ev[3]);
    fprintf(stderr, "vh times: setup %.3f ms, scan %.3f ms, emit+readback %.3f ms (groups %llu, returned %llu)\n", a, r->info.scan_kernel_ms, b,
            (unsigned long long)r->info.ngroups, (unsigned long long)r->info.returned_groups);
  }
  r->finalized = true;
  return VH_OK;
}


extern "C" int vh_query_launch(vh_table* t, const vh_plan* plan, vh_result** out) {
  if (!t || !plan || !out) return vh_fail(VH_E_INVALID, "null argument");
  VH_ENTER();
  VhExec* x = nullptr;
  if (int rc = exec_acquire(t, &x)) return rc;
  vh_result* r = nullptr;
  int rc;
  { std::lock_guard<std::mutex> lk(t->mu); rc = query_launch_locked(t, x, plan, &r, 0, false); }
  if (rc) { (void)hipStreamSynchronize(x->stream()); exec_release(t, x); return rc; }
  r->exec = x;
  *out = r;
  return VH_OK;
}

extern "C" int vh_result_finalize(vh_result* r) {
  if (!r) return vh_fail(VH_E_INVALID, "null result");
  if (r->finalized) return VH_OK;
  VH_ENTER();
  int retry = 0;
  int rc = result_finalize(r, &retry);
  if (rc) return rc;
  if (retry) return vh_fail(VH_E_RANGE, "partial result needs a re-plan (code %d); use vh_query_agg", retry);
  return VH_OK;
}

// One attempt's verdict -> the overrides of the next one. Shared by vh_query_agg and the sharded form.
struct VhReplan { uint64_t cap_override = 0, part_override = 0; bool force_hash = false, no_part = false; uint32_t hp_passes = 0; bool no_hpart = false; };
static void replan_after(vh_table* t, vh_result* r, int retry, VhReplan* rp) {
  if (retry == 1) {
    // table too small. The number of groups is bounded by the number of surviving rows: estimate those
    // once with the selectivity probe and size for them, instead of quadrupling blindly
    uint64_t next = (r->plan.hmask + 1) * 4;
    if (!rp->cap_override) {
      double sel = 1.0;
      if (r->info.reserved & 1) { std::lock_guard<std::mutex> lk(t->mu); (void)estimate_selectivity(t, r->exec, r->plan, r->h_prog, r->h_lits, r->plan.nseg, &sel); }
      uint64_t survivors = (uint64_t)((double)r->info.scanned_recs * std::min(1.0, sel * 1.1)) + 1024;
      uint64_t sized = 1;
      while (sized < survivors * 2) sized <<= 1;
      next = std::max(next, sized);
    }
    rp->cap_override = next;
  }
  else if (retry == 6) { if (r->hpart) rp->no_hpart = true; else rp->no_part = true; }      // a value beyond its column's recorded range in a packed tuple
  else if (retry == 4) {                                     // hashed partitioning: a range held more groups (or ids) than its passes' LDS tables take
    // ... skewed beyond help (a group with more ids than any number of passes fits into a range's LDS set): the plain hash table — and the table
    // remembers the shape, like groups_seen for hash sizing: its next queries start there instead of paying for the void attempts every time
    if (r->plan.hp_passes >= 64) { rp->no_hpart = true; std::lock_guard<std::mutex> lk(t->mu); t->hpart_hopeless.insert(r->group_sig); }
    else rp->hp_passes = (uint32_t)r->plan.hp_passes * 4;
    if (rp->hp_passes > 64) rp->hp_passes = 64;
  }
  else if (retry == 3 && r->hpart) {                         // hashed partitioning ran out of tuple extents: size for the survivors it counted, then give up
    // (the ring writer's pools hold any skew between their streams — positional extents + a shared overflow region as big as the estimate —, so
    // running out means more survivors than estimated: the re-run is sized for what this attempt counted, through the same writer)
    if (rp->part_override) rp->no_hpart = true;
    else rp->part_override = std::max<uint64_t>(r->info.passed_recs + r->info.passed_recs / 16 + 1024, 1ull << 16);
  }
  else if (retry == 3) {                                     // tuple extents exhausted: more room, then give up on partitioning
    // the attempt counted its survivors even though it dropped their tuples: the next one is sized for exactly that many
    const uint64_t had = (uint64_t)r->plan.max_extents * r->plan.ext_tuples;
    // the piecewise writers' positional chunks (VhPlanDev::ext_waves: tuples of three or more words, the pre-built kernels) ran out with room to
    // spare: some waves met far more survivors than others. Remembered for the shape, like groups_seen for hash sizing: its next queries start on the
    // shared cursor. (The ring writer needs no such memory: its streams overflow into the pool's shared region inside the kernel.)
    if (r->plan.ext_waves && r->info.passed_recs + r->info.passed_recs / 16 <= had) { std::lock_guard<std::mutex> lk(t->mu); t->part_clustered.insert(r->group_sig); }
    if (rp->part_override && had >= r->info.scanned_recs) rp->no_part = true;
    else rp->part_override = std::max<uint64_t>(std::max<uint64_t>(rp->part_override * 2, r->info.passed_recs + r->info.passed_recs / 16), 1ull << 16);
  }
  else rp->force_hash = true;                                // a digit left its planned range
}

// More metrics than one pass carries (VH_MAX_METRIC states per group in the kernel arguments): several passes over
// the same snapshot, each with a slice of the metrics, joined on the group key. The reference has no such limit
// (AggTuple::Metrics is a generated struct of any width, store.cc:31-169).
static int query_agg_multipass(vh_table* t, const vh_plan* plan, vh_result** out) {
  if (plan->nhaving || plan->top_k)
    return vh_fail(VH_E_UNSUPPORTED, "%d metrics take several passes: apply HAVING / top-N to the returned groups", plan->nmetrics);
  const int per = VH_MAX_METRIC - 4;
  std::vector<std::unique_ptr<vh_result>> parts;
  int count_col = -1;
  for (int j = 0; j < plan->nmetrics; ++j)
    if (plan->metrics[j] >= 0 && (size_t)plan->metrics[j] < t->cols.size() && t->cols[plan->metrics[j]].kind == VH_METRIC_COUNT) count_col = plan->metrics[j];
  std::vector<int> part_user;                                  // metrics of each pass that belong to the caller's list
  for (int off = 0; off < plan->nmetrics; off += per) {
    const int nm = std::min(per, plan->nmetrics - off);
    std::vector<int32_t> cm(plan->metrics + off, plan->metrics + off + nm);
    bool avg = false, cnt = false;
    for (int32_t c : cm) if (c >= 0 && (size_t)c < t->cols.size()) { avg |= t->cols[c].kind == VH_METRIC_AVG; cnt |= t->cols[c].kind == VH_METRIC_COUNT; }
    if (avg && !cnt && count_col >= 0) cm.push_back(count_col);   // an AVG slice still divides by the query's COUNT (scan.cc:239-241): it rides along
    vh_plan cp = *plan;
    cp.metrics = cm.data(); cp.nmetrics = (int32_t)cm.size();
    vh_result* r = nullptr;
    if (int rc = vh_query_agg(t, &cp, &r)) return rc;
    parts.emplace_back(r);
    part_user.push_back(nm);
  }
  vh_result* base = parts[0].get();
  const uint64_t n = base->ngroups_host;
  const int nk = base->plan.ngroup;
  std::unique_ptr<vh_result> rf(new vh_result());
  rf->table = t; rf->info = base->info; rf->mode = base->mode; rf->kernel = base->kernel;
  rf->plan.ngroup = nk; rf->plan.key_words = base->plan.key_words;
  for (int i = 0; i < nk; ++i) rf->plan.g[i] = base->plan.g[i];
  rf->group_elem = base->group_elem;
  auto key_of = [&](const vh_result* r, uint64_t row) {
    std::string k;
    for (int i = 0; i < nk; ++i) { const int es = vh_elem_size(r->plan.g[i].type()); k.append(r->h_base + r->off_key[i] + row * es, es); }
    return k;
  };
  std::unordered_map<std::string, uint64_t> where;
  if (parts.size() > 1) { where.reserve(n * 2); for (uint64_t row = 0; row < n; ++row) where.emplace(key_of(base, row), row); }
  // layout of the joined result: keys, then every pass's user metrics in plan order, then the hidden count (if the plan has one)
  const vh_result* hidden_from = nullptr;
  bool any_count = false;
  for (int j = 0; j < plan->nmetrics; ++j) any_count |= plan->metrics[j] >= 0 && t->cols[plan->metrics[j]].kind == VH_METRIC_COUNT;
  for (auto& pr : parts) if (pr->info.has_hidden_count && !any_count && !hidden_from) hidden_from = pr.get();
  size_t bytes = 0;
  for (int i = 0; i < nk; ++i) { rf->off_key[i] = bytes; bytes += (std::max<uint64_t>(n, 1) * vh_elem_size(base->plan.g[i].type()) + 255) / 256 * 256; }
  std::vector<std::pair<const vh_result*, int>> src;      // joined device-metric index -> (pass, its device metric)
  for (size_t k = 0; k < parts.size(); ++k) for (int j = 0; j < part_user[k]; ++j) src.push_back({parts[k].get(), parts[k]->user_metric[j]});
  if (hidden_from) src.push_back({hidden_from, hidden_from->plan.nmetric - 1});
  if (src.size() > 4096) return vh_fail(VH_E_UNSUPPORTED, "too many metrics");
  std::vector<size_t> off_state(src.size());
  for (size_t u = 0; u < src.size(); ++u) {
    rf->metric_elem.push_back(src[u].first->metric_elem[src[u].second]);
    off_state[u] = bytes; bytes += (std::max<uint64_t>(n, 1) * vh_elem_size(rf->metric_elem.back()) + 255) / 256 * 256;
  }
  HIP_TRY(host_alloc_near_device((void**)&rf->h_own, bytes, hipHostMallocDefault));
  for (int i = 0; i < nk; ++i) if (n) memcpy(rf->h_own + rf->off_key[i], base->h_base + base->off_key[i], n * vh_elem_size(base->plan.g[i].type()));
  for (size_t u = 0; u < src.size(); ++u) {
    const vh_result* pr = src[u].first;
    const int es = vh_elem_size(rf->metric_elem[u]);
    const char* from = pr->h_base + pr->off_state[src[u].second];
    char* to = rf->h_own + off_state[u];
    if (pr == base) { if (n) memcpy(to, from, n * es); continue; }
    if (pr->ngroups_host != n) return vh_fail(VH_E_DEVICE, "passes of one query returned %llu and %llu groups", (unsigned long long)n, (unsigned long long)pr->ngroups_host);
    for (uint64_t row = 0; row < n; ++row) {
      auto it = where.find(key_of(pr, row));
      if (it == where.end()) return vh_fail(VH_E_DEVICE, "passes of one query returned different groups");
      memcpy(to + it->second * es, from + row * es, es);
    }
  }
  rf->wide_off_state = off_state;
  for (int j = 0; j < plan->nmetrics; ++j) rf->user_metric.push_back(j);
  rf->info.nmetrics = plan->nmetrics; rf->info.has_hidden_count = hidden_from ? 1 : 0;
  rf->plan.nmetric = (int32_t)std::min<size_t>(src.size(), VH_MAX_METRIC);
  rf->h_base = rf->h_own; rf->ngroups_host = n; rf->finalized = true;
  *out = rf.release();
  return VH_OK;
}

static int place_layouts(vh_table* t, const vh_plan* plan, vh_result_info* info_out, double budget_ms);
static bool placing_now();
extern "C" int vh_query_agg(vh_table* t, const vh_plan* plan, vh_result** out) {
  if (!t || !plan || !out) return vh_fail(VH_E_INVALID, "null argument");
  VH_ENTER();
  if (plan->nmetrics > VH_MAX_METRIC - 1 && plan->metrics) return query_agg_multipass(t, plan, out);
  VhExec* x = nullptr;
  if (int rc = exec_acquire(t, &x, !placing_now())) return rc;      // (a placement's own queries never wait for a context: its caller may hold the last one — no placement then)
  VhReplan rp;
  int rc = VH_OK;
  for (uint32_t attempt = 0; attempt < 12; ++attempt) {
    vh_result* r = nullptr;
    const auto h0 = std::chrono::steady_clock::now();
    // planned and launched under the table lock; the wait for the device and the read-back happen outside it, so
    // queries of other threads on this table run meanwhile (each on its own context)
    {
      std::lock_guard<std::mutex> lk(t->mu);
      g_heavy.allow_mark = true;        // (this caller can run the second pass over heavy ranges: it holds the plan when the verdict is in)
      rc = query_launch_locked(t, x, plan, &r, rp.cap_override, rp.force_hash, rp.part_override, rp.no_part, false, nullptr, nullptr, false, rp.hp_passes, rp.no_hpart);
      g_heavy = VhHeavyCtx{};
    }
    if (rc) break;
    r->exec = x;
    int retry = 0;
    const auto h1 = std::chrono::steady_clock::now();
    rc = result_finalize(r, &retry, plan);
    if (knobs().times) fprintf(stderr, "vh host: plan + enqueue %.1f us, finalize (enqueue tail + wait + read-back) %.1f us\n", std::chrono::duration<double, std::micro>(h1 - h0).count(),
                               std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h1).count());
    if (rc) { r->exec = nullptr; delete r; break; }
    if (!retry) {
      r->info.retries += attempt;                              // (+ 1 where a second pass took the heavy ranges of a hashed partitioning)
      r->stream_quiet = true;                                  // (result_finalize waited for everything it enqueued)
      if (r->mode == VH_MODE_HASH) { std::lock_guard<std::mutex> lk(t->mu); t->groups_seen[r->group_sig] = r->info.ngroups; }
      *out = r;
      return VH_OK;
    }
    replan_after(t, r, retry, &rp);
    r->exec = nullptr;                                       // the next attempt runs on the same context
    delete r;
    rc = vh_fail(VH_E_NOMEM, "aggregate table kept overflowing");
  }
  (void)hipStreamSynchronize(x->stream());
  exec_release(t, x);
  return rc;
}

// A place for the derived layouts a plan reads (vhh_derived.h, derived_move): up to `prepare_place` other places tried, each measured with three
// queries, the fastest kept. Only where it can matter (a scan of 0.3 ms and more through a projection or predicate planes) and while the
// candidates fit the free device memory next to a quarter of the device. Called by vh_table_prepare only: doing the same from inside the first
// query that finds unprepared layouts in use was built and taken out again — its queries run on a second execution context while the caller's
// result holds the first (seconds of pinned staging memory behind a big result), and a fresh allocation of a few GB takes the driver 2 ms or
// 400 depending on what the memory was last used for: first queries of 3.4 s were seen. A setup call can afford that; a query cannot.
static thread_local bool g_placing = false;
static bool placing_now() { return g_placing; }
static int place_layouts(vh_table* t, const vh_plan* plan, vh_result_info* info_out, double budget_ms) {
  const auto t_begin = std::chrono::steady_clock::now();
  struct Guard { Guard() { g_placing = true; } ~Guard() { g_placing = false; } } guard;
  const int cand = knobs().prepare_place;
  auto measure = [&](float* ms) -> int {
    *ms = 1e30f;
    for (int i = 0; i < 3; ++i) {
      vh_result* r = nullptr;
      if (int rc = vh_query_agg(t, plan, &r)) return rc;
      *ms = std::min(*ms, r->info.scan_kernel_ms);
      if (info_out) *info_out = r->info;
      vh_result_free(r);
    }
    return VH_OK;
  };
  if (cand > 0) {
    float best = 0;
    if (int rc = measure(&best)) return rc;
    std::vector<char*> held;
    for (int k = 0; k < cand && best >= 0.3f; ++k) {
      // (a fresh allocation of a few GB takes the driver 2 ms or 400, depending on what the memory was last used for: the candidates stop when the
      // budget is spent — a second inside vh_table_prepare, 0.3 s inside a query that places layouts nobody prepared)
      if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count() > budget_ms) break;
      std::vector<VhMoved> moved;
      // (both layouts, then the planes alone, then the projections alone, and again: what decides is how the two lie to each other as much as
      // where either lies; a spacer of 1-3 GB in front, because neighbouring allocations tend to behave alike)
      const uint32_t which = k % 3 == 0 ? 3u : k % 3 == 1 ? 2u : 1u;
      const auto tk0 = std::chrono::steady_clock::now();
      {
        std::lock_guard<std::mutex> lk(t->mu);
        size_t free_b = 0, total_b = 0;
        const size_t spacer = (size_t)(1 + k % 3) << 30;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < derived_bytes(t, which) + spacer + total_b / 4) break;
        char* sp = nullptr;
        if (hipMalloc(&sp, spacer) == hipSuccess) held.push_back(sp); else (void)hipGetLastError();
        if (int rc = derived_move(t, which, &moved)) { derived_settle(t, moved, false, &held); for (char* p : held) (void)hipFree(p); return rc; }
      }
      if (knobs().times) fprintf(stderr, "vh prepare: candidate %d: spacer + allocations + copies in %.1f ms\n", k, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tk0).count());
      if (moved.empty()) break;
      float ms = 0;
      const auto tm0 = std::chrono::steady_clock::now();
      const int mrc = measure(&ms);
      if (knobs().times) fprintf(stderr, "vh prepare: candidate %d: three queries in %.1f ms\n", k, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tm0).count());
      const bool keep = !mrc && ms < best * 0.985f;
      if (knobs().times) fprintf(stderr, "vh prepare: %s at another place: %.4f ms against %.4f ms (%s)\n", which == 3u ? "projections and predicate planes" : which == 2u ? "predicate planes" : "projections", ms, best, keep ? "kept" : "given back");
      { std::lock_guard<std::mutex> lk(t->mu); derived_settle(t, moved, keep, &held); }
      if (mrc) { for (char* p : held) (void)hipFree(p); return mrc; }
      if (keep) best = ms;
    }
    if (!held.empty()) {
      const auto tf0 = std::chrono::steady_clock::now();
      std::lock_guard<std::mutex> lk(t->mu);
      table_quiesce(t);
      for (char* p : held) (void)hipFree(p);
      if (knobs().times) fprintf(stderr, "vh prepare: %zu buffers released in %.1f ms\n", held.size(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tf0).count());
    }
  }
  return VH_OK;
}

// First-use costs paid up front: the scan kernel compiled for the plan's shape (1-2 s of hipRTC, or milliseconds from the disk
// cache), the payload projection and the narrow predicate copies a selective query reads (built at once instead of after VH_AUTO_PACK /
// VH_AUTO_NARROW uses). (Rounds 3-5 also searched for a good place for a big tuple pool here; see vhh_place.h for why that is gone.) The reference's analogue is Compiler::Compile running when a
// query shape is first seen (src/codegen/compiler.cc:97-144, QueryStats::compile_time); a caller that knows its hot shapes at table-load
// time runs them through here. The plan is executed (up to three times: a narrow copy, then a projection can appear) and
// the last attempt's info is returned, so the caller sees what a steady-state query of this shape will run on.
extern "C" int vh_table_prepare(vh_table* t, const vh_plan* plan, vh_result_info* info_out) {
  if (!t || !plan) return vh_fail(VH_E_INVALID, "null argument");
  struct Guard { Guard() { g_preparing = true; } ~Guard() { g_preparing = false; } } guard;
  uint32_t last = ~0u;
  for (int round = 0; round < 3; ++round) {
    vh_result* r = nullptr;
    if (int rc = vh_query_agg(t, plan, &r)) return rc;
    const uint32_t now = r->info.reserved;
    if (info_out) *info_out = r->info;
    vh_result_free(r);
    if (now == last) break;
    last = now;
  }
  if (last & (8u | 2048u)) return place_layouts(t, plan, info_out, 1000.0);
  return VH_OK;
}


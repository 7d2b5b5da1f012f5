// vhh_select.h — host side of libviya_hip, part of viya_hip.hip's translation unit (included there, in order; not a stand-alone header):
// select (ordered row emission): vh_query_select.
// ----------------------------------------------------------------- select (ordered row emission)
struct vh_rows {
  vh_rows_info info{};
  std::vector<int> elem;
  std::vector<size_t> off;
  char* d_out = nullptr;
  char* h_out = nullptr;
  ~vh_rows() { if (d_out) (void)hipFree(d_out); if (h_out) (void)hipHostFree(h_out); }
};

extern "C" void vh_rows_free(vh_rows* r) { if (r) { VH_ENTER(); delete r; } }

extern "C" int vh_rows_get_info(vh_rows* r, vh_rows_info* info) {
  if (!r || !info) return vh_fail(VH_E_INVALID, "null argument");
  *info = r->info;
  return VH_OK;
}

extern "C" int vh_rows_view(vh_rows* r, const void** cols) {
  if (!r || !cols) return vh_fail(VH_E_INVALID, "null argument");
  for (size_t c = 0; c < r->elem.size(); ++c) cols[c] = r->h_out ? r->h_out + r->off[c] : nullptr;
  return VH_OK;
}

extern "C" int vh_query_select(vh_table* t, const vh_select_plan* sp, vh_rows** out) {
  if (!t || !sp || !out) return vh_fail(VH_E_INVALID, "null argument");
  if (sp->ncols < 0 || sp->ncols > VH_MAX_SELECT) return vh_fail(VH_E_UNSUPPORTED, "%d selected columns (max %d)", sp->ncols, VH_MAX_SELECT);
  VH_ENTER();
  VhExec* x = nullptr;
  if (int rc = exec_acquire(t, &x)) return rc;
  struct Release { vh_table* t; VhExec* x; ~Release() { (void)hipStreamSynchronize(x->stream()); exec_release(t, x); } } release{t, x};
  // select launches twice with a host decision in between: it keeps the table lock throughout (not the hot path)
  std::lock_guard<std::mutex> lk(t->mu);
  const int ncols_t = (int)t->cols.size();
  for (int c = 0; c < sp->ncols; ++c)
    if (sp->cols[c] < 0 || sp->cols[c] >= ncols_t) return vh_fail(VH_E_INVALID, "selected column %d: bad column %d", c, sp->cols[c]);
  vh_plan p{};
  p.filter = sp->filter; p.nfilter = sp->nfilter; p.lits = sp->lits; p.nlits = sp->nlits;
  p.seg_rows = sp->seg_rows; p.nseg = sp->nseg; p.flags = sp->flags;
  vh_result* pr = nullptr;
  int rc = query_launch_locked(t, x, &p, &pr, 0, false, 0, false, true);
  if (rc) return rc;
  std::unique_ptr<vh_result> plan_holder(pr);
  VhPlanDev P = pr->plan;
  const uint32_t nseg = P.nseg;
  hipStream_t st = x->stream();
  if (int frc = derived_fence(t, st)) return frc;
  std::unique_ptr<vh_rows> rows(new vh_rows());
  rows->info.scanned_recs = pr->info.scanned_recs;
  rows->info.scanned_segments = pr->info.scanned_segments;
  for (int c = 0; c < sp->ncols; ++c) rows->elem.push_back(is_bitset_elem(t->cols[sp->cols[c]].elem) ? VH_U64 : t->cols[sp->cols[c]].elem);
  rows->off.assign(sp->ncols, 0);
  if (nseg == 0) { *out = rows.release(); return VH_OK; }

  const uint32_t cps = (uint32_t)((t->padded_rows + VH_WAVE_STEP_ROWS - 1) / VH_WAVE_STEP_ROWS);
  const uint64_t nchunks = (uint64_t)nseg * cps;
  ScratchPlan spn;
  const size_t o_ctr = spn.take(256), o_segrows = spn.take(pr->plan_words * 4), o_counts = spn.take(nchunks * 4),
               o_totals = spn.take((size_t)nseg * 8), o_win = spn.take((size_t)nseg * sizeof(VhSelectWindow)),
               o_sel = spn.take(sizeof(VhSelectDev));
  size_t o_bs[VH_MAX_SELECT] = {}, o_fbs[VH_MAX_BITSET] = {};
  for (int c = 0; c < sp->ncols; ++c) if (is_bitset_elem(t->cols[sp->cols[c]].elem)) o_bs[c] = spn.take((size_t)nseg * 8);
  for (size_t k = 0; k < pr->filter_bitset_cols.size(); ++k) o_fbs[k] = spn.take((size_t)nseg * 8);
  rc = ensure_scratch(x, spn.off);
  if (rc) return rc;
  char* S = x->scratch;
  HIP_TRY(hipEventRecord(x->ev[0], st));
  HIP_TRY(hipMemsetAsync(S + o_ctr, 0, 256, st));
  HIP_TRY(hipMemcpyAsync(S + o_segrows, x->h_segrows, pr->plan_words * 4, hipMemcpyHostToDevice, st));
  P.prog = reinterpret_cast<const VhProgOp*>(S + o_segrows + pr->seg_words * 4);
  P.lits = reinterpret_cast<const uint64_t*>(S + o_segrows + pr->seg_words * 4 + pr->h_prog.size() * sizeof(VhProgOp));
  for (size_t k = 0; k < pr->filter_bitset_cols.size(); ++k) {     // bitset metrics in the filter: per-segment CSR offsets
    const VhColumn& fc = t->cols[pr->filter_bitset_cols[k]];
    for (uint32_t s = 0; s < nseg; ++s)
      if (x->h_segrows[s] && !fc.bs_offsets[s]) return vh_fail(VH_E_INVALID, "bitset column %d of segment %u was never synced", pr->filter_bitset_cols[k], s);
    HIP_TRY(hipMemcpy(S + o_fbs[k], fc.bs_offsets.data(), (size_t)nseg * 8, hipMemcpyHostToDevice));
    P.fbs_offs[k] = reinterpret_cast<const uint64_t* const*>(S + o_fbs[k]);
  }
  P.seg_rows = reinterpret_cast<const uint32_t*>(S + o_segrows);
  P.counters = reinterpret_cast<unsigned long long*>(S + o_ctr);
  uint32_t* d_counts = reinterpret_cast<uint32_t*>(S + o_counts);
  unsigned long long* d_totals = reinterpret_cast<unsigned long long*>(S + o_totals);
  const unsigned grid = (unsigned)std::min<uint64_t>((nchunks + 3) / 4, (uint64_t)g_ctx.num_cu * 8);
  HIP_TRY(hipEventRecord(x->ev[1], st));
  hipLaunchKernelGGL(select_count_kernel, dim3(grid), dim3(256), 0, st, P, cps, d_counts);
  hipLaunchKernelGGL(select_scan_kernel, dim3(nseg), dim3(256), 0, st, d_counts, cps, d_totals);
  HIP_TRY(hipGetLastError());
  std::vector<unsigned long long> totals(nseg);
  HIP_TRY(hipMemcpyAsync(totals.data(), d_totals, (size_t)nseg * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));

  // The reference's loop (src/codegen/query/scan.cc:103-104,156-160), per segment instead of per row:
  //   if (skip > 0 && row_index++ < skip) continue;  ...send...  if (limit > 0 && output_recs >= limit) break;
  // `break` leaves the tuple loop only, so once the limit is reached every LATER segment still sends its first
  // passing row before it breaks again. Kept: results must be identical to the reference's.
  std::vector<VhSelectWindow> win(nseg);
  uint64_t remaining_skip = sp->skip, output_recs = 0, passed = 0;
  for (uint32_t s = 0; s < nseg; ++s) {
    const uint64_t n = totals[s];
    passed += n;
    const uint64_t skipped = std::min(n, remaining_skip);
    remaining_skip -= skipped;
    const uint64_t avail = n - skipped;
    uint64_t emit = avail;
    if (sp->limit > 0 && avail > 0) emit = output_recs >= sp->limit ? 1 : std::min(avail, sp->limit - output_recs);
    win[s] = VhSelectWindow{skipped, skipped + emit, output_recs};
    output_recs += emit;
  }
  rows->info.nrows = output_recs;
  rows->info.passed_recs = passed;
  if (output_recs) {
    size_t bytes = 0;
    for (int c = 0; c < sp->ncols; ++c) { rows->off[c] = bytes; bytes += (output_recs * vh_elem_size(rows->elem[c]) + 255) / 256 * 256; }
    if (bytes > ((size_t)64 << 30)) return vh_fail(VH_E_NOMEM, "select would return %llu rows (%zu bytes): add a limit", (unsigned long long)output_recs, bytes);
    if (bytes) {
      HIP_TRY(hipMalloc((void**)&rows->d_out, bytes));
      HIP_TRY(host_alloc_near_device((void**)&rows->h_out, bytes, hipHostMallocDefault));
    }
    VhSelectDev D{};
    D.ncols = sp->ncols;
    for (int c = 0; c < sp->ncols; ++c) {
      const VhColumn& col = t->cols[sp->cols[c]];
      D.esize[c] = (uint32_t)vh_elem_size(rows->elem[c]);
      D.out[c] = rows->d_out + rows->off[c];
      if (is_bitset_elem(col.elem)) {
        for (uint32_t s = 0; s < nseg; ++s)
          if (x->h_segrows[s] && !col.bs_offsets[s]) return vh_fail(VH_E_INVALID, "bitset column %d of segment %u was never synced", sp->cols[c], s);
        HIP_TRY(hipMemcpyAsync(S + o_bs[c], col.bs_offsets.data(), (size_t)nseg * 8, hipMemcpyHostToDevice, st));
        D.base[c] = nullptr; D.bs_offs[c] = reinterpret_cast<const uint64_t* const*>(S + o_bs[c]);
      } else { D.base[c] = col.base; D.stride[c] = col.stride; }
    }
    HIP_TRY(hipMemcpyAsync(S + o_sel, &D, sizeof(D), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(S + o_win, win.data(), (size_t)nseg * sizeof(VhSelectWindow), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(select_emit_kernel, dim3(grid), dim3(256), 0, st, P, cps, (const uint32_t*)d_counts,
                       reinterpret_cast<const VhSelectWindow*>(S + o_win), reinterpret_cast<const VhSelectDev*>(S + o_sel));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(x->ev[2], st));
    if (bytes) HIP_TRY(hipMemcpyAsync(rows->h_out, rows->d_out, bytes, hipMemcpyDeviceToHost, st));
  } else {
    HIP_TRY(hipEventRecord(x->ev[2], st));
  }
  HIP_TRY(hipEventRecord(x->ev[3], st));
  HIP_TRY(hipStreamSynchronize(st));   // D and win live on this frame
  float ms = 0;
  (void)hipEventElapsedTime(&ms, x->ev[1], x->ev[2]); rows->info.kernel_ms = ms;
  (void)hipEventElapsedTime(&ms, x->ev[0], x->ev[3]); rows->info.total_ms = ms;
  *out = rows.release();
  return VH_OK;
}


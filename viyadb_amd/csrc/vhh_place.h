// vhh_place.h — host side of libviya_hip, part of viya_hip.hip's translation unit (included there, in order; not a stand-alone header):
// device scratch of an execution context, and where a big tuple pool should lie (place_search: only inside vh_table_prepare).
// buffer out before the query depends on it.
struct VhPlaceHint {
  const void* stream_src[4] = {nullptr, nullptr, nullptr, nullptr}; size_t stream_bytes[4] = {0, 0, 0, 0}; int nstream = 0;     // the predicate columns (arenas or narrow copies) ...
  const void* gather_src = nullptr; size_t gather_bytes = 0;     // ... and where the survivors' values come from (projection or arena)
  size_t pool_off = 0, pool_bytes = 0;                           // the first tuple pool inside the scratch layout
};

// Where a tuple pool lands decides up to 10 % of a partitioning scan (C3 in round 3: 2.05 vs 2.35 ms, reproducibly for as long as the buffer
// lives; profiles/r03/NOTES.md "Where the tuple pool lands"). What is known: it is not the allocation call (hipMalloc of any size, or a
// 4 GiB-aligned VMM mapping next to hipMalloc'ed sources, land in either class alike), not the extent geometry, and it does not show in
// stores alone or in streams and gathers alone — only when whole-line stores to the buffer are MIXED with the table's read streams, i.e. it is
// how the pool's physical pages relate to the pages being read (consecutive allocations share a class over tens of GB). Round 5's experiment
// (tools/experiments/vmm_order.hip, profiles/r05/NOTES.md): with the sources AND the pool built from hipMemCreate chunks the same access mix
// runs at ONE speed — 3.54-3.59 ms over 24 builds, whatever the order and size of the chunks, between hipMalloc's fast (3.47) and slow (4.10)
// classes — but a library whose big buffers all came from that allocator returned wrong groups from DENSE_PART intermittently and died with
// GPU memory access faults in a third of its processes on this ROCm build (hipMalloc: never): a deterministic layout exists, it is not usable
// here. So the search stays, inside vh_table_prepare only and small: candidates one after the other, each pushed away from the last by a 6 GB
// spacer — at most VH_PLACE_TRIALS = 4 of them, half of what is free and VH_PLACE_GB = 16 GB; everything but the winner released again —,
// the access mix of a partitioning scan in miniature against THIS query's own columns on each (place_probe_kernel: ~1 ms per run), the
// fastest kept. One-off per context and size, like a kernel compile.
// vh_table_prepare: the calling thread's queries build derived layouts at once (not after VH_AUTO_PACK / VH_AUTO_NARROW uses) and may place a
// big tuple pool by measurement. An ORDINARY query never searches: it would hold tens of GB of free memory under the table lock for
// up to seconds, and a database process has other tables to allocate for meanwhile.
static thread_local bool g_preparing = false;
static std::mutex g_place_mu;      // one trial at a time: while it runs, most of the free memory is held (for some tens of milliseconds)
static int place_search(VhExec* x, size_t nb, const VhPlaceHint& h, void** out_ptr, float* out_score) {
  const int trials = g_preparing ? knobs().place_trials : 1;
  std::lock_guard<std::mutex> lk(g_place_mu);
  const auto t_begin = std::chrono::steady_clock::now();
  size_t free_b = 0, total_b = 0;
  if (trials < 2 || h.pool_bytes < ((size_t)128 << 20) || h.nstream < 1 || h.stream_bytes[0] < ((size_t)64 << 20) || !h.gather_src || h.gather_bytes < ((size_t)64 << 20) ||
      hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b / 2 < 2 * nb) return 1;      // (1: not tried, the caller allocates plainly)
  // (the driver clears memory another process left dirty when it is handed out again, at ~35 GB/s: the search stops early and is bounded,
  // so that it stays a 0.3-2.5 s one-off — about a kernel compile — and tens of milliseconds on a clean device)
  const size_t budget = std::min<size_t>(free_b / 2, (size_t)knobs().place_gb << 30);      // never more than half of what is free, nor VH_PLACE_GB (48 GB)
  const size_t spacer = (size_t)6 << 30;          // classes last for tens of GB: candidates ~9 GB apart sample them
  hipStream_t st = x->stream();
  if (g_ctx.stream != st) (void)hipStreamSynchronize(g_ctx.stream);      // (the probes read derived layouts a refresh may still be writing)
  VhPlaceArgs A{};
  {   // longest stream first; at most 3 GB each (the probe runs ~1 ms)
    int order[4] = {0, 1, 2, 3};
    std::sort(order, order + h.nstream, [&](int a, int b) { return h.stream_bytes[a] > h.stream_bytes[b]; });
    A.nsrc = h.nstream;
    for (int s = 0; s < h.nstream; ++s) {
      A.src[s] = reinterpret_cast<const vh_u32x4*>(h.stream_src[order[s]]);
      A.n16[s] = std::min<size_t>(h.stream_bytes[order[s]], (size_t)3 << 30) / 4096 * 256;
    }
  }
  A.rec = reinterpret_cast<const uint64_t*>(reinterpret_cast<uintptr_t>(h.gather_src) & ~(uintptr_t)7);      // (a projection's column may start anywhere inside its first record)
  A.nrec = (uint64_t)h.gather_bytes / 8;
  A.lines = std::min<size_t>(h.pool_bytes, (size_t)1 << 30) / 128;
  // One candidate at a time, each behind a spacer that pushes it away from the last; the search stops once it has seen four candidates and
  // holds one that beats the slowest seen by 5.5 % (both classes seen, a fast one in hand), or when the trials / the memory bound are used up.
  std::vector<void*> cand, spacers;
  int best = -1; float best_ms = 0, worst_ms = 0;
  size_t held = 0;
  for (int i = 0; i < trials && held + nb <= budget; ++i) {
    void* c = nullptr;
    if (hipMalloc(&c, nb) != hipSuccess) { (void)hipGetLastError(); break; }
    cand.push_back(c); held += nb;
    float ms = 1e9f;
    A.dst = reinterpret_cast<vh_u32x4*>(static_cast<char*>(c) + (h.pool_off + 127) / 128 * 128);
    A.sink = reinterpret_cast<unsigned long long*>(c);
    for (int rep = 0; rep < 4; ++rep) {
      (void)hipEventRecord(x->ev[0], st);
      hipLaunchKernelGGL(place_probe_kernel, dim3((unsigned)g_ctx.num_cu * 8), dim3(256), 0, st, A);
      (void)hipEventRecord(x->ev[1], st);
      float m = 0;
      if (hipEventSynchronize(x->ev[1]) != hipSuccess || hipEventElapsedTime(&m, x->ev[0], x->ev[1]) != hipSuccess) { (void)hipGetLastError(); m = 1e9f; }
      if (rep && m < ms) ms = m;
    }
    if (knobs().trace_alloc) fprintf(stderr, "vh alloc scratch candidate %d %p %.3f ms\n", i, c, ms);
    if (best < 0 || ms < best_ms) { best = i; best_ms = ms; }
    if (ms < 1e8f && ms > worst_ms) worst_ms = ms;
    if (i >= 3 && best_ms * 1.055f <= worst_ms) break;
    void* sp = nullptr;
    if (i + 1 < trials && held + spacer + nb <= budget) { if (hipMalloc(&sp, spacer) == hipSuccess) { spacers.push_back(sp); held += spacer; } else (void)hipGetLastError(); }
  }
  for (void* sp : spacers) (void)hipFree(sp);
  for (size_t i = 0; i < cand.size(); ++i) if ((int)i != best) (void)hipFree(cand[i]);
  if (best < 0) return 1;
  *out_ptr = cand[best]; *out_score = best_ms;
  if (knobs().trace_alloc) fprintf(stderr, "vh alloc scratch trial: %zu candidates of %zu bytes, %zu spacers of %zu, kept %d (%.3f ms), %.1f ms in all\n", cand.size(), nb, spacers.size(), spacer,
                                   best, best_ms, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
  return VH_OK;
}

static int install_scratch(VhExec* x, void* ptr, size_t nb, bool placed = false) {
  x->scratch_placed = placed;
  x->scratch = static_cast<char*>(ptr);
  trace_alloc("scratch", x->scratch, nb);
  x->scratch_bytes = nb;
  if (test_env("VH_POISON")) {   // tests: nothing may depend on what fresh scratch holds
    HIP_TRY(hipMemsetAsync(x->scratch, 0xA5, nb, x->stream()));
    HIP_TRY(hipStreamSynchronize(x->stream()));
  }
  return VH_OK;
}
static size_t scratch_size_for(size_t bytes) { return std::max(bytes + bytes / 4, (size_t)1 << 20); }
static int ensure_scratch(VhExec* x, size_t bytes, const VhPlaceHint* hint = nullptr) {
  if (bytes <= x->scratch_bytes) return VH_OK;
  HIP_TRY(hipStreamSynchronize(x->stream()));
  if (x->scratch) { HIP_TRY(hipFree(x->scratch)); x->scratch = nullptr; x->scratch_bytes = 0; }
  const size_t nb = scratch_size_for(bytes);
  void* ptr = nullptr; float score = 0;
  bool placed = true;
  if (!hint || place_search(x, nb, *hint, &ptr, &score) != VH_OK) { HIP_TRY(hipMalloc(&ptr, nb)); placed = false; }
  return install_scratch(x, ptr, nb, placed);
}

// The pool search compares candidates against the query's read streams WHERE THEY LIE; in about a third of the processes every candidate
// scores alike and slow, because the class is set by where the projection and the narrow copies landed (tools/derived_probe.py). Once per
// table, the first time a big tuple pool is placed for a query that reads derived layouts, a second configuration is tried: the derived
// layouts copied to another place (the table's data stays where it is), the pool search repeated against the copies, and whichever
// configuration scores better is kept — the other's buffers are released. The probe orders configurations of ONE process reliably; it
// was not reliable as an absolute measure (profiles/r03/NOTES.md), hence a comparison and not a threshold. *moved: the derived layouts
// now live elsewhere — the query being planned holds their old addresses and has to be planned again.
static int place_with_derived(vh_table* t, VhExec* x, size_t bytes, const VhPlaceHint& h, bool* moved) {
  *moved = false;
  HIP_TRY(hipStreamSynchronize(x->stream()));
  if (x->scratch) { HIP_TRY(hipFree(x->scratch)); x->scratch = nullptr; x->scratch_bytes = 0; }
  const size_t nb = scratch_size_for(bytes);
  void* A = nullptr; float sA = 0;
  if (place_search(x, nb, h, &A, &sA) != VH_OK) return VH_OK;        // (no search possible: ensure_scratch allocates plainly)
  struct Clone { char** ref; char* was; char* now; size_t bytes; };
  std::vector<Clone> clones;
  size_t need = 0;
  for (auto& pk : t->packs) if (pk->base) { clones.push_back(Clone{&pk->base, pk->base, nullptr, (size_t)pk->cap_seg * pk->stride + 256}); need += clones.back().bytes; }
  for (auto& nw : t->narrows) if (nw->base) { clones.push_back(Clone{&nw->base, nw->base, nullptr, (size_t)nw->cap_seg * nw->stride + 256}); need += clones.back().bytes; }
  for (auto& pp : t->predpacks) for (int q = 0; q < pp->nplanes; ++q) if (pp->pbase[q]) { clones.push_back(Clone{&pp->pbase[q], pp->pbase[q], nullptr, (size_t)pp->cap_seg * pp->pstride[q] + 256}); need += clones.back().bytes; }
  size_t free_b = 0, total_b = 0;
  const size_t spacer_bytes = (size_t)8 << 30;
  if (clones.empty() || hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b / 2 < need + nb + spacer_bytes) return install_scratch(x, A, nb, true);
  void* spacer = nullptr;
  if (hipMalloc(&spacer, spacer_bytes) != hipSuccess) { (void)hipGetLastError(); spacer = nullptr; }
  bool ok = true;
  for (auto& c : clones) {
    if (hipMalloc((void**)&c.now, c.bytes) != hipSuccess) { (void)hipGetLastError(); c.now = nullptr; ok = false; break; }
    if (hipMemcpyAsync(c.now, c.was, c.bytes, hipMemcpyDeviceToDevice, x->stream()) != hipSuccess) { (void)hipGetLastError(); ok = false; break; }
  }
  if (spacer) (void)hipFree(spacer);
  if (ok && hipStreamSynchronize(x->stream()) != hipSuccess) { (void)hipGetLastError(); ok = false; }
  void* B = nullptr; float sB = 0;
  if (ok) {
    VhPlaceHint hb = h;
    auto remap = [&](const void* p) -> const void* {
      const char* q = static_cast<const char*>(p);
      for (auto& c : clones) if (q >= c.was && q < c.was + c.bytes) return c.now + (q - c.was);
      return p;
    };
    for (int i = 0; i < hb.nstream; ++i) hb.stream_src[i] = remap(hb.stream_src[i]);
    hb.gather_src = remap(hb.gather_src);
    if (place_search(x, nb, hb, &B, &sB) != VH_OK) B = nullptr;
  }
  if (knobs().trace_alloc) fprintf(stderr, "vh alloc derived layouts: where they lie %.3f ms, copied elsewhere %.3f ms -> %s\n", sA, B ? sB : 0.f, B && sB < sA * 0.985f ? "moved" : "kept");
  if (B && sB < sA * 0.985f) {
    table_quiesce(t);                       // (queries of other contexts may still read the old copies)
    for (auto& c : clones) { (void)hipFree(c.was); *c.ref = c.now; }
    (void)hipFree(A);
    *moved = true;
    return install_scratch(x, B, nb, true);
  }
  for (auto& c : clones) if (c.now) (void)hipFree(c.now);
  if (B) (void)hipFree(B);
  return install_scratch(x, A, nb, true);
}

// vhh_place.h — host side of libviya_hip, part of viya_hip.hip's translation unit (included there, in order; not a stand-alone header): the
// device scratch of an execution context.
//
// Where a big tuple pool lands decides up to 10 % of a partitioning scan for as long as the buffer lives (profiles/r03/NOTES.md "Where the
// tuple pool lands": how the pool's physical pages relate to the pages being read). Rounds 3-5 searched for a good place inside
// vh_table_prepare — candidates behind spacers, a probe kernel per candidate, the derived layouts copied elsewhere as a second configuration —
// and round 5 measured what that still bought once the ring writer wrote whole lines from a block's shared waiting lines: about 3 %
// (ten prepared against ten unprepared processes), for 44 % of a profiled run's GPU time and tens of GB held for up to seconds. The search is
// gone (round 6); a deterministic layout from hipMemCreate chunks exists but is not usable on this ROCm build (profiles/r05/NOTES.md,
// tools/experiments/vmm_order.hip). vh_table_prepare still makes the calling thread's queries build their derived layouts at once.
static thread_local bool g_preparing = false;
static int install_scratch(VhExec* x, void* ptr, size_t nb) {
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
static int ensure_scratch(VhExec* x, size_t bytes) {
  if (bytes <= x->scratch_bytes) return VH_OK;
  HIP_TRY(hipStreamSynchronize(x->stream()));
  if (x->scratch) { HIP_TRY(hipFree(x->scratch)); x->scratch = nullptr; x->scratch_bytes = 0; }
  const size_t nb = scratch_size_for(bytes);
  void* ptr = nullptr;
  HIP_TRY(hipMalloc(&ptr, nb));
  return install_scratch(x, ptr, nb);
}


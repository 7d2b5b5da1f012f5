// viya_hip.hip — C-ABI implementation (include/viya_hip.h) over the gfx950 kernels.
//
// Host-side responsibilities that the reference performs inside the generated
// function and that therefore live on this side of the boundary:
//   * segment skipping from per-segment min/max (SegmentSkipBuilder,
//     src/codegen/query/filter.cc:263-335; use at src/codegen/query/scan.cc:48-51)
//   * stats.scanned_recs / scanned_segments / aggregated_recs bookkeeping
//     (scan.cc:44,51,246)
//   * choosing the aggregate-table organisation (the reference always uses
//     std::unordered_map, scan.cc:174-177; here: LDS-resident dense table, per-XCD
//     private dense tables in HBM/L2, or an open-addressing hash table in HBM).
#include "vh_small_kernels.h"
#include "vh_launch.h"
#include "vh_jit.h"
#include "vh_hpart.h"

#include <sched.h>
#include <algorithm>
#include <cctype>
#include <cfloat>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

// ------------------------------------------------------------------- errors
static thread_local char g_err[512] = "";
static int vh_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return vh_fail(e_ == hipErrorOutOfMemory ? VH_E_NOMEM : VH_E_DEVICE, "%s failed: %s (%s:%d)", \
                     #expr, hipGetErrorString(e_), __FILE__, __LINE__);                        \
  } while (0)

// ------------------------------------------------------------------ context
struct VhContext {
  bool inited = false;
  int device = 0;
  int num_cu = 256;
  int num_xcd = 8;
  size_t lds_per_block = 65536;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
};
static VhContext g_ctx;
static std::mutex g_mu;

// Every environment knob of the library, read ONCE (first use, normally vh_init) instead of wherever the planner happened to want
// one. They exist for measurements and tests: defaults are what every reported number was taken with (DESIGN.md "Knobs"). The
// VH_TEST_* / VH_POISON / VH_NO_SPLIT_TILE / VH_PART_TABLE_KB knobs are the exception — tests switch them between two queries of one process — and stay getenv() calls
// at their (cold) sites.
struct VhKnobs {
  bool trace_alloc, no_topk, no_stage, jit_verbose, skip_phase2, no_direct_emit, times;
  int max_exec, auto_narrow, auto_pack, jit_ablate, hp_ablate, hp_bpp, pack_plain, lanes_block, blocks_per_cu, unit_rows, grid, ext_tuples, ext_pad, split_bpc, bw_blocks_per_cu, place_trials, place_gb, hp_list, hp_stream, deliver_blocks;
  double hp_load_g, hp_load_s;
};
static const VhKnobs& knobs() {
  static const VhKnobs k = [] {
    auto flag = [](const char* n) { return getenv(n) != nullptr; };
    auto num = [](const char* n, int dflt) { const char* e = getenv(n); return e ? atoi(e) : dflt; };
    auto real = [](const char* n, double dflt) { const char* e = getenv(n); return e ? atof(e) : dflt; };
    VhKnobs x{};
    x.trace_alloc = flag("VH_TRACE_ALLOC"); x.no_topk = flag("VH_NO_TOPK"); x.no_stage = flag("VH_NO_STAGE");
    x.jit_verbose = flag("VH_JIT_VERBOSE"); x.skip_phase2 = flag("VH_ABLATE_NO_PHASE2"); x.no_direct_emit = flag("VH_NO_DIRECT_EMIT"); x.times = flag("VH_TIMES");
    x.max_exec = std::max(1, num("VH_MAX_EXEC", 16));
    x.auto_narrow = num("VH_AUTO_NARROW", 3); x.auto_pack = num("VH_AUTO_PACK", 3);
    x.jit_ablate = num("VH_JIT_ABLATE", 0); x.hp_ablate = num("VH_HP_ABLATE", 0); x.hp_bpp = num("VH_HP_BPP", 0); x.pack_plain = num("VH_PACK_PLAIN", 0);
    x.lanes_block = num("VH_LANES_BLOCK", 0); x.blocks_per_cu = num("VH_BLOCKS_PER_CU", 0); x.unit_rows = num("VH_UNIT_ROWS", 0); x.grid = num("VH_GRID", 0);
    x.ext_tuples = num("VH_EXT_TUPLES", 0); x.ext_pad = std::max(0, num("VH_EXT_PAD", 8)) / 8 * 8; x.place_trials = num("VH_PLACE_TRIALS", 8); x.place_gb = std::max(1, num("VH_PLACE_GB", 48)); x.hp_list = num("VH_HP_LIST", 0);
    x.hp_stream = num("VH_HP_STREAM", 0);             // chunk launches of a streamed result (0: off — measured: the link, not the wait for the kernels, bounds the delivery; profiles/r04/NOTES.md)
    x.deliver_blocks = num("VH_DELIVER_BLOCKS", 64);  // blocks of deliver_kernel; 0: big results through hipMemcpyAsync (the DMA engine)
    x.split_bpc = num("VH_SPLIT_BPC", 4); x.bw_blocks_per_cu = std::max(1, num("VH_BW_BLOCKS_PER_CU", 8));
    x.hp_load_g = real("VH_HP_LOAD_G", 0.7); x.hp_load_s = real("VH_HP_LOAD_S", 0.7);
    return x;
  }();
  return k;
}

// Pinned host memory NEXT TO THE GPU. Linux places pages on the NUMA node of the CPU that first touches them, and hipHostMalloc pins (touches)
// them in the calling thread: on a two-socket host the staging buffer of a big result landed on either socket, and device-to-host copies
// into the far one run at 30 GB/s instead of 57 (tools/experiments/d2h_bw2.hip; C5 delivered its 35 M groups in 23 ms or in 13). The calling
// thread therefore sits on the CPUs of the device's own node (sysfs: the PCI device's numa_node and that node's cpulist) while the buffer is
// allocated and touched, and gets its affinity back afterwards. No node information (one socket, a container that hides sysfs): plain allocation.
static std::vector<int> device_node_cpus() {
  std::vector<int> cpus;
  char bdf[64] = "";
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), g_ctx.device) != hipSuccess) return cpus;
  for (char* c = bdf; *c; ++c) *c = (char)tolower(*c);
  char path[160];
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
  int node = -1;
  if (FILE* f = fopen(path, "r")) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
  if (getenv("VH_HOST_NUMA_NODE")) node = atoi(getenv("VH_HOST_NUMA_NODE"));      // measurement / hosts whose sysfs says -1
  if (node < 0) return cpus;
  snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
  FILE* f = fopen(path, "r");
  if (!f) return cpus;
  char list[4096] = "";
  if (!fgets(list, sizeof(list), f)) list[0] = 0;
  fclose(f);
  for (char* p = list; *p;) {          // "0-63,128-191"
    char* end = nullptr;
    const long a = strtol(p, &end, 10);
    if (end == p) break;
    long b = a;
    p = end;
    if (*p == '-') { b = strtol(p + 1, &end, 10); p = end; }
    for (long c = a; c <= b && c < CPU_SETSIZE; ++c) cpus.push_back((int)c);
    if (*p == ',') ++p; else break;
  }
  return cpus;
}
static hipError_t host_alloc_near_device(void** out, size_t bytes, unsigned flags) {
  static const std::vector<int> cpus = device_node_cpus();
  cpu_set_t old_set, near_set;
  bool moved = false;
  if (!cpus.empty() && bytes >= ((size_t)1 << 20) && sched_getaffinity(0, sizeof(old_set), &old_set) == 0) {
    CPU_ZERO(&near_set);
    int n = 0;
    for (int c : cpus) if (CPU_ISSET(c, &old_set)) { CPU_SET(c, &near_set); ++n; }      // (only CPUs the thread may run on anyway: a cpuset is respected)
    moved = n > 0 && sched_setaffinity(0, sizeof(near_set), &near_set) == 0;
  }
  hipError_t he = hipHostMalloc(out, bytes, flags);
  if (he == hipSuccess && moved) { volatile char* p = static_cast<volatile char*>(*out); for (size_t i = 0; i < bytes; i += 4096) p[i] = 0; }
  if (moved) (void)sched_setaffinity(0, sizeof(old_set), &old_set);
  return he;
}

// The HIP current device is per thread: every entry point that allocates or launches binds the calling thread to the
// library's device first (query threads of a server pool never called vh_init themselves).
#define VH_ENTER() do { if (g_ctx.inited) (void)hipSetDevice(g_ctx.device); } while (0)

extern "C" const char* vh_last_error(void) { return g_err; }
extern "C" const char* vh_version(void) { return "viya_hip 0.1 (gfx950)"; }

extern "C" int vh_init(int device_id) {
  std::lock_guard<std::mutex> lk(g_mu);
  HIP_TRY(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device_id));
  g_ctx.device = device_id;
  g_ctx.num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  g_ctx.lds_per_block = prop.sharedMemPerBlock ? prop.sharedMemPerBlock : 65536;
  g_ctx.num_xcd = 8;
  if (!g_ctx.own_stream) HIP_TRY(hipStreamCreateWithFlags(&g_ctx.own_stream, hipStreamNonBlocking));
  if (!g_ctx.stream) g_ctx.stream = g_ctx.own_stream;
  g_ctx.inited = true;
  (void)knobs();
  return VH_OK;
}

extern "C" int vh_set_stream(void* hip_stream) {
  if (!g_ctx.inited) return vh_fail(VH_E_INVALID, "vh_init has not been called");
  // NULL is a real stream (the legacy default stream, which is what torch.cuda.current_stream() is unless
  // the caller changed it); VH_OWN_STREAM restores the library's private stream
  g_ctx.stream = hip_stream == VH_OWN_STREAM ? g_ctx.own_stream : (hipStream_t)hip_stream;
  return VH_OK;
}

// -------------------------------------------------------------------- table
struct VhColumn {
  int kind = 0, elem = 0, esize = 0;
  char* base = nullptr;     // arena: cap_seg x stride bytes (+ tail pad)
  uint64_t stride = 0;      // bytes between segments
  // bitset CSR mirrors (one pair per segment)
  std::vector<uint64_t*> bs_offsets;
  std::vector<void*> bs_values;
  std::vector<uint64_t> bs_nvalues;
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
  char* scratch = nullptr; size_t scratch_bytes = 0; bool scratch_placed = false;      // placed: chosen among candidates by vh_table_prepare (place_search)
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
  uint32_t rec_bytes = 0;           // power of two, 8..64
  char* base = nullptr; uint64_t stride = 0; uint32_t cap_seg = 0;
  std::vector<uint64_t> seg_mod;    // value of vh_table::seg_mod[s] the segment was packed at (0: never)
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
  bool automatic = false;
};
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
  std::map<std::string, uint64_t> groups_seen;           // group-column signature -> groups of the last query (hash sizing)
  std::map<std::string, std::pair<uint64_t, uint64_t>> sel_cache;   // filter signature + table state -> (passed, sampled) of the selectivity probe
  std::vector<std::unique_ptr<VhPack>> packs;
  std::vector<std::unique_ptr<VhNarrow>> narrows;
  bool derived_tried = false;                   // place_with_derived ran (once per table)
  std::map<int, uint32_t> pred_seen;                     // column -> queries that filtered on it (automatic narrow copies)
  std::vector<uint64_t> seg_mod;                          // sync_epoch of the last change to a segment's columns
  uint32_t* d_packrows = nullptr; size_t d_packrows_cap = 0;
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
      c.bs_offsets.resize(ncap, nullptr); c.bs_values.resize(ncap, nullptr); c.bs_nvalues.resize(ncap, 0); c.bs_maxid.resize(ncap, 0);
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
static int exec_acquire(vh_table* t, VhExec** out) {
  const size_t max_exec = (size_t)knobs().max_exec;
  std::unique_lock<std::mutex> lk(t->pool_mu);
  for (;;) {
    for (auto& x : t->execs) if (!x->busy) { x->busy = true; *out = x.get(); return VH_OK; }
    if (t->execs.size() < max_exec) break;
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
  { int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi); HIP_TRY(hipStreamCreateWithPriority(&x->copy, hipStreamNonBlocking, getenv("VH_COPY_PRIO") ? hi : 0)); }
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

extern "C" void vh_table_destroy(vh_table* t) {
  if (!t) return;
  VH_ENTER();
  (void)hipStreamSynchronize(g_ctx.stream);
  for (auto& x : t->execs) { (void)hipStreamSynchronize(x->stream()); exec_free(x.get()); }
  for (auto& c : t->cols) {
    if (c.base) (void)hipFree(c.base);
    for (auto p : c.bs_offsets) if (p) (void)hipFree(p);
    for (auto p : c.bs_values) if (p) (void)hipFree(p);
  }
  if (t->d_stats) (void)hipFree(t->d_stats);
  if (t->d_packflag) (void)hipFree(t->d_packflag);
  for (auto& pk : t->packs) if (pk->base) (void)hipFree(pk->base);
  for (auto& nw : t->narrows) if (nw->base) (void)hipFree(nw->base);
  if (t->d_packrows) (void)hipFree(t->d_packrows);
  delete t;
}

// What a partitioning query reads while it appends tuples, and where in its scratch the tuple pool will lie: enough to try a scratch
// buffer out before the query depends on it.
struct VhPlaceHint {
  const void* stream_src[4] = {nullptr, nullptr, nullptr, nullptr}; size_t stream_bytes[4] = {0, 0, 0, 0}; int nstream = 0;     // the predicate columns (arenas or narrow copies) ...
  const void* gather_src = nullptr; size_t gather_bytes = 0;     // ... and where the survivors' values come from (projection or arena)
  size_t pool_off = 0, pool_bytes = 0;                           // the first tuple pool inside the scratch layout
};

// Where a tuple pool lands decides 10 % of a partitioning scan (C3: 2.05 vs 2.35 ms, reproducibly for as long as the buffer lives;
// profiles/r03/NOTES.md "Where the tuple pool lands"). What was learned about it: it is not the allocation call (hipMalloc of any size,
// or a 4 GiB-aligned VMM mapping, land in either class alike), not the extent geometry, and it does not show in stores alone or in streams
// and gathers alone — only when whole-line stores to the buffer are MIXED with the table's read streams, i.e. it is how the pool's
// physical pages relate to the pages being read (consecutive allocations share a class over tens of GB; the mapping of physical
// addresses to HBM stacks / ranks is not visible from here). So the library measures: when a context needs a new scratch buffer for
// a tuple pool of >= 256 MB, it allocates candidates one after the other, each pushed away from the last by a 6 GB spacer (at most VH_PLACE_TRIALS = 12 of them,
// three quarters of what is free and VH_PLACE_GB = 96 GB; everything but the winner released again), runs the access mix of a partitioning scan in miniature against THIS query's
// own columns on each (place_probe_kernel: ~1 ms per run) and keeps the fastest. One-off per context and size, like a kernel compile.
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
  if (getenv("VH_POISON")) {   // tests: nothing may depend on what fresh scratch holds
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
static int ensure_segrows(VhExec* x, size_t n) {
  if (n <= x->h_segrows_cap) return VH_OK;
  if (x->h_segrows) (void)hipHostFree(x->h_segrows);
  size_t cap = std::max<size_t>(n * 2, 1024);
  HIP_TRY(hipHostMalloc((void**)&x->h_segrows, cap * sizeof(uint32_t), hipHostMallocDefault));
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

// ------------------------------------------------------- payload projections (vh_table_pack)
#define VH_PACK_STALE 9001      // (internal) pack_refresh: a value no longer fits its stored width, the projection must go
// (Re)pack the segments of [first, first + n) whose columns changed since they were last packed.
static int pack_refresh(vh_table* t, VhPack* pk, uint32_t first, uint32_t n) {
  if (first + n > t->nseg) n = t->nseg > first ? t->nseg - first : 0;
  if (!n) return VH_OK;
  if (pk->cap_seg < t->cap_seg) {                      // the table grew: move the arena
    table_quiesce(t);
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
  uint32_t s = first;
  while (s < first + n) {
    if (pk->seg_mod[s] == t->seg_mod[s]) { ++s; continue; }
    uint32_t e = s;
    while (e < first + n && pk->seg_mod[e] != t->seg_mod[e] && e - s < 4096) ++e;
    const uint32_t cnt = e - s;
    if (t->d_packrows_cap < cnt) {
      HIP_TRY(hipStreamSynchronize(g_ctx.stream));
      if (t->d_packrows) HIP_TRY(hipFree(t->d_packrows));
      t->d_packrows = nullptr; t->d_packrows_cap = 0;
      HIP_TRY(hipMalloc((void**)&t->d_packrows, (size_t)std::max<uint32_t>(cnt, 1024) * sizeof(uint32_t)));
      t->d_packrows_cap = std::max<uint32_t>(cnt, 1024);
    }
    std::vector<uint32_t> rows(cnt);
    for (uint32_t i = 0; i < cnt; ++i) rows[i] = (uint32_t)t->seg_rows[s + i];
    HIP_TRY(hipMemcpyAsync(t->d_packrows, rows.data(), (size_t)cnt * sizeof(uint32_t), hipMemcpyHostToDevice, g_ctx.stream));
    VhPackArgs A{};
    A.ncols = (int32_t)pk->cols.size(); A.rec_bytes = pk->rec_bytes;
    for (size_t c = 0; c < pk->cols.size(); ++c) {
      const VhColumn& col = t->cols[pk->cols[c]];
      A.src[c] = col.base; A.src_stride[c] = col.stride; A.esize[c] = (uint32_t)col.esize; A.off[c] = pk->off[c];
      A.wbytes[c] = pk->width[c];
      if (col.elem == VH_I8 || col.elem == VH_I16 || col.elem == VH_I32 || col.elem == VH_I64) A.sgn_mask |= 1u << c;
    }
    if (!t->d_packflag) { HIP_TRY(hipMalloc((void**)&t->d_packflag, 256)); HIP_TRY(hipMemsetAsync(t->d_packflag, 0, 256, g_ctx.stream)); }
    A.overflow = t->d_packflag;
    A.dst = pk->base; A.dst_stride = pk->stride; A.rows = t->d_packrows; A.seg_first = s;
    dim3 grid((unsigned)std::min<uint64_t>(64, (t->segment_rows + 255) / 256), cnt);
    hipLaunchKernelGGL(pack_kernel, grid, dim3(256), 256 * pk->rec_bytes, g_ctx.stream, A);
    HIP_TRY(hipGetLastError());
    unsigned int ovf = 0;
    if (pk->compressed) HIP_TRY(hipMemcpyAsync(&ovf, t->d_packflag, sizeof(ovf), hipMemcpyDeviceToHost, g_ctx.stream));
    HIP_TRY(hipStreamSynchronize(g_ctx.stream));        // `rows` lives on this frame; d_packrows is reused by the next batch
    if (ovf) {                                          // a synced value outgrew its stored width: the projection is void (the caller drops it)
      HIP_TRY(hipMemsetAsync(t->d_packflag, 0, 256, g_ctx.stream));
      return VH_PACK_STALE;
    }
    for (uint32_t i = s; i < e; ++i) pk->seg_mod[i] = t->seg_mod[i];
    s = e;
  }
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
    std::unique_ptr<VhPack> pk(new VhPack());
    pk->cols = ord; pk->off = off; pk->width = width; pk->rec_bytes = rec; pk->automatic = automatic; pk->compressed = compress;
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
// (Re)copy the segments of [first, first + n) whose column changed since they were last copied.
static int narrow_refresh(vh_table* t, VhNarrow* nw, uint32_t first, uint32_t n) {
  if (first + n > t->nseg) n = t->nseg > first ? t->nseg - first : 0;
  if (!n) return VH_OK;
  const uint64_t padded = t->padded_rows;
  if (nw->cap_seg < t->cap_seg) {
    table_quiesce(t);
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
  const VhColumn& c = t->cols[nw->col];
  uint32_t s = first;
  while (s < first + n) {
    if (nw->seg_mod[s] == t->seg_mod[s]) { ++s; continue; }
    uint32_t e = s;
    while (e < first + n && nw->seg_mod[e] != t->seg_mod[e] && e - s < 4096) ++e;
    const dim3 grid((unsigned)std::min<uint64_t>(64, (padded + 1023) / 1024), e - s);
    if (nw->width == 1)
      hipLaunchKernelGGL((narrow_kernel<uint8_t>), grid, dim3(256), 0, g_ctx.stream, reinterpret_cast<const uint32_t*>(c.base), c.stride / 4,
                         reinterpret_cast<uint8_t*>(nw->base), nw->stride, padded, s);
    else
      hipLaunchKernelGGL((narrow_kernel<uint16_t>), grid, dim3(256), 0, g_ctx.stream, reinterpret_cast<const uint32_t*>(c.base), c.stride / 4,
                         reinterpret_cast<uint16_t*>(nw->base), nw->stride / 2, padded, s);
    HIP_TRY(hipGetLastError());
    for (uint32_t i = s; i < e; ++i) nw->seg_mod[i] = t->seg_mod[i];
    s = e;
  }
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
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

extern "C" int vh_table_narrow(vh_table* t, const int32_t* cols, int32_t ncols) {
  if (!t || (!cols && ncols)) return vh_fail(VH_E_INVALID, "null argument");
  VH_ENTER();
  std::lock_guard<std::mutex> lk(t->mu);
  for (int i = 0; i < ncols; ++i)
    if (int rc = table_narrow_locked(t, cols[i], false)) return rc;
  return VH_OK;
}

extern "C" int vh_table_pack_ex(vh_table* t, const int32_t* cols, int32_t ncols, uint32_t form) {
  if (!t) return vh_fail(VH_E_INVALID, "null table");
  if (form > VH_PACK_COMPRESSED) return vh_fail(VH_E_INVALID, "vh_table_pack_ex: form %u", form);
  VH_ENTER();
  std::lock_guard<std::mutex> lk(t->mu);
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

extern "C" int vh_table_unpack(vh_table* t) {
  if (!t) return vh_fail(VH_E_INVALID, "null table");
  VH_ENTER();
  std::lock_guard<std::mutex> lk(t->mu);
  table_quiesce(t);
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  for (auto& pk : t->packs) if (pk->base) { (void)hipFree(pk->base); t->device_bytes -= (size_t)pk->cap_seg * pk->stride + 256; }
  t->packs.clear();
  t->gather_seen.clear();
  for (auto& nw : t->narrows) if (nw->base) { (void)hipFree(nw->base); t->device_bytes -= (size_t)nw->cap_seg * nw->stride + 256; }
  t->narrows.clear();
  t->pred_seen.clear();
  return VH_OK;
}

// ------------------------------------------------------- typed host helpers
template <typename T> static T vh_lit_host(uint64_t bits) { T v; memcpy(&v, &bits, sizeof(T)); return v; }
static uint64_t order_key_of_bits(int elem, uint64_t bits) {
  switch (elem) {
    case VH_U8: return (uint8_t)bits;
    case VH_U16: return (uint16_t)bits;
    case VH_U32: return (uint32_t)bits;
    case VH_U64: return bits;
    case VH_I8: return (uint64_t)(int64_t)(int8_t)bits ^ (1ull << 63);
    case VH_I16: return (uint64_t)(int64_t)(int16_t)bits ^ (1ull << 63);
    case VH_I32: return (uint64_t)(int64_t)(int32_t)bits ^ (1ull << 63);
    case VH_I64: return bits ^ (1ull << 63);
    case VH_F32: { uint32_t b = (uint32_t)bits; return (b & 0x80000000u) ? (uint32_t)~b : (b | 0x80000000u); }
    default: return (bits & (1ull << 63)) ? ~bits : (bits | (1ull << 63));
  }
}
static uint64_t bits_of_order_key(int elem, uint64_t k) {
  switch (elem) {
    case VH_U8: case VH_U16: case VH_U32: case VH_U64: return k;
    case VH_I8: case VH_I16: case VH_I32: case VH_I64: return k ^ (1ull << 63);  // sign-extended 64-bit
    case VH_F32: { uint32_t b = (uint32_t)k; return (b & 0x80000000u) ? (b & 0x7FFFFFFFu) : (uint32_t)~b; }
    default: return (k & (1ull << 63)) ? (k & ~(1ull << 63)) : ~k;
  }
}
// identities of SegmentStats (src/codegen/db/store.cc:171-186): dmax = cpp_min_value,
// dmin = cpp_max_value — FLT_MIN / DBL_MIN (smallest positive) for floating dims.
static uint64_t stat_min_identity_key(int elem) {  // dmin initial = type max
  switch (elem) {
    case VH_F32: { float f = FLT_MAX; uint32_t b; memcpy(&b, &f, 4); return order_key_of_bits(elem, b); }
    case VH_F64: { double d = DBL_MAX; uint64_t b; memcpy(&b, &d, 8); return order_key_of_bits(elem, b); }
    case VH_U8: return 0xFFull; case VH_U16: return 0xFFFFull; case VH_U32: return 0xFFFFFFFFull;
    case VH_U64: return ~0ull;
    case VH_I8: return order_key_of_bits(elem, (uint64_t)(int64_t)INT8_MAX);
    case VH_I16: return order_key_of_bits(elem, (uint64_t)(int64_t)INT16_MAX);
    case VH_I32: return order_key_of_bits(elem, (uint64_t)(int64_t)INT32_MAX);
    default: return order_key_of_bits(elem, (uint64_t)INT64_MAX);
  }
}
static uint64_t stat_max_identity_key(int elem) {  // dmax initial = cpp_min_value
  switch (elem) {
    case VH_F32: { float f = FLT_MIN; uint32_t b; memcpy(&b, &f, 4); return order_key_of_bits(elem, b); }
    case VH_F64: { double d = DBL_MIN; uint64_t b; memcpy(&b, &d, 8); return order_key_of_bits(elem, b); }
    case VH_U8: case VH_U16: case VH_U32: case VH_U64: return 0;
    case VH_I8: return order_key_of_bits(elem, (uint64_t)(int64_t)INT8_MIN);
    case VH_I16: return order_key_of_bits(elem, (uint64_t)(int64_t)INT16_MIN);
    case VH_I32: return order_key_of_bits(elem, (uint64_t)(int64_t)INT32_MIN);
    default: return order_key_of_bits(elem, (uint64_t)INT64_MIN);
  }
}

extern "C" int vh_segment_stats(vh_table* t, uint32_t seg, int32_t col, vh_anynum* min_out, vh_anynum* max_out) {
  if (!t || col < 0 || (size_t)col >= t->cols.size() || seg >= t->nseg) return vh_fail(VH_E_INVALID, "vh_segment_stats: bad argument");
  auto& c = t->cols[col];
  if (!is_dim(c.kind)) return vh_fail(VH_E_INVALID, "column %d is not a dimension", col);
  const VhSegStat& s = t->stats[col][seg];
  const uint64_t lo = std::min(s.lo, stat_min_identity_key(c.elem));
  const uint64_t hi = std::max(s.hi, stat_max_identity_key(c.elem));
  if (min_out) { min_out->u64 = 0; uint64_t b = bits_of_order_key(c.elem, lo); memcpy(min_out, &b, c.esize); }
  if (max_out) { max_out->u64 = 0; uint64_t b = bits_of_order_key(c.elem, hi); memcpy(max_out, &b, c.esize); }
  return VH_OK;
}

// ------------------------------------------------------------------ results
struct vh_result {
  bool hpart = false;               // hashed partitioning ran: the table is a compact list of group records ...
  bool hp_direct = false;           // ... or its aggregation kernel already wrote the output columns (no emission kernel to run)
  int hp_chunks = 0;                // ... in this many chunk launches, each with a region of `hp_chunk_rows` rows of the output columns: delivered chunk by chunk
  uint64_t hp_chunk_rows = 0;
  VhHpArgs hp_args;                 // ... and the pool descriptors its kernels were given
  vh_table* table = nullptr;
  vh_result_info info{};
  int mode = 0;
  bool finalized = false;
  size_t plan_words = 0, seg_words = 0;        // layout of the pinned staging block [segment snapshot | program | literals] in u32 words
  int h_slot = -1;                             // staging buffer of `exec` this query finalises into
  std::string kernel;                          // symbol(s) of the scan kernel(s) launched for this query
  std::vector<VhProgOp> h_prog; std::vector<uint64_t> h_lits;   // the filter program as uploaded (VhPlanDev::prog / lits point into device scratch)
  std::vector<int> filter_bitset_cols;         // bitset metrics the filter compares the cardinality of (VhPlanDev::fbs_offs order)
  bool device_rows = false;                    // emitted rows must (also) exist in device memory: they are exchanged or gathered next
  VhExec* exec = nullptr;                      // owned from launch to vh_result_free: stream, scratch (device-side state), staging (host view)
  // device-side partial state
  VhPlanDev plan{};
  int nxcd = 1;
  std::vector<int> metric_elem;        // output element type per device metric (P.m order)
  std::vector<int> group_elem;
  std::string group_sig;
  int nhaving = 0;
  VhProgOp hprog[VH_MAX_HAVING] = {};
  uint8_t htype[VH_MAX_HAVING] = {};
  uint64_t hlits[VH_MAX_HAVING_LITS] = {};
  std::vector<int> user_metric;        // per plan metric: >= 0 index into P.m, < 0: -(bitset index + 1)
  uint64_t out_cap = 0;                // rows the output arrays can hold
  unsigned long long* d_out_count = nullptr;
  void* d_out_key[VH_MAX_GROUP] = {};
  void* d_out_state[VH_MAX_METRIC] = {};
  // host side after finalize: key / state arrays live in the table's pinned staging buffer `h_base`
  // (valid until the second-next query on the same table) at these offsets
  size_t out_region_off = 0, out_region_bytes = 0;   // device scratch: [counters | out_count | keys | states]
  size_t off_key[VH_MAX_GROUP] = {}, off_state[VH_MAX_METRIC] = {};
  std::vector<size_t> wide_off_state;   // a multi-pass result (more than VH_MAX_METRIC states): offsets of ALL its state arrays
  char* h_base = nullptr;
  uint64_t ngroups_host = 0;
  // device top-N (vh_plan.top_k): a second set of output arrays holding the kept superset
  uint64_t topk = 0;
  bool topk_active = false;
  int topk_src = 0; bool topk_src_is_key = false; int topk_elem = 0, topk_cls = 0, topk_desc = 0;
  uint64_t* d_topk_keys = nullptr;
  VhTopkState* d_topk_state = nullptr;
  void* d_out_key2[VH_MAX_GROUP] = {};
  void* d_out_state2[VH_MAX_METRIC] = {};
  const char* zero_begin = nullptr; const char* zero_end = nullptr;   // scratch range cleared by the one state memset
  char* d_xchg = nullptr;              // vh_result_partition: rows regrouped by owner (own allocation)
  std::vector<char*> d_pairs;          // vh_result_partition_pairs: one allocation per call
  // sharded queries: a merged result lives on a temporary merge table (owned), a gathered one in buffers of its own
  vh_table* owned_table = nullptr;
  char* d_own = nullptr; char* h_own = nullptr;
  ~vh_result() {
    if (d_xchg) (void)hipFree(d_xchg);
    for (char* p : d_pairs) (void)hipFree(p);
    if (exec) { (void)hipStreamSynchronize(exec->stream()); exec_release(table, exec); }   // nothing of this query may still run on a context the next one takes
    if (d_own) (void)hipFree(d_own);
    if (h_own) (void)hipHostFree(h_own);
    if (owned_table) vh_table_destroy(owned_table);
  }
};

extern "C" void vh_result_free(vh_result* r) { if (r) { VH_ENTER(); delete r; } }

extern "C" int vh_result_get_info(vh_result* r, vh_result_info* info) {
  if (!r || !info) return vh_fail(VH_E_INVALID, "null argument");
  *info = r->info;
  return VH_OK;
}

extern "C" const char* vh_result_kernel(vh_result* r) { return r ? r->kernel.c_str() : ""; }

extern "C" int vh_result_state_elem(vh_result* r, int32_t metric) {
  if (!r || metric < 0 || (size_t)metric >= r->user_metric.size()) return -1;
  const int u = r->user_metric[metric];
  return u >= 0 && (size_t)u < r->metric_elem.size() ? r->metric_elem[u] : -1;
}

extern "C" int vh_result_view(vh_result* r, const void** key_cols, const void** state_cols, const uint64_t** hidden_count) {
  if (!r || !r->finalized) return vh_fail(VH_E_INVALID, "result is not finalised");
  for (int i = 0; i < r->plan.ngroup; ++i)
    if (key_cols) key_cols[i] = r->h_base + r->off_key[i];
  for (size_t j = 0; j < r->user_metric.size(); ++j) {
    if (!state_cols) break;
    const int u = r->user_metric[j];
    state_cols[j] = r->h_base + (r->wide_off_state.empty() ? r->off_state[u] : r->wide_off_state[u]);
  }
  if (hidden_count) *hidden_count = !r->info.has_hidden_count ? nullptr
      : reinterpret_cast<const uint64_t*>(r->h_base + (r->wide_off_state.empty() ? r->off_state[r->plan.nmetric - 1] : r->wide_off_state.back()));
  return VH_OK;
}

extern "C" int vh_result_copy(vh_result* r, void* const* key_cols, void* const* state_cols, uint64_t* hidden_count) {
  if (!r || !r->finalized) return vh_fail(VH_E_INVALID, "result is not finalised");
  const void* kp[VH_MAX_GROUP]; const uint64_t* hp = nullptr;
  std::vector<const void*> spv(std::max<size_t>(r->user_metric.size(), 1));
  const void** sp = spv.data();
  int rc = vh_result_view(r, kp, sp, &hp);
  if (rc) return rc;
  const uint64_t ng = r->ngroups_host;
  for (int i = 0; i < r->plan.ngroup; ++i)
    if (key_cols && key_cols[i] && ng) memcpy(key_cols[i], kp[i], ng * vh_elem_size(r->plan.g[i].type()));
  for (size_t j = 0; j < r->user_metric.size(); ++j) {
    if (!state_cols || !state_cols[j] || !ng) continue;
    const int u = r->user_metric[j];
    memcpy(state_cols[j], sp[j], ng * vh_elem_size(r->metric_elem[u]));
  }
  if (hidden_count && hp && ng) memcpy(hidden_count, hp, ng * 8);
  return VH_OK;
}

// ----------------------------------------------------------------- planning
static int sop_for(int kind, int elem, int* sop, uint64_t* ident) {
  const bool sum = kind == VH_METRIC_SUM || kind == VH_METRIC_AVG || kind == VH_METRIC_COUNT || kind == VH_METRIC_HIDDEN_COUNT;
  const bool mx = kind == VH_METRIC_MAX, mn = kind == VH_METRIC_MIN;
  if (!sum && !mx && !mn) return -1;
  *ident = 0;
  auto fbits = [](float f) { uint32_t b; memcpy(&b, &f, 4); return (uint64_t)b; };
  auto dbits = [](double d) { uint64_t b; memcpy(&b, &d, 8); return b; };
  switch (elem) {
    case VH_U8: case VH_U16: case VH_U32:
      if (sum) *sop = SOP_ADD32;
      else if (mx) { *sop = SOP_MAX_U32; *ident = 0; }
      else { *sop = SOP_MIN_U32; *ident = elem == VH_U8 ? 0xFFu : elem == VH_U16 ? 0xFFFFu : 0xFFFFFFFFu; }
      return 0;
    case VH_I8: case VH_I16: case VH_I32:
      if (sum) *sop = SOP_ADD32;
      else if (mx) { *sop = SOP_MAX_I32; *ident = (uint32_t)(elem == VH_I8 ? INT8_MIN : elem == VH_I16 ? INT16_MIN : INT32_MIN); }
      else { *sop = SOP_MIN_I32; *ident = (uint32_t)(elem == VH_I8 ? INT8_MAX : elem == VH_I16 ? INT16_MAX : INT32_MAX); }
      return 0;
    case VH_U64:
      if (sum) *sop = SOP_ADD64;
      else if (mx) { *sop = SOP_MAX_U64; *ident = 0; }
      else { *sop = SOP_MIN_U64; *ident = ~0ull; }
      return 0;
    case VH_I64:
      if (sum) *sop = SOP_ADD64;
      else if (mx) { *sop = SOP_MAX_I64; *ident = (uint64_t)INT64_MIN; }
      else { *sop = SOP_MIN_I64; *ident = (uint64_t)INT64_MAX; }
      return 0;
    case VH_F32:
      if (sum) *sop = SOP_ADDF32;
      else if (mx) { *sop = SOP_MAX_F32; *ident = fbits(FLT_MIN); }   // reference quirk: cpp_min_value
      else { *sop = SOP_MIN_F32; *ident = fbits(FLT_MAX); }
      return 0;
    case VH_F64:
      if (sum) *sop = SOP_ADDF64;
      else if (mx) { *sop = SOP_MAX_F64; *ident = dbits(DBL_MIN); }
      else { *sop = SOP_MIN_F64; *ident = dbits(DBL_MAX); }
      return 0;
    default: return -1;
  }
}

// typed comparison a <= b of two literals/stats given as raw bits
static bool typed_le(int elem, uint64_t a_bits, uint64_t b_bits) {
  if (elem == VH_F32) { float a, b; uint32_t x = (uint32_t)a_bits, y = (uint32_t)b_bits; memcpy(&a, &x, 4); memcpy(&b, &y, 4); return a <= b; }
  if (elem == VH_F64) { double a, b; memcpy(&a, &a_bits, 8); memcpy(&b, &b_bits, 8); return a <= b; }
  return order_key_of_bits(elem, a_bits) <= order_key_of_bits(elem, b_bits);
}

// SegmentSkipBuilder (src/codegen/query/filter.cc:263-335) for one segment.
static bool segment_passes(const vh_table* t, const vh_plan* p, uint32_t seg) {
  if (p->nfilter <= 0) return true;
  std::vector<char> st((size_t)p->nfilter + 1);
  int sp = 0;
  for (int i = 0; i < p->nfilter; ++i) {
    const vh_filter_node& n = p->filter[i];
    switch (n.kind) {
      case VH_F_TRUE: st[sp++] = true; break;
      case VH_F_AND: { bool a = st[--sp]; for (int k = 1; k < n.count; ++k) a = a & st[--sp]; st[sp++] = a; } break;
      case VH_F_OR: { bool a = st[--sp]; for (int k = 1; k < n.count; ++k) a = a | st[--sp]; st[sp++] = a; } break;
      default: {
        const VhColumn& c = t->cols[n.col];
        bool r = true;
        if (c.kind == VH_DIM_NUMERIC || c.kind == VH_DIM_TIME) {
          const VhSegStat& s = t->stats[n.col][seg];
          const uint64_t dmin = bits_of_order_key(c.elem, std::min(s.lo, stat_min_identity_key(c.elem)));
          const uint64_t dmax = bits_of_order_key(c.elem, std::max(s.hi, stat_max_identity_key(c.elem)));
          if (n.kind == VH_F_REL) {
            const uint64_t v = p->lits[n.lit].u64;
            switch (n.op) {
              case VH_OP_EQ: r = typed_le(c.elem, dmin, v) & typed_le(c.elem, v, dmax); break;
              case VH_OP_LT: case VH_OP_LE: r = typed_le(c.elem, dmin, v); break;
              case VH_OP_GT: case VH_OP_GE: r = typed_le(c.elem, v, dmax); break;
              default: r = true; break;
            }
          } else {  // IN and NOT IN alike (the reference does not look at equal())
            r = false;
            for (int k = 0; k < n.count; ++k) {
              const uint64_t v = p->lits[n.lit + k].u64;
              r = r | (typed_le(c.elem, dmin, v) & typed_le(c.elem, v, dmax));
            }
            if (n.count == 0) r = false;
          }
        }
        st[sp++] = r;
      } break;
    }
  }
  return st[0];
}

// Fraction of rows that pass the filter, estimated by running the scan kernel in counting mode over
// the first 16 K rows of up to 64 evenly spaced segments (one extra ~20 us launch + a 64-byte read-back).
// Decides between direct global atomics (cheap per query, ~30-60 G updates/s) and radix-partitioned
// LDS aggregation (two passes over 16 B per survivor, but no global atomics).
static int estimate_selectivity(vh_table* t, VhExec* x, const VhPlanDev& P, const std::vector<VhProgOp>& prog, const std::vector<uint64_t>& lits, uint32_t nseg,
                                double* sel, uint64_t* passed_out = nullptr, uint64_t* sampled_out = nullptr, bool generic = false) {
  const uint32_t kRows = 16384;
  const size_t rows_bytes = ((size_t)std::max<uint32_t>(nseg, 1) * sizeof(uint32_t) + 7) / 8 * 8;
  const size_t need = 256 + 256 + rows_bytes + prog.size() * sizeof(VhProgOp) + lits.size() * sizeof(uint64_t);
  if (need > x->d_sample_bytes) {
    if (x->d_sample) HIP_TRY(hipFree(x->d_sample));
    HIP_TRY(hipMalloc(&x->d_sample, need * 2));
    x->d_sample_bytes = need * 2;
  }
  std::vector<uint32_t> rows(std::max<uint32_t>(nseg, 1), 0);
  const uint32_t stride = std::max<uint32_t>(1, nseg / 64);
  uint64_t sampled = 0;
  for (uint32_t s = 0; s < nseg; s += stride) { rows[s] = std::min<uint32_t>(x->h_segrows[s], kRows); sampled += rows[s]; }
  if (passed_out) *passed_out = 0;
  if (sampled_out) *sampled_out = sampled;
  if (!sampled) { *sel = 0; return VH_OK; }
  VhPlanDev S = P;
  S.ngroup = 0; S.nmetric = 0; S.nbitset = 0; S.G = 1; S.nxcd = 1; S.xcd_stride = 64;
  S.lds_present_off = 0; S.lds_bytes = 16; S.present_carrier = -1;
  S.counters = reinterpret_cast<unsigned long long*>(x->d_sample);
  S.present = reinterpret_cast<uint8_t*>(x->d_sample + 256);
  S.seg_rows = reinterpret_cast<const uint32_t*>(x->d_sample + 512);
  S.nseg = nseg; S.unit_rows = kRows; S.units_per_seg = 1; S.total_units = nseg;
  hipStream_t st = x->stream();
  HIP_TRY(hipMemsetAsync(x->d_sample, 0, 512, st));
  HIP_TRY(hipMemcpyAsync(x->d_sample + 512, rows.data(), nseg * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  S.prog = reinterpret_cast<const VhProgOp*>(x->d_sample + 512 + rows_bytes);
  S.lits = reinterpret_cast<const uint64_t*>(x->d_sample + 512 + rows_bytes + prog.size() * sizeof(VhProgOp));
  HIP_TRY(hipMemcpyAsync(const_cast<VhProgOp*>(S.prog), prog.data(), prog.size() * sizeof(VhProgOp), hipMemcpyHostToDevice, st));
  if (!lits.empty()) HIP_TRY(hipMemcpyAsync(const_cast<uint64_t*>(S.lits), lits.data(), lits.size() * sizeof(uint64_t), hipMemcpyHostToDevice, st));
  const size_t qbytes = (size_t)16 * VhScanCfg<1024>::kQueueCap * sizeof(uint32_t);
  // (generic: predicate columns of other widths than 4 bytes — plans only the per-query compiled kernels run register-resident)
  if (generic) vh_launch_scan_generic(VH_MODE_DENSE_LDS, S, (int)std::min<uint32_t>(nseg, (uint32_t)g_ctx.num_cu), 16 + qbytes, false, st);
  else vh_launch_scan_fast_lds(S, (int)std::min<uint32_t>(nseg, (uint32_t)g_ctx.num_cu), 16 + qbytes, false, st);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(x->h_counters + 8, S.counters, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  *sel = (double)x->h_counters[8] / (double)sampled;
  if (passed_out) *passed_out = x->h_counters[8];
  return VH_OK;
}

struct ScratchPlan {
  size_t off = 0;
  size_t take(size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; }
};

static int fill_states(void* p, uint64_t n, int bytes, uint64_t ident, hipStream_t s) {
  if (ident == 0) { HIP_TRY(hipMemsetAsync(p, 0, n * bytes, s)); return VH_OK; }
  const int grid = (int)std::min<uint64_t>(2048, (n + 255) / 256);
  if (bytes == 4) hipLaunchKernelGGL(fill_kernel<uint32_t>, dim3(grid), dim3(256), 0, s, (uint32_t*)p, n, (uint32_t)ident);
  else hipLaunchKernelGGL(fill_kernel<uint64_t>, dim3(grid), dim3(256), 0, s, (uint64_t*)p, n, ident);
  return VH_OK;
}

// Multi-GPU (vh_query_agg_sharded): what a rank's planner looks at, exchanged between the ranks ...
struct VhSummary {
  uint64_t klo[VH_MAX_GROUP], khi[VH_MAX_GROUP];   // order keys of a group column over the segments this rank will scan; klo > khi: none
  uint64_t rows_to_scan, probe_passed, probe_sampled;
  uint64_t cap_override, part_override;            // re-plan requests of the previous attempt (VhReplan)
  uint32_t force_hash, no_part, fatal, pad;
};
// ... and what every rank plans with instead of its own view, so that all of them build the same table organisation.
struct VhAgreed {
  uint64_t klo[VH_MAX_GROUP], khi[VH_MAX_GROUP];
  uint64_t rows_to_scan;      // over all ranks
  uint64_t rows_max;          // the largest shard: what one rank's kernels will see
  double sel;
};

// One aggregate query on its way to the device. query_launch_locked() runs the steps in order; each step reads what the earlier
// ones decided from the members below. `done`: the query has been handed over (or, for a plan-only / summary call, answered) early.
struct QueryBuild {
  // ---- the call
  vh_table* t; VhExec* x; const vh_plan* p; vh_result** out;
  uint64_t hash_capacity_override; bool force_hash; uint64_t part_tuples_override; bool no_part, plan_only;
  VhSummary* summary_out; const VhAgreed* ag; bool device_rows; uint32_t hp_passes_override; bool no_hpart;
  std::unique_ptr<vh_result> holder;      // every early return drops it
  vh_result* r;
  VhPlanDev& P;
  std::vector<VhProgOp>& prog;
  bool done = false;
  // ---- plan shape
  uint32_t nseg = 0; int ncols = 0;
  int slot_of[256];
  int slot_col[VH_MAX_SLOTS];                    // table column behind a slot (-1: a narrow copy / projection member added later)
  int slot_rec[VH_MAX_SLOTS], slot_recoff[VH_MAX_SLOTS];   // payload projection a slot reads from (-1: a column arena) and the member's offset in its record
  int slot_stored[VH_MAX_SLOTS];                 // ... and the bytes it takes there (0: the element size)
  uint64_t bytes_per_row = 0;
  bool fast_ok = false;
  int pred_col[VH_MAX_PRED] = {-1, -1, -1, -1};        // table column behind predicate slot k of the register-resident kernels
  int pred_wide_slot[VH_MAX_PRED] = {-1, -1, -1, -1};  // its 4-byte arena's slot when the plan was pointed at a narrow copy
  VhJitShape jshape;
  int jit_pred_col[VJ_MAX_PRED];
  bool jit_try = false;
  uint64_t rows_to_scan = 0;
  std::vector<uint32_t> live;                    // segments with rows to scan
  uint64_t probe_passed = 0, probe_sampled = 0;
  bool dense_ok = false;
  uint64_t G = 1;                                // dense group-id space
  int bitset_col[VH_MAX_BITSET];
  int metric_col[VH_MAX_METRIC];                 // table column behind device metric j (-1: virtual row id / bitset)
  uint64_t bitset_ids[VH_MAX_BITSET] = {};       // ids stored in the scanned segments, per bitset metric
  // ---- organisation
  int mode = 0;
  size_t lds_table = 0;
  bool fast = false, fastj = false, lanes = false;
  uint64_t part_tuple_cap = 0;
  int nxcd = 1, part_bpp = 1;
  uint64_t capacity = 0;
  bool hpart = false;
  uint64_t hp_tuple_cap = 0;
  int hp_units = 1;                 // 16-byte units per tuple of the hashed partitioning: 2 when the tuples carry the ids of a bitset metric
  bool hp_pack = false;             // ... or 1 all the same: payload, two ids and their count packed into the tuple's second word (VhHpArgs::pk)
  int hp_pbits = 0, hp_idbits = 0;
  int hp_bpp = 1;
  uint32_t hp_chunk = 256;
  bool packed = false, packed_compressed = false;
  VhJitKernel* jk = nullptr;
  int jit_block = 256;
  // ---- work decomposition, scratch
  int BLOCK = 256, grid = 1;
  size_t o_segrows = 0, zero_begin = 0, zero_end = 0;
  uint64_t table_n = 0;
  size_t rec_off[VH_MAX_METRIC] = {};
  int split_bpp = 1;
  struct HpOff { size_t ta = 0, fa = 0, ga = 0, tb = 0, fb = 0, gb = 0, meta = 0; uint64_t maxa = 0, maxb = 0; } hpo[2];
  size_t o_hpargs = 0, hp_meta_bytes = 0;
  char* S = nullptr;

  QueryBuild(vh_table* t_, VhExec* x_, const vh_plan* p_, vh_result** out_, uint64_t hash_capacity_override_, bool force_hash_,
             uint64_t part_tuples_override_, bool no_part_, bool plan_only_, VhSummary* summary_out_, const VhAgreed* ag_,
             bool device_rows_, uint32_t hp_passes_override_, bool no_hpart_)
      : t(t_), x(x_), p(p_), out(out_), hash_capacity_override(hash_capacity_override_), force_hash(force_hash_),
        part_tuples_override(part_tuples_override_), no_part(no_part_), plan_only(plan_only_), summary_out(summary_out_), ag(ag_),
        device_rows(device_rows_), hp_passes_override(hp_passes_override_), no_hpart(no_hpart_),
        holder(new vh_result()), r(holder.get()), P(r->plan), prog(r->h_prog) {}

  int slot(int col);                              // the plan's slot of a table column (-1: none left / bad column, -2: a bitset metric)
  int probed_selectivity(double* sel);
  void scan_dispatch(int grid_, int* occ);        // one place decides which scan kernel runs; with `occ` it only asks how many of its blocks fit a CU
  // the steps, in order
  int shape_filter();
  int snapshot_segments();
  int shape_groups();
  int shape_metrics();
  int choose_organisation();
  int plan_hashed_partitioning();
  int choose_projection();
  int compile_kernel();
  int decompose_work();
  int layout_scratch();
  int launch();
};

static int query_launch_locked(vh_table* t, VhExec* x, const vh_plan* p, vh_result** out, uint64_t hash_capacity_override,
                               bool force_hash, uint64_t part_tuples_override, bool no_part, bool plan_only, VhSummary* summary_out,
                               const VhAgreed* ag, bool device_rows, uint32_t hp_passes_override, bool no_hpart);

int QueryBuild::slot(int col) {
  if (col < 0 || col >= ncols || col >= 256) return -1;
  if (slot_of[col] >= 0) return slot_of[col];
  if (P.nslots >= VH_MAX_SLOTS) return -1;
  const VhColumn& c = t->cols[col];
  if (is_bitset_elem(c.elem)) return -2;
  slot_of[col] = P.nslots;
  slot_col[P.nslots] = col;
  P.colbase[P.nslots] = c.base;
  P.colstride[P.nslots] = c.stride;
  P.colpitch[P.nslots] = (uint32_t)c.esize;
  bytes_per_row += c.esize;
  return P.nslots++;
}

int QueryBuild::shape_filter() {
  int rc = VH_OK; (void)rc;
  // ---------------- validate
  if (p->nfilter < 0 || p->nlits < 0 || p->ngroups < 0 || p->nmetrics < 0 || p->nhaving < 0)
    return vh_fail(VH_E_INVALID, "plan has a negative count");
  if ((p->nfilter && !p->filter) || (p->nlits && !p->lits) || (p->ngroups && !p->groups) || (p->nmetrics && !p->metrics) || (p->nhaving && !p->having))
    return vh_fail(VH_E_INVALID, "plan has a count without its array");
  if (p->nlits > VH_MAX_LITS) return vh_fail(VH_E_UNSUPPORTED, "filter has %d literals (max %d)", p->nlits, VH_MAX_LITS);
  if (p->ngroups > VH_MAX_GROUP) return vh_fail(VH_E_UNSUPPORTED, "%d group columns (max %d)", p->ngroups, VH_MAX_GROUP);
  if (p->nmetrics > VH_MAX_METRIC - 1) return vh_fail(VH_E_UNSUPPORTED, "%d metrics in one pass (max %d; vh_query_agg splits wider queries into passes)", p->nmetrics, VH_MAX_METRIC - 1);
  nseg = p->seg_rows ? p->nseg : t->nseg;
  if (nseg > t->nseg) return vh_fail(VH_E_INVALID, "plan snapshots %u segments, table mirrors %u", nseg, t->nseg);
  ncols = (int)t->cols.size();

  r->table = t;
  memset(&P, 0, sizeof(P));

  // ---------------- column slots
  for (int i = 0; i < 256; ++i) slot_of[i] = -1;
  for (int i = 0; i < VH_MAX_SLOTS; ++i) { slot_col[i] = -1; slot_rec[i] = -1; slot_recoff[i] = 0; slot_stored[i] = 0; }

  // ---------------- filter program (+ stack depth check)
  fast_ok = !(p->flags & VH_PLAN_NO_FAST);
  int depth = 0, maxdepth = 0;
  std::vector<size_t> seg_start;          // where the piece of program behind each value on the (simulated) stack begins
  for (int i = 0; i < p->nfilter; ++i) {
    const vh_filter_node& n = p->filter[i];
    VhProgOp o{};
    if (n.kind == VH_F_REL || n.kind == VH_F_IN || n.kind == VH_F_TRUE) seg_start.push_back(prog.size());
    o.set_kind((uint8_t)n.kind); o.set_op((uint8_t)n.op); o.set_count((uint8_t)std::min(n.count, 255));
    if (n.kind == VH_F_REL || n.kind == VH_F_IN) {
      int s = slot(n.col);
      if (s == -2) {     // a bitset metric: the predicate compares the row's cardinality (offsets of the CSR mirror)
        s = -1;
        for (size_t k = 0; k < r->filter_bitset_cols.size(); ++k) if (r->filter_bitset_cols[k] == n.col) s = (int)k;
        if (s < 0) {
          if (r->filter_bitset_cols.size() >= VH_MAX_BITSET) return vh_fail(VH_E_UNSUPPORTED, "more than %d bitset metrics in one filter", VH_MAX_BITSET);
          s = (int)r->filter_bitset_cols.size();
          r->filter_bitset_cols.push_back(n.col);
          bytes_per_row += 8;
        }
        fast_ok = false;
      }
      if (s < 0) { return vh_fail(VH_E_INVALID, "filter node %d: bad column %d", i, n.col); }
      const int cnt = n.kind == VH_F_REL ? 1 : n.count;
      if (n.lit < 0 || n.count < 0 || n.lit + cnt > p->nlits) { return vh_fail(VH_E_INVALID, "filter node %d: literal range", i); }
      o.set_slot((uint8_t)s); o.set_type((uint8_t)t->cols[n.col].elem); o.set_lit((uint16_t)n.lit);
      // fast path bookkeeping: distinct 4-byte predicate columns
      if (is_bitset_elem(t->cols[n.col].elem) || vh_elem_size(t->cols[n.col].elem) != 4) fast_ok = false;
      if (fast_ok) {
        int ps = -1;
        for (int k = 0; k < P.npred; ++k) if (P.pred_slot[k] == s) ps = k;
        if (ps < 0) { if (P.npred < VH_MAX_PRED) { ps = P.npred; pred_col[P.npred] = n.col; P.pred_width[P.npred] = 4; P.pred_slot[P.npred++] = (uint8_t)s; } else fast_ok = false; }
        o.set_pslot((uint8_t)std::max(ps, 0));
      }
      if (n.kind == VH_F_IN && n.count > 255) {
        // a long IN list (the reference emits one comparison per value, filter.cc:223-241): chunks of 255 literals, folded
        // pairwise — OR of the chunks for IN, AND for NOT IN — so the mask stack grows by one entry only
        for (int first = 0; first < n.count; first += 255) {
          VhProgOp c = o;
          c.set_count((uint8_t)std::min(255, n.count - first)); c.set_lit((uint16_t)(n.lit + first));
          prog.push_back(c);
          if (first) { VhProgOp f{}; f.set_kind(n.op ? VH_F_OR : VH_F_AND); f.set_count(2); prog.push_back(f); }
        }
        ++depth;
        continue;
      }
      ++depth;
    } else if (n.kind == VH_F_TRUE) {
      ++depth;
    } else if (n.kind == VH_F_AND || n.kind == VH_F_OR) {
      if (n.count < 1 || n.count > depth) { return vh_fail(VH_E_INVALID, "filter node %d: operand count %d", i, n.count); }
      if (n.count > 3) {
        // Bitwise & and | are associative: a composite of n operands is folded pairwise (a b OP c OP ...), so the mask stack of
        // the kernels holds one entry per NESTING level, not per operand — an OR of 120 comparisons needs depth 2, not 120.
        // The operands are the last n contiguous pieces of the program emitted so far (seg_start remembers where each begins).
        std::vector<VhProgOp> folded;
        const size_t first = seg_start.size() - (size_t)n.count;
        folded.reserve(prog.size() - seg_start[first] + (size_t)n.count);
        for (int k = 0; k < n.count; ++k) {
          const size_t b = seg_start[first + k], e = k + 1 < n.count ? seg_start[first + k + 1] : prog.size();
          folded.insert(folded.end(), prog.begin() + b, prog.begin() + e);
          if (k) { VhProgOp f{}; f.set_kind((uint8_t)n.kind); f.set_count(2); folded.push_back(f); }
        }
        prog.resize(seg_start[first]);
        prog.insert(prog.end(), folded.begin(), folded.end());
        seg_start.resize(first + 1);
        depth -= n.count - 1;
        continue;
      }
      seg_start.resize(seg_start.size() - (size_t)n.count + 1);
      depth -= n.count - 1;
    } else { return vh_fail(VH_E_INVALID, "filter node %d: kind %d", i, n.kind); }
    prog.push_back(o);
  }
  if (p->nfilter == 0) { VhProgOp o{}; o.set_kind(VH_F_TRUE); prog.push_back(o); depth = 1; }
  P.nprog = (int32_t)prog.size();
  if (depth != 1) { return vh_fail(VH_E_INVALID, "filter program leaves %d values on the stack", depth); }
  for (int d = 0, k = 0; k < P.nprog; ++k) {     // depth of the program as the kernels will run it
    const int kind = prog[k].kind();
    d += (kind == VH_F_AND || kind == VH_F_OR) ? 1 - (int)prog[k].count() : 1;
    maxdepth = std::max(maxdepth, d);
  }
  if (maxdepth > VH_MAX_STACK) { return vh_fail(VH_E_UNSUPPORTED, "filter needs stack depth %d (max %d)", maxdepth, VH_MAX_STACK); }
  {   // conjunctions / disjunctions of leaves — most filters — need no stack in the register-resident kernels (vh_eval_filter_fast)
    bool leaves = true;
    for (int k = 0; k + 1 < P.nprog; ++k) leaves &= prog[k].kind() != VH_F_AND && prog[k].kind() != VH_F_OR;
    const int last = prog[P.nprog - 1].kind();
    P.prog_flat = 0;
    if (P.nprog == 1 && last != VH_F_AND && last != VH_F_OR) P.prog_flat = 1;
    else if (leaves && P.nprog > 1 && (last == VH_F_AND || last == VH_F_OR) && (int)prog[P.nprog - 1].count() == P.nprog - 1) P.prog_flat = last == VH_F_AND ? 1 : 2;
  }
  r->h_lits.resize(std::max(p->nlits, 0));
  for (int i = 0; i < p->nlits; ++i) r->h_lits[i] = p->lits[i].u64;
  if (prog.size() <= VH_INLINE_PROG && r->h_lits.size() <= VH_INLINE_LITS) {   // the register-resident kernels read the program from the kernel arguments
    memcpy(P.iprog, prog.data(), prog.size() * sizeof(VhProgOp));
    memcpy(P.ilits, r->h_lits.data(), r->h_lits.size() * sizeof(uint64_t));
  } else fast_ok = false;                                                       // long programs (IN lists of hundreds of values): the generic kernel

  // ---------------- per-query compiled scan kernel (vh_jit.hip): which predicate columns it would hold packed in registers
  // Eligible so far: every leaf compares a fixed-width column, the program and its literals fit the kernel arguments, the packed
  // columns fit VJ_MAX_NV registers. The table organisation decides the rest further down.
  jit_try = vh_jit_policy() != VH_JIT_OFF && !(p->flags & (VH_PLAN_NO_JIT | VH_PLAN_NO_FAST)) && prog.size() <= VH_INLINE_PROG && r->h_lits.size() <= VH_INLINE_LITS;
  if (jit_try) {
    jshape.prog = prog;
    int nv = 0;
    for (VhProgOp& o : jshape.prog) {
      if (o.kind() != VH_F_REL && o.kind() != VH_F_IN) continue;
      const int es = vh_elem_size((int)o.type());
      if (!es) { jit_try = false; break; }                       // a bitset metric's cardinality: the generic kernel
      int ps = -1;
      for (int k = 0; k < jshape.npred; ++k) if (jshape.pred[k].slot == (int)o.slot()) ps = k;
      if (ps < 0) {
        if (jshape.npred >= VJ_MAX_PRED || nv + VH_SUBSTEPS * es > VJ_MAX_NV) { jit_try = false; break; }
        ps = jshape.npred++;
        jshape.pred[ps] = VhJitPred{(int)o.slot(), (int)o.type(), es};
        jit_pred_col[ps] = slot_col[o.slot()];
        nv += VH_SUBSTEPS * es;
      }
      o.set_pslot((uint8_t)ps);
    }
    jshape.nlits = (int)r->h_lits.size();
  }

  // Narrow copies of predicate columns (vh_table_narrow; built unasked for a column the third query filters on): the
  // register-resident kernels — and the selectivity probe, which is one of them — stream those instead of the 4-byte arenas.
  if ((fast_ok || jit_try) && !(p->flags & VH_PLAN_NO_NARROW)) {
    const int auto_after = g_preparing ? 1 : knobs().auto_narrow;     // 0: never unasked
    std::map<int, int> narrow_slot;          // column -> slot of its narrow copy (looked up, and counted, once per query)
    auto narrow_for = [&](int col) -> int {
      auto hit = narrow_slot.find(col);
      if (hit != narrow_slot.end()) return hit->second;
      bool have = false;
      for (auto& nw : t->narrows) have |= nw->col == col;
      const int nwidth = have ? 0 : narrow_width_for(t, col, t->nseg);
      // (every query that filters on the column reads it in full, whatever passes: the copy pays from the first query that uses it on)
      if (!have && auto_after > 0 && nwidth && ++t->pred_seen[col] >= (uint32_t)auto_after) {
        size_t free_b = 0, total_b = 0;
        const size_t need = (size_t)t->cap_seg * ((t->segment_rows + 255) / 256 * 256) * (size_t)nwidth;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > need + total_b / 4) (void)table_narrow_locked(t, col, true);
        else t->pred_seen[col] = 0;
      }
      int ns = -1;
      VhNarrow* nw = narrow_usable(t, col, nseg);
      if (nw && P.nslots < VH_MAX_SLOTS) {
        P.colbase[P.nslots] = nw->base; P.colstride[P.nslots] = nw->stride; P.colpitch[P.nslots] = (uint32_t)nw->width;
        ns = P.nslots++;
      }
      return narrow_slot[col] = ns;
    };
    if (fast_ok)
      for (int k = 0; k < P.npred; ++k) {
        const int ns = narrow_for(pred_col[k]);
        if (ns < 0) continue;
        pred_wide_slot[k] = P.pred_slot[k];
        P.pred_slot[k] = (uint8_t)ns;
        P.pred_width[k] = (uint8_t)P.colpitch[ns];
      }
    if (jit_try)
      for (int k = 0; k < jshape.npred; ++k) {
        if (jit_pred_col[k] < 0 || t->cols[jit_pred_col[k]].elem != VH_U32) continue;
        const int ns = narrow_for(jit_pred_col[k]);
        if (ns < 0) continue;
        jshape.pred[k].slot = ns;
        jshape.pred[k].width = (int)P.colpitch[ns];
      }
  }
  return VH_OK;
}

int QueryBuild::snapshot_segments() {
  int rc = VH_OK; (void)rc;
  // ---------------- segments: snapshot + skip
  // one pinned staging block [segment snapshot | program | literals] -> one upload per query
  const size_t seg_words = ((size_t)std::max<uint32_t>(nseg, 1) + 1) / 2 * 2;
  const size_t plan_words = seg_words + 2 * (prog.size() + r->h_lits.size());
  rc = ensure_segrows(x, plan_words);
  if (rc) { return rc; }
  memcpy(x->h_segrows + seg_words, prog.data(), prog.size() * sizeof(VhProgOp));
  memcpy(x->h_segrows + seg_words + 2 * prog.size(), r->h_lits.data(), r->h_lits.size() * sizeof(uint64_t));
  r->plan_words = plan_words; r->seg_words = seg_words;
  uint64_t scanned_recs = 0, scanned_segments = 0;
  for (uint32_t s = 0; s < nseg; ++s) {
    uint64_t rows = p->seg_rows ? p->seg_rows[s] : t->seg_rows[s];
    if (rows > t->seg_rows[s]) { return vh_fail(VH_E_INVALID, "segment %u: snapshot %llu rows > mirrored %llu", s, (unsigned long long)rows, (unsigned long long)t->seg_rows[s]); }
    scanned_recs += rows;
    const bool keep = segment_passes(t, p, s);
    if (keep) { ++scanned_segments; rows_to_scan += rows; if (rows) live.push_back(s); }
    x->h_segrows[s] = keep ? (uint32_t)rows : 0u;
  }
  r->info.scanned_recs = scanned_recs;
  r->info.scanned_segments = scanned_segments;
  // a compile pays off for scans of some size (or when asked for): small tables keep the interpreting kernels
  if (jit_try && !(p->flags & VH_PLAN_FORCE_JIT) && vh_jit_policy() != VH_JIT_FORCE && (ag ? ag->rows_max : rows_to_scan) < vh_jit_min_rows()) jit_try = false;
  if (plan_only) {   // vh_query_select: filter program, column slots and the segment snapshot are all it shares
    P.nseg = nseg;
    r->info.algorithmic_bytes = rows_to_scan * bytes_per_row;
    *out = holder.release();
    done = true;
    return VH_OK;
  }
  return VH_OK;
}

// selectivity of the filter, from a one-launch probe; it only depends on the filter and the rows, so it is cached
// until the table changes. Sharded queries plan with the estimate all ranks agreed on.
int QueryBuild::probed_selectivity(double* sel) {
  if (ag) { *sel = ag->sel; return VH_OK; }
  if (p->nfilter == 0) { *sel = 1.0; probe_passed = probe_sampled = rows_to_scan; return VH_OK; }   // no filter: every row passes
  std::string key((const char*)prog.data(), sizeof(VhProgOp) * prog.size());
  key.append((const char*)r->h_lits.data(), sizeof(uint64_t) * r->h_lits.size());
  key += "|" + std::to_string(nseg) + "|" + std::to_string(rows_to_scan) + "|" + std::to_string(t->sync_epoch);
  auto hit = t->sel_cache.find(key);
  if (hit != t->sel_cache.end()) { probe_passed = hit->second.first; probe_sampled = hit->second.second; *sel = probe_sampled ? (double)probe_passed / (double)probe_sampled : 0.0; return VH_OK; }
  const int prc = estimate_selectivity(t, x, P, r->h_prog, r->h_lits, nseg, sel, &probe_passed, &probe_sampled, !fast_ok);
  if (prc) return prc;
  if (t->sel_cache.size() > 256) t->sel_cache.clear();
  t->sel_cache[key] = std::make_pair(probe_passed, probe_sampled);
  return VH_OK;
}

int QueryBuild::shape_groups() {
  int rc = VH_OK; (void)rc;
  // ---------------- group columns
  P.ngroup = p->ngroups;
  if (summary_out) for (int i = 0; i < VH_MAX_GROUP; ++i) { summary_out->klo[i] = ~0ull; summary_out->khi[i] = 0; }
  r->device_rows = device_rows;
  dense_ok = !force_hash && !(p->flags & VH_PLAN_FORCE_HASH);
  int key_bits_total = 0;
  for (int i = 0; i < p->ngroups; ++i) {
    const vh_group_col& gc = p->groups[i];
    const int s = slot(gc.col);
    if (s < 0 || !is_dim(t->cols[gc.col].kind)) { return vh_fail(VH_E_INVALID, "group column %d: bad column %d", i, gc.col); }
    const VhColumn& c = t->cols[gc.col];
    if (gc.nrollup < 0 || gc.nrollup > VH_MAX_ROLLUP) { return vh_fail(VH_E_UNSUPPORTED, "group column %d: %d rollup rules", i, gc.nrollup); }
    if (gc.granularity > VH_T_NONE) { return vh_fail(VH_E_INVALID, "group column %d: granularity %d", i, gc.granularity); }
    for (int k = 0; k < gc.nrollup; ++k)
      if (gc.rollup_unit[k] < VH_T_YEAR || gc.rollup_unit[k] > VH_T_SECOND) { return vh_fail(VH_E_INVALID, "group column %d: rollup unit %d", i, gc.rollup_unit[k]); }
    VhGroupDev& g = P.g[i];
    g.set_slot((uint16_t)s); g.set_type((uint8_t)c.elem);
    g.set_gran((uint8_t)(gc.granularity < 0 ? VH_T_NONE : gc.granularity));
    g.set_nroll((uint8_t)gc.nrollup); g.set_micro((uint8_t)gc.micro);

    if (g.gran() == VH_T_WEEK) { return vh_fail(VH_E_UNSUPPORTED, "week granularity: the reference has no Truncator::trunc<WEEK> (src/util/time.h:57-89)"); }
    for (int k = 0; k < gc.nrollup; ++k) {
      if (gc.rollup_unit[k] == VH_T_WEEK) { return vh_fail(VH_E_UNSUPPORTED, "week rollup granularity is not supported by the reference"); }
      g.set_roll_unit(k, (uint8_t)gc.rollup_unit[k]); g.roll_before[k] = gc.rollup_before[k];
    }
    const bool timey = g.gran() != VH_T_NONE || g.nroll();
    if (timey && c.kind != VH_DIM_TIME) { return vh_fail(VH_E_INVALID, "group column %d: truncation on a non-time dimension", i); }
    r->group_elem.push_back(c.elem);
    key_bits_total += c.esize * 8;
    // dense digit range
    uint64_t lo = 0, extent = 0;
    if (c.elem == VH_F32 || c.elem == VH_F64 || timey) {
      dense_ok = false;
    } else if (gc.cardinality > 0 && (c.kind == VH_DIM_STRING || c.kind == VH_DIM_BOOLEAN)) {
      lo = 0; extent = gc.cardinality;
    } else {
      uint64_t klo = ~0ull, khi = 0;
      for (uint32_t sgi : live) { klo = std::min(klo, t->stats[gc.col][sgi].lo); khi = std::max(khi, t->stats[gc.col][sgi].hi); }
      if (summary_out) { summary_out->klo[i] = klo; summary_out->khi[i] = khi; }
      if (ag) { klo = ag->klo[i]; khi = ag->khi[i]; }      // the range over ALL ranks' segments: identically indexed tables everywhere
      if (klo > khi) { lo = 0; extent = 1; }
      else {
        lo = bits_of_order_key(c.elem, klo);
        const uint64_t span = khi - klo;
        extent = span == ~0ull ? 0 : span + 1;
        if (extent == 0) dense_ok = false;
      }
    }
    g.lo = lo; g.extent = extent;
    if (dense_ok) {
      if (extent == 0 || G > (1ull << 40) / std::max<uint64_t>(extent, 1)) dense_ok = false;
      else G *= extent;
    }
  }
  if (summary_out) {       // sharded queries, first half: report and stop
    summary_out->rows_to_scan = rows_to_scan;
    double sel = 0;
    if (fast_ok || jit_try || p->nfilter == 0) { rc = probed_selectivity(&sel); if (rc) return rc; }
    summary_out->probe_passed = probe_passed; summary_out->probe_sampled = probe_sampled;
    done = true;
    return VH_OK;
  }
  const uint64_t plan_rows = ag ? ag->rows_to_scan : rows_to_scan;
  const uint64_t dense_limit = std::max<uint64_t>(4096, std::min<uint64_t>(1ull << 24, plan_rows * 4));
  if (G > dense_limit) dense_ok = false;
  if (dense_ok) {
    uint64_t stride = 1;
    for (int i = p->ngroups - 1; i >= 0; --i) { P.g[i].stride = stride; stride *= P.g[i].extent; }
  } else {
    // pack key columns into u64 words, widest first within a word, never straddling
    int word = 0, used = 0;
    for (int i = 0; i < p->ngroups; ++i) {
      const int bits = vh_elem_size(P.g[i].type()) * 8;
      if (used + bits > 64) { ++word; used = 0; }
      P.g[i].set_key_word((uint8_t)word); P.g[i].set_key_shift((uint8_t)used);
      used += bits;
    }
    P.key_words = p->ngroups ? word + 1 : 1;
    if (P.key_words > VH_KEY_WORDS) { return vh_fail(VH_E_UNSUPPORTED, "group key of %d bits is too wide", key_bits_total); }
  }
  return VH_OK;
}

int QueryBuild::shape_metrics() {
  int rc = VH_OK; (void)rc;
  // ---------------- metrics
  P.nmetric = 0;
  bool has_avg = false, has_count = false;
  for (int j = 0; j < VH_MAX_METRIC; ++j) metric_col[j] = -1;
  uint64_t bitset_ids_before = 0;
  uint64_t pair_cap = 0;
  for (int j = 0; j < p->nmetrics; ++j) {
    const int col = p->metrics[j];
    if (col == VH_COL_ROWID) {   // virtual column: storage position of the row, aggregated with MIN (first occurrence)
      VhMetricDev& m = P.m[P.nmetric];
      m.set_slot(VH_SLOT_ROWID); m.set_type(VH_U64); m.set_sop(SOP_MIN_U64); m.ident = ~0ull;
      r->user_metric.push_back(P.nmetric++);
      r->metric_elem.push_back(VH_U64);
      continue;
    }
    if (col < 0 || col >= ncols || is_dim(t->cols[col].kind)) { return vh_fail(VH_E_INVALID, "metric %d: bad column %d", j, col); }
    const VhColumn& c = t->cols[col];
    if (c.kind == VH_METRIC_BITSET) {
      if (P.nbitset >= VH_MAX_BITSET) { return vh_fail(VH_E_UNSUPPORTED, "more than %d bitset metrics in one query", VH_MAX_BITSET); }
      for (uint32_t sgi : live) {
        if (!c.bs_offsets[sgi]) { return vh_fail(VH_E_INVALID, "bitset column %d of segment %u was never synced", col, sgi); }
        pair_cap += c.bs_nvalues[sgi];
      }
      bitset_col[P.nbitset] = col;
      bitset_ids[P.nbitset] = pair_cap - bitset_ids_before;
      bitset_ids_before = pair_cap;
      P.bs_wide[P.nbitset] = c.elem == VH_BITSET64;
      // a set of 32-bit ids has at most 2^32 - 1 distinct members: its cardinality fits 32 bits, and a caller that says so
      // (VH_PLAN_CARD32) gets the column that narrow — a third less to deliver for C5's 35 M groups
      const int card_elem = (p->flags & VH_PLAN_CARD32) && c.elem == VH_BITSET32 ? VH_U32 : VH_U64;
      VhMetricDev& m = P.m[P.nmetric];
      m.set_slot((uint16_t)P.nbitset); m.set_type((uint8_t)card_elem); m.set_sop(SOP_BITSET); m.ident = 0;
      r->user_metric.push_back(P.nmetric++);
      r->metric_elem.push_back(card_elem);
      ++P.nbitset;
      continue;
    }
    const int s = slot(col);
    if (s < 0) { return vh_fail(VH_E_UNSUPPORTED, "too many referenced columns"); }
    int sop; uint64_t ident;
    if (sop_for(c.kind, c.elem, &sop, &ident)) { return vh_fail(VH_E_INVALID, "metric %d: kind %d / elem %d", j, c.kind, c.elem); }
    VhMetricDev& m = P.m[P.nmetric];
    m.set_slot((uint16_t)s); m.set_type((uint8_t)c.elem); m.set_sop((uint8_t)sop); m.ident = ident;
    metric_col[P.nmetric] = col;
    r->user_metric.push_back(P.nmetric++);
    r->metric_elem.push_back(c.elem);
    has_avg |= c.kind == VH_METRIC_AVG; has_count |= c.kind == VH_METRIC_COUNT;
  }
  if (has_avg && !has_count) {
    // hidden uint64_t _count (src/codegen/query/scan.cc:239-241)
    int hc = -1;
    for (int c = 0; c < ncols; ++c) if (t->cols[c].kind == VH_METRIC_HIDDEN_COUNT) hc = c;
    if (hc < 0) { return vh_fail(VH_E_INVALID, "AVG selected without COUNT but the table has no hidden count column"); }
    const int s = slot(hc);
    if (s < 0) { return vh_fail(VH_E_UNSUPPORTED, "too many referenced columns"); }
    metric_col[P.nmetric] = hc;
    VhMetricDev& m = P.m[P.nmetric++];
    m.set_slot((uint16_t)s); m.set_type(VH_U64); m.set_sop(SOP_ADD64); m.ident = 0;
    r->metric_elem.push_back(VH_U64);
    r->info.has_hidden_count = 1;
  }
  r->info.ngroup_cols = p->ngroups;
  r->info.nmetrics = p->nmetrics;
  r->info.algorithmic_bytes = rows_to_scan * bytes_per_row;

  // ---------------- HAVING pushed down to the group-emission kernel
  if (p->nhaving > 0) {
    if (p->nhaving > VH_MAX_HAVING) { return vh_fail(VH_E_UNSUPPORTED, "having has %d nodes (max %d)", p->nhaving, VH_MAX_HAVING); }
    int hdepth = 0, nl = 0;
    for (int i = 0; i < p->nhaving; ++i) {
      const vh_filter_node& n = p->having[i];
      VhProgOp& o = r->hprog[i];
      o.set_kind((uint8_t)n.kind); o.set_op((uint8_t)n.op); o.set_count((uint8_t)n.count);
      if (n.kind == VH_F_REL || n.kind == VH_F_IN) {
        const int cnt = n.kind == VH_F_REL ? 1 : n.count;
        if (n.col < 0 || n.col >= p->ngroups + p->nmetrics || n.lit < 0 || n.lit + cnt > p->nlits || nl + cnt > VH_MAX_HAVING_LITS) {
          return vh_fail(VH_E_INVALID, "having node %d: bad result column / literal range", i);
        }
        if (n.col < p->ngroups) { o.set_slot((uint8_t)n.col); r->htype[i] = (uint8_t)t->cols[p->groups[n.col].col].elem; }
        else {
          const int dj = r->user_metric[n.col - p->ngroups];
          o.set_slot((uint8_t)(p->ngroups + dj));
          const int mcol = p->metrics[n.col - p->ngroups];
          if (mcol == VH_COL_ROWID) r->htype[i] = VH_U64;
          else {
            const VhColumn& mc = t->cols[mcol];
            r->htype[i] = (uint8_t)(mc.kind == VH_METRIC_BITSET ? (mc.elem == VH_BITSET64 ? VH_U64 : VH_U32) : mc.elem);
          }
        }
        o.set_lit((uint16_t)nl);
        for (int k = 0; k < cnt; ++k) r->hlits[nl++] = p->lits[n.lit + k].u64;
        ++hdepth;
      } else if (n.kind == VH_F_TRUE) ++hdepth;
      else if ((n.kind == VH_F_AND || n.kind == VH_F_OR) && n.count >= 1 && n.count <= hdepth) hdepth -= n.count - 1;
      else { return vh_fail(VH_E_INVALID, "having node %d: kind %d / count %d", i, n.kind, n.count); }
      if (hdepth > VH_MAX_STACK) { return vh_fail(VH_E_UNSUPPORTED, "having needs stack depth %d", hdepth); }
    }
    if (hdepth != 1) { return vh_fail(VH_E_INVALID, "having program leaves %d values on the stack", hdepth); }
    r->nhaving = p->nhaving;
  }
  // ---------------- device top-N request (vh_plan.top_*)
  if (p->top_k > 0) {
    if (p->top_col < 0 || p->top_col >= p->ngroups + p->nmetrics) { return vh_fail(VH_E_INVALID, "top_col %d is not a result column", p->top_col); }
    int kind, elem;
    if (p->top_col < p->ngroups) {
      const VhColumn& c = t->cols[p->groups[p->top_col].col];
      kind = c.kind; elem = c.elem;
      r->topk_src = p->top_col; r->topk_src_is_key = true;
    } else {
      const int mcol = p->metrics[p->top_col - p->ngroups];
      kind = mcol == VH_COL_ROWID ? (int)VH_METRIC_MIN : t->cols[mcol].kind;
      r->topk_src = r->user_metric[p->top_col - p->ngroups]; r->topk_src_is_key = false;
      elem = r->metric_elem[r->topk_src];
    }
    if (kind == VH_DIM_STRING || kind == VH_DIM_TIME || kind == VH_DIM_BOOLEAN || kind == VH_METRIC_AVG) {
        return vh_fail(VH_E_UNSUPPORTED, "top-N on a string / time / boolean / AVG column: the reference orders those as formatted strings");
    }
    r->topk = p->top_k; r->topk_elem = elem; r->topk_desc = p->top_desc ? 1 : 0;
    r->topk_cls = (elem == VH_F32 || elem == VH_F64) ? VH_TOPK_FLOAT : VH_TOPK_INT;
  }
  return VH_OK;
}

int QueryBuild::choose_organisation() {
  int rc = VH_OK; (void)rc;
  // ---------------- choose the table organisation
  size_t state_bytes_per_group = 1;  // presence byte
  for (int j = 0; j < P.nmetric; ++j) state_bytes_per_group += vh_sop_bytes(P.m[j].sop());
  if (dense_ok) {
    // LDS layout: [8-byte states][4-byte states][presence bytes], 16 B aligned
    size_t off = 0;
    for (int pass = 0; pass < 2; ++pass)
      for (int j = 0; j < P.nmetric; ++j) {
        const int b = vh_sop_bytes(P.m[j].sop());
        if ((pass == 0) != (b == 8)) continue;
        P.m[j].lds_off = (uint32_t)off; off += G * b;
      }
    off = (off + 7) / 8 * 8;
    P.lds_present_off = (uint32_t)off; off += G;
    lds_table = (off + 15) / 16 * 16;
    const size_t lds_budget = 40 * 1024;
    mode = (lds_table <= lds_budget && !(p->flags & VH_PLAN_FORCE_GLOBAL) && P.nbitset == 0) ? VH_MODE_DENSE_LDS : VH_MODE_DENSE_GLOBAL;
    P.G = G;
    P.lds_bytes = (uint32_t)lds_table;
  } else {
    mode = VH_MODE_HASH;
  }
  fast = fast_ok && P.ngroup <= VH_FAST_COLS && P.nmetric <= VH_FAST_COLS && P.nbitset == 0;   // npred == 0: no filter
  if (P.ngroup > VJ_MAX_COLS || P.nmetric > VJ_MAX_COLS || P.nbitset > 1 || (P.nbitset && P.bs_wide[0])) jit_try = false;
  // (a bitset metric: only the hashed partitioning below has a compiled form for it)
  fastj = fast || (jit_try && P.nbitset == 0);       // a register-resident scan: pre-built, or compiled for this plan shape
  // "Lanes" kernel (no compaction) for small LDS tables when most rows pass: see scan_agg_lanes_kernel
  if (mode == VH_MODE_DENSE_LDS && fast && !(p->flags & VH_PLAN_NO_LANES) && P.ngroup <= VH_LANES_COLS && P.nmetric <= VH_LANES_COLS &&
      P.nmetric >= 1 && rows_to_scan) {
    bool ok = true;
    for (int i = 0; i < P.ngroup; ++i) ok &= vh_elem_size(P.g[i].type()) >= 4 && P.g[i].gran() == VH_T_NONE && P.g[i].nroll() == 0;
    for (int j = 0; j < P.nmetric; ++j) ok &= P.m[j].slot() != VH_SLOT_ROWID && P.m[j].sop() != SOP_BITSET && vh_elem_size(P.m[j].type()) >= 4;
    if (ok) {
      if (p->flags & VH_PLAN_FORCE_LANES) lanes = true;
      else {
        double sel = 0;
        rc = probed_selectivity(&sel);
        if (rc) { return rc; }
        lanes = sel >= 0.25;
      }
    }
  }
  // Global atomics are written through to the fabric one by one; when the group-id space is too big
  // for one LDS table but splits into <= VH_MAX_PART LDS-sized ranges, radix-partition the survivors
  // and aggregate each range in LDS instead (DENSE_PART).
  if (mode == VH_MODE_DENSE_GLOBAL && fastj && !no_part && !(p->flags & (VH_PLAN_NO_PART | VH_PLAN_FORCE_GLOBAL)) && P.nmetric >= 1 && P.nmetric <= VH_FAST_COLS) {
    int shift = 0;
    const size_t part_table_bytes = getenv("VH_PART_TABLE_KB") ? (size_t)atoi(getenv("VH_PART_TABLE_KB")) * 1024 : 128 * 1024;      // (tests shrink it between two queries to force many ranges)   // one 1024-thread block per CU in phase 2 (160 KB LDS)
    // The presence byte rides in a 32-bit SUM state when there is one (SOP_ADD32P: a 64-bit word whose upper half counts rows): two LDS
    // updates per tuple instead of three. (Round 2 took phase 2 for bound by LDS read-modify-writes; in isolation the LDS does 2.5 such
    // tuples per clock and CU — 34 us for C3's 50 M — so what the kernel waits for is its tuples: profiles/r03/NOTES.md.)
    int part_carrier = -1;
    if (!(p->flags & VH_PLAN_NO_CARRIER))
      for (int j = 0; j < P.nmetric && part_carrier < 0; ++j) if (P.m[j].sop() == SOP_ADD32) part_carrier = j;
    const size_t part_state_bytes = part_carrier >= 0 ? state_bytes_per_group - 1 + 4 : state_bytes_per_group;
    while (((size_t)2 << shift) * part_state_bytes <= part_table_bytes) ++shift;
    const uint64_t np = (G + (1ull << shift) - 1) >> shift;
    // more LDS-sized ranges than a wave has lanes: two levels (phase 1 partitions into ceil(np / 64), part_split_kernel splits each 64 ways)
    const bool two_level = np > VH_MAX_PART;
    bool want_part = np <= (uint64_t)VH_MAX_PART * 64 && G <= 0xFFFFFFFFull && !(two_level && (p->flags & VH_PLAN_NO_PART2));
    double sel = 0;
    if (want_part && !part_tuples_override) {       // (a forced plan still sizes its tuple buffer from the estimate)
      rc = probed_selectivity(&sel);
      if (rc) { return rc; }
    }
    if (want_part && !(p->flags & VH_PLAN_FORCE_PART) && !part_tuples_override) {
      bool covered = false;
      if (!(p->flags & VH_PLAN_NO_PACK)) {
        for (auto& pk : t->packs) {
          bool all = true;
          for (int i = 0; i < p->ngroups; ++i) all &= pk->col_index(p->groups[i].col) >= 0;
          for (int j = 0; j < P.nmetric; ++j) if (metric_col[j] >= 0) all &= pk->col_index(metric_col[j]) >= 0;
          covered |= all;
        }
      }
      // Crossover on the C3 table (1 B rows; profiles/r02/NOTES.md). Direct atomics cost 2 x survivors / 23.3 G/s on top of the
      // scan and are the same on every box: 3 % 3.26 ms, 5 % 4.60, 6 % 5.58, 8 % 7.38, 11 % 10.2. Partitioned, with the payload
      // gathered from a projection: 3.40-3.55 / 4.05-4.5 / 4.45-4.85 / 5.2-5.6 / 6.5-6.7 (it varies by +-5 % from run to run: it
      // lives off scattered writes, whose cost depends on where the tuple extents land). Without a projection the gathers
      // dominate both and the switch stays at 5.5 %. A split — some partitions through tuples, the rest straight to the table, so
      // that the atomic unit and the write path work side by side — was measured too: SLOWER than either pure form at every
      // selectivity (5 %: 4.8 ms, 8 %: 6.1, 11 %: 7.3): written-through atomics and tuple stores queue for the same thing.
      want_part = sel >= (covered ? 0.04 : 0.055);
      // Two levels move every tuple once more (16 B read + 16 B written), and still win from the same point on: C3 table,
      // GROUP BY (d5, d2) = 4 M groups, 1 B rows (tools/part2_probe.py, profiles/r02/NOTES.md): 2 % 2.11 vs 1.85 ms direct,
      // 5 % 3.69 vs 4.43, 8 % 5.15 vs 6.99, 25 % 11.8 vs 21.6, 100 % 28.4 vs 85.0.
      // ... and the second phase has a price that does not depend on the rows (every block clears and merges a 120 KB LDS
      // table: ~0.25 ms for 13 partitions), while what partitioning saves grows with the survivors: ~50 ms per 1 G rows and
      // point of selectivity beyond the crossover. A 125 M-row shard of C3 (8 GPUs) stays on direct atomics, 1 G rows do not.
      const double shard_rows = (double)(ag ? ag->rows_max : rows_to_scan);
      if (want_part && shard_rows * (sel - (covered ? 0.03 : 0.045)) < (two_level ? 1e7 : 3.5e6)) want_part = false;      // (C3 shards: 125 M rows 0.733 ms direct vs 0.74-0.78 partitioned, 250 M rows 1.30 vs 1.18)
      // Round 3: with the scan compiled for the plan and two-word tuples leaving as whole lines (vh_part_staged_add) a tuple costs ~10 ps
      // against ~86 ps for its two direct atomics, and phase 2's fixed cost is ~0.1 ms: an eighth of C3 (125 M rows, 6.2 M survivors) runs
      // 0.47 ms partitioned against 0.60 ms direct, a quarter 0.78 against 1.12 (profiles/r03/NOTES.md). From 2 M survivors on, one level.
      if (!want_part && !two_level && jit_try && np <= VH_STAGE_PARTS) {
        int words = 1, halves = 1;           // tuple words this plan would need: 64-bit states own one, 32-bit ones pair up (word 0 has one half free)
        for (int j = 0; j < P.nmetric; ++j) { if (vh_sop_bytes(P.m[j].sop()) == 8) ++words; else if (halves) --halves; else { ++words; halves = 1; } }
        if (words == 2 && shard_rows * sel >= 2e6 && sel >= 0.015) want_part = true;      // (at 1 % of 1 B rows the atomics still hide behind the scan: 1.18 ms direct, 1.31 partitioned)
      }
    }
    if (want_part) {
      // most rows pass: build the tuples without compacting survivors first (lanes kernel, phase 1 only)
      if (fast && !(p->flags & VH_PLAN_NO_LANES) && P.ngroup <= VH_LANES_COLS && P.nmetric <= VH_LANES_COLS && rows_to_scan) {
        bool ok = true;
        for (int i = 0; i < P.ngroup; ++i) ok &= vh_elem_size(P.g[i].type()) >= 4;
        for (int j = 0; j < P.nmetric; ++j) ok &= P.m[j].slot() != VH_SLOT_ROWID && vh_elem_size(P.m[j].type()) >= 4;
        if (ok) {
          if (p->flags & VH_PLAN_FORCE_LANES) lanes = true;
          else {
            double s2 = sel;
            if (s2 == 0) { rc = probed_selectivity(&s2); if (rc) { return rc; } }
            // (not when the scan is compiled for the plan and its tuples leave as whole lines — one level, <= 16 partitions: that kernel
            // beats the no-compaction form even when every row passes, 13.5 vs 14.1 ms per 1 B rows, profiles/r03/NOTES.md; with two levels
            // the 64-way phase 1 writes its tuples piecewise and the no-compaction form keeps its lead from 50 % on: 21.4 vs 22.8 ms)
            lanes = s2 >= 0.5 && !(jit_try && !two_level && np <= VH_STAGE_PARTS);
          }
        }
      }
      mode = VH_MODE_DENSE_PART;
      P.nlevel = two_level ? 2 : 1;
      P.agg_shift = shift;
      P.nfine = (int32_t)np;
      P.part_shift = two_level ? shift + 6 : shift;
      P.npart = (int32_t)(two_level ? (np + 63) / 64 : np);    // every partition goes through tuples (a split with direct atomics for the rest lost to both pure forms)
      // tuple words: word 0 = gid | first 32-bit value << 32; 64-bit values own a word; 32-bit values pair up
      int tw = 1, half_free_word = 0;  // word 0 has its upper half free
      bool have_half = true;
      for (int j = 0; j < P.nmetric; ++j) {
        if (vh_sop_bytes(P.m[j].sop()) == 8) { P.m[j].set_tword((uint8_t)tw++); P.m[j].set_tshift(0); }
        else if (have_half) { P.m[j].set_tword((uint8_t)half_free_word); P.m[j].set_tshift(32); have_half = false; }
        else { P.m[j].set_tword((uint8_t)tw); P.m[j].set_tshift(0); half_free_word = tw++; have_half = true; }
      }
      P.tw = tw;
      // ONE-word tuples when gid and every metric value fit 63 bits together — what the values need is known from the columns' recorded
      // min / max (refresh_stats keeps them for metric columns too): C3's (gid 17 bits, SUM value 10, COUNT 2) is 8 bytes instead of 16,
      // half the tuple bytes written by phase 1 and read back by phase 2. Only the compiled scan with the whole-line writer packs them.
      P.gid_bits = 0;
      if (jit_try && !lanes && !two_level && np <= VH_STAGE_PARTS_MAX && !knobs().no_stage && !(p->flags & VH_PLAN_NO_NARROW_TUPLES) && !getenv("VH_NO_NARROW_TUPLES")) {
        auto bits_of = [](uint64_t v) { int b = 1; while (b < 64 && (v >> b)) ++b; return b; };
        const int gb = bits_of(G - 1);
        int used = gb, mb[VH_MAX_METRIC] = {};
        bool fits = true;
        for (int j = 0; j < P.nmetric && fits; ++j) {
          const int col = metric_col[j];
          if (col < 0) { fits = false; break; }
          const VhColumn& c = t->cols[col];
          if (c.elem == VH_F32 || c.elem == VH_F64) { fits = false; break; }
          uint64_t klo = ~0ull, khi = 0;
          for (uint32_t sgi : live) { const VhSegStat& st = t->stats[col][sgi]; if (st.lo > st.hi) continue; klo = std::min(klo, st.lo); khi = std::max(khi, st.hi); }
          if (klo > khi) klo = khi = order_key_of_bits(c.elem, 0);
          const bool sgn = c.elem == VH_I8 || c.elem == VH_I16 || c.elem == VH_I32 || c.elem == VH_I64;
          if (sgn && (int64_t)(klo ^ (1ull << 63)) < 0) { fits = false; break; }      // negative values: the tuple's fields are unsigned
          mb[j] = bits_of(sgn ? (khi ^ (1ull << 63)) : bits_of_order_key(c.elem, khi));
          used += mb[j];
        }
        if (fits && used <= 63) {
          P.gid_bits = gb; P.tw = 1;
          int at = gb;
          for (int j = 0; j < P.nmetric; ++j) { P.m[j].set_tword(0); P.m[j].set_tshift((uint8_t)at); P.m[j].tbits = (uint32_t)mb[j]; at += mb[j]; }
        }
      }
      // the drain specialised for "two unsigned 32-bit group columns, SUM(64-bit) + SUM(32-bit)" (vh_consume_fast, SHAPE 1)
      P.shape = 0;
      if (!jit_try && !lanes && !(p->flags & VH_PLAN_NO_SHAPE) && (P.ngroup == 1 || P.ngroup == 2) && P.nmetric == 2 && tw == 2 && G <= 0xFFFFFFFFull) {   // (a per-query compiled kernel knows the whole plan, not two shapes of it)
        bool ok = true;
        for (int i = 0; i < P.ngroup; ++i)
          ok &= (P.g[i].type() == VH_U32 || P.g[i].type() == VH_U16 || P.g[i].type() == VH_U8) && P.g[i].gran() == VH_T_NONE && P.g[i].nroll() == 0 && P.g[i].lo <= 0xFFFFFFFFull &&
                P.g[i].extent <= 0xFFFFFFFFull && P.g[i].stride <= 0xFFFFFFFFull;
        auto is64 = [&](int j) { return P.m[j].sop() == SOP_ADD64 && P.m[j].slot() != VH_SLOT_ROWID && vh_elem_size(P.m[j].type()) == 8 && P.m[j].tword() == 1; };
        auto is32 = [&](int j) { return P.m[j].sop() == SOP_ADD32 && P.m[j].slot() != VH_SLOT_ROWID && vh_elem_size(P.m[j].type()) == 4 && P.m[j].tword() == 0 && P.m[j].tshift() == 32; };
        const int shape = is64(0) && is32(1) ? 1 : is32(0) && is64(1) ? 2 : 0;
        if (ok && shape) {
          P.shape = shape;
          for (int i = 0; i < P.ngroup; ++i) P.g[i].set_key_shift(32u - 8u * (uint32_t)vh_elem_size(P.g[i].type()));
          if (P.ngroup == 1) {        // the drain always folds two digits: the second one re-reads the first column and counts for nothing
            P.g[1] = P.g[0];
            P.g[1].lo = 0; P.g[1].extent = 0xFFFFFFFFull; P.g[1].stride = 0; P.g[1].set_key_shift(31);    // (one bit of it: never out of range)
          }
        }
      }
      if (part_carrier >= 0) { P.m[part_carrier].set_sop(SOP_ADD32P); state_bytes_per_group += 4; }   // (its tuple slot stays 32 bits wide)
      // phase-2 LDS table for one partition
      const uint64_t gpp = 1ull << shift;
      size_t off = 0;
      for (int pass = 0; pass < 2; ++pass)
        for (int j = 0; j < P.nmetric; ++j) {
          const int b = vh_sop_bytes(P.m[j].sop());
          if ((pass == 0) != (b == 8)) continue;
          P.m[j].lds_off = (uint32_t)off; off += gpp * b;
        }
      off = (off + 7) / 8 * 8;
      P.lds_present_off = (uint32_t)off; if (part_carrier < 0) off += gpp;
      lds_table = (off + 15) / 16 * 16;
      P.lds_bytes = (uint32_t)lds_table;
      part_tuple_cap = part_tuples_override ? part_tuples_override
                     : std::max<uint64_t>((uint64_t)((double)rows_to_scan * std::max(sel, 0.02) * 1.25), 1ull << 16);
      part_tuple_cap = std::min<uint64_t>(part_tuple_cap, rows_to_scan + 1);
    }
  }
  // direct global atomics: fold the presence flag into a 32-bit SUM state (SOP_ADD32P) when there is one
  P.present_carrier = -1;
  if (mode == VH_MODE_DENSE_PART)
    for (int j = 0; j < P.nmetric; ++j) if (P.m[j].sop() == SOP_ADD32P) P.present_carrier = j;
  if (mode == VH_MODE_DENSE_GLOBAL && !(p->flags & VH_PLAN_NO_CARRIER)) {
    for (int j = 0; j < P.nmetric; ++j)
      if (P.m[j].sop() == SOP_ADD32) { P.m[j].set_sop(SOP_ADD32P); P.present_carrier = j; state_bytes_per_group += 4; break; }
  }
  r->mode = mode;
  r->info.path = mode == VH_MODE_DENSE_LDS ? (p->ngroups ? VH_PATH_DENSE_LDS : VH_PATH_SCALAR)
               : mode == VH_MODE_DENSE_GLOBAL ? VH_PATH_DENSE_GLOBAL : mode == VH_MODE_DENSE_PART ? VH_PATH_DENSE_PART : VH_PATH_HASH;

  // per-XCD private copies only while they stay cache-sized
  if (mode != VH_MODE_HASH && mode != VH_MODE_DENSE_PART && P.nbitset == 0 && !(p->flags & VH_PLAN_NO_XCD_PRIVATE) &&
      G <= 16384)   // private copies pay off only against same-address contention (C2 forced to HBM: 3.4 vs 10 ms);
    nxcd = g_ctx.num_xcd;   // with >= 100 K groups one table is as fast and needs no merge pass
  // DENSE_PART: the blocks that share one LDS-sized range each write a private copy of it with plain stores (block b -> copy b;
  // every group of the range, present or not) and dense_merge_kernel adds the copies up — C3: 16 blocks x 13 ranges used to
  // flush 3.2 M global atomics (0.14 ms of phase 2's 0.35) into one table
  if (mode == VH_MODE_DENSE_PART) {
    part_bpp = std::max(1, std::min(32, g_ctx.num_cu / std::max(1, P.nfine)));    // one 1024-thread block per CU: phase 2 lives off LDS atomics, so every CU counts
    if (!(p->flags & VH_PLAN_NO_XCD_PRIVATE)) nxcd = part_bpp;
  }
  P.nxcd = nxcd; r->nxcd = nxcd;
  P.xcd_stride = (G + 63) / 64 * 64;

  if (mode == VH_MODE_HASH) {
    // sizing: explicit override (regrow) > caller's hint > what the same group columns produced last time > 1 M
    std::string sig;
    for (int i = 0; i < p->ngroups; ++i) sig += std::to_string(p->groups[i].col) + ":" + std::to_string(P.g[i].gran()) + ":" + std::to_string(P.g[i].nroll()) + ",";
    r->group_sig = sig;
    const auto seen = t->groups_seen.find(sig);
    const uint64_t hint = p->groups_hint ? p->groups_hint : (seen != t->groups_seen.end() ? seen->second + seen->second / 4 : 0);
    uint64_t want = hash_capacity_override ? hash_capacity_override
                  : std::max<uint64_t>(hint ? hint * 2 : (1ull << 20), 1ull << 12);
    const uint64_t cap_rows = std::max<uint64_t>(rows_to_scan * 2, 1ull << 12);
    if (!hash_capacity_override) want = std::min(want, cap_rows);
    capacity = 1; while (capacity < want) capacity <<= 1;
    P.hmask = capacity - 1;
    P.max_probe = (uint32_t)std::min<uint64_t>(capacity - 1, 2048);
    // LDS front table (north_star's "LDS-bucketed open-address tables"): single-word keys, no count-distinct (its
    // sets are keyed by the HBM slot). Skipped when the same group columns are known to produce far more groups
    // than it holds; otherwise every wave decides for itself after a warm-up (VhLdsHashWave).
    if (P.key_words == 1 && P.nbitset == 0 && P.nmetric >= 1 && !(p->flags & VH_PLAN_NO_LDS_HASH)) {
      size_t sb = 8;
      for (int j = 0; j < P.nmetric; ++j) sb += vh_sop_bytes(P.m[j].sop());
      uint32_t slots = 2048;
      while (slots > 256 && (size_t)slots * sb > 24 * 1024) slots >>= 1;
      const uint64_t known = p->groups_hint ? p->groups_hint : (seen != t->groups_seen.end() ? seen->second : 0);
      if ((size_t)slots * sb <= 24 * 1024 && known <= (uint64_t)slots * 4) {
        size_t off = 0;
        P.lds_hkeys_off = 0; off += (size_t)slots * 8;
        for (int pass = 0; pass < 2; ++pass)
          for (int j = 0; j < P.nmetric; ++j) {
            const int b = vh_sop_bytes(P.m[j].sop());
            if ((pass == 0) != (b == 8)) continue;
            P.m[j].lds_off = (uint32_t)off; off += (size_t)slots * b;
          }
        P.lds_hash_slots = slots;
        lds_table = (off + 15) / 16 * 16;
        P.lds_bytes = (uint32_t)lds_table;
      }
    }
    // the no-compaction kernel over the LDS front table (time-bucket GROUP BYs over most of the data)
    if (P.lds_hash_slots && fast && !(p->flags & VH_PLAN_NO_LANES) && P.ngroup >= 1 && P.ngroup <= VH_LANES_COLS &&
        P.nmetric <= VH_LANES_COLS && rows_to_scan) {
      bool ok = true;
      for (int i = 0; i < P.ngroup; ++i) ok &= vh_elem_size(P.g[i].type()) >= 4;
      for (int j = 0; j < P.nmetric; ++j) ok &= P.m[j].slot() != VH_SLOT_ROWID && vh_elem_size(P.m[j].type()) >= 4;
      if (ok) {
        if (p->flags & VH_PLAN_FORCE_LANES) lanes = true;
        else {
          double sel = 0;
          rc = probed_selectivity(&sel);
          if (rc) { return rc; }
          // per-row work here is heavy (calendar arithmetic, LDS probe) and runs once per ROW SLOT, passing or not:
          // measured on 100 M rows into day buckets, 50 % pass: 1.05 ms compacted vs 1.38 ms lanes; 100 %: 2.29 vs 1.92
          lanes = sel >= 0.7;
        }
      }
    }
  }
  return VH_OK;
}

int QueryBuild::plan_hashed_partitioning() {
  int rc = VH_OK; (void)rc;
  // ---------------- hashed partitioning (HASH organisation with MANY groups: hash_part_agg_kernel, vh_kernels.h)
  // With tens of millions of groups every survivor costs the plain hash table 2-5 read-modify-writes at random addresses of a
  // table no cache holds — the device does ~20 G of those per second (C5: 312 M per 125 M rows = 15.9 ms) — and a count-distinct
  // makes it three more per row. Survivors are instead written out as 16-byte tuples keyed by a bijective mix of the packed group key,
  // radix-partitioned by its top bits (64 ways in the scan kernel, 64 more in part_split_tile_kernel) and aggregated range by range
  // in LDS: sequential traffic of 16 B per tuple and level instead of a 128-byte line read and written per update.
  // (Sharded queries take it too: what ranks exchange — finalised groups and, for a bitset metric, (group, id) pairs by owner — does not
  // depend on how a rank aggregated its shard, so the choice need not even agree between ranks; with an agreement it is made from the
  // agreed figures all the same.)
  if (mode == VH_MODE_HASH && jit_try && (!lanes || (p->flags & VH_PLAN_FORCE_HPART)) && !no_hpart && !(p->flags & VH_PLAN_NO_HPART) && P.key_words == 1 && P.nmetric >= 1 && rows_to_scan) {
    int bits = 0, nb = 0;
    bool ok = true;
    for (int j = 0; j < P.nmetric; ++j) {
      if (P.m[j].sop() == SOP_BITSET) { ++nb; ok &= !P.bs_wide[P.m[j].slot()]; }
      else bits += 8 * vh_sop_bytes(P.m[j].sop());
    }
    ok &= bits <= 64 && nb == P.nbitset && nb <= 1;
    if (ok) {
      double sel = 1.0;
      rc = probed_selectivity(&sel);
      if (rc) { return rc; }
      const double survivors = (double)rows_to_scan * sel;
      const auto seen = t->groups_seen.find(r->group_sig);
      const uint64_t known = p->groups_hint ? p->groups_hint : (seen != t->groups_seen.end() ? seen->second : 0);
      const double survivors_dec = ag ? (double)ag->rows_max * sel : survivors;      // (what the decision looks at: the largest shard's)
      // worth it when the groups are many (the LDS front table then only wastes probes) and the tuples pay for three more launches:
      // C5 (count-distinct) 15.9 ms through the plain table against 5.6 ms, C5t (groups + COUNT alone) 4.0 against 2.8 ms
      // (profiles/r03/NOTES.md; the first version of the tuple path lost that one, 5-6 ms)
      hpart = (p->flags & VH_PLAN_FORCE_HPART) || (survivors_dec >= 8e6 && known >= 2000000);
      if (hpart) {
        hp_tuple_cap = part_tuples_override ? part_tuples_override : std::max<uint64_t>((uint64_t)(survivors * 1.25) + 1024, 1ull << 16);
        hp_tuple_cap = std::min<uint64_t>(hp_tuple_cap, rows_to_scan + 1);
        if (nb) {    // the tuples carry the row's ids, two at a time: a row of k ids writes max(1, ceil(k / 2)) tuples — at most rows + (ids + rows) / 2 of them
          const uint64_t worst = (bitset_ids[0] + std::min<uint64_t>(hp_tuple_cap, rows_to_scan)) / 2 + 1;
          const uint64_t by_ids = std::min<uint64_t>(worst, (uint64_t)(((double)bitset_ids[0] * std::max(sel, 0.02) * 1.25 + (double)hp_tuple_cap) / 2) + (1ull << 16));
          hp_tuple_cap = part_tuples_override ? hp_tuple_cap + worst : std::max(hp_tuple_cap, by_ids) + (1ull << 16);
          hp_units = 2;
          // Packed tuples: 16 bytes instead of 32 when the payload values and two ids fit ONE word next to their count. The bits come from
          // what the mirror knows about the scanned segments: min / max of the metric columns (refresh_stats), the largest id (bs_maxid).
          auto bits_of = [](uint64_t v) { int b = 1; while (b < 64 && (v >> b)) ++b; return b; };
          bool fits = !(p->flags & VH_PLAN_NO_HP_PACK) && !getenv("VH_NO_HP_PACK");
          int pbits = 0, mb[VH_MAX_METRIC] = {};
          for (int j = 0; j < P.nmetric && fits; ++j) {
            if (P.m[j].sop() == SOP_BITSET) continue;
            const int col = metric_col[j];
            if (col < 0) { fits = false; break; }                                       // (the virtual row id)
            const VhColumn& c = t->cols[col];
            if (c.elem == VH_F32 || c.elem == VH_F64) { fits = false; break; }
            uint64_t klo = ~0ull, khi = 0;
            for (uint32_t sgi : live) { const VhSegStat& st = t->stats[col][sgi]; if (st.lo > st.hi) continue; klo = std::min(klo, st.lo); khi = std::max(khi, st.hi); }
            if (klo > khi) { klo = khi = order_key_of_bits(c.elem, 0); }
            const bool sgn = c.elem == VH_I8 || c.elem == VH_I16 || c.elem == VH_I32 || c.elem == VH_I64;
            const uint64_t vlo = bits_of_order_key(c.elem, klo), vhi = bits_of_order_key(c.elem, khi);
            if (sgn && ((int64_t)(klo ^ (1ull << 63)) < 0)) { fits = false; break; }     // negative values: the tuple's fields are unsigned
            (void)vlo;
            mb[j] = bits_of(sgn ? (khi ^ (1ull << 63)) : vhi);
            pbits += mb[j];
          }
          uint64_t maxid = 0;
          for (uint32_t sgi : live) maxid = std::max(maxid, t->cols[bitset_col[0]].bs_maxid[sgi]);
          int idbits = bits_of(maxid);
          if (const char* e = getenv("VH_TEST_HP_IDBITS")) idbits = std::max(1, atoi(e));      // tests: ids that do NOT fit -> VH_ERR_HP_WIDE -> the plain hash table
          if (fits && idbits <= 32 && pbits + 2 * idbits <= 61) {
            hp_pack = true; hp_units = 1; hp_pbits = pbits; hp_idbits = idbits;
            for (int j = 0; j < P.nmetric; ++j) P.m[j].tbits = (uint32_t)mb[j];
          }
        }
        lanes = false;
        P.hpart = 1; P.gid_shift = 32;
        P.npart = 1; P.part_shift = 0; P.nlevel = 1; P.agg_shift = 0; P.nfine = 1;      // (the scan kernel writes ONE stream per kind; vh_hpart.h partitions it)
        P.tw = 2 * hp_units;
        // payload word: the 64-bit state alone, or up to two 32-bit ones
        int used = 0;
        for (int j = 0; j < P.nmetric; ++j) {
          if (P.m[j].sop() == SOP_BITSET) continue;
          P.m[j].set_tword(1); P.m[j].set_tshift((uint8_t)used);
          used += hp_pack ? (int)P.m[j].tbits : 8 * vh_sop_bytes(P.m[j].sop());
        }
        // LDS tables of hp_aggregate_kernel, one of 65 536 ranges at a time: group slots for the range's expected groups at <= 70 % load,
        // (group slot, id) slots likewise; what does not fit even 4096 / 16384 slots is worked through in passes
        const double groups_est = (known ? (double)known * 1.1 : survivors * 1.1) / 65536.0;
        const double ids_est = nb ? (double)bitset_ids[0] * std::max(sel, 0.02) * 1.1 / 65536.0 : 0.0;
        size_t slot_bytes = 8;                                  // a group slot: the mixed key + every state
        for (int j = 0; j < P.nmetric; ++j) slot_bytes += P.m[j].sop() == SOP_BITSET ? 4 : vh_sop_bytes(P.m[j].sop());      // (a cardinality, in LDS: 32 bits)
        auto table_bytes = [&](uint32_t g, uint32_t q) { return (size_t)(g + 1) * slot_bytes + (size_t)q * 8; };
        const size_t budget = 136 * 1024;                       // of the 160 KB a block may own (lists, counters and alignment take the rest)
        uint32_t passes = hp_passes_override ? hp_passes_override : 1, gs = 256, ss = nb ? 1024 : 0;
        if (const char* env_passes = getenv("VH_TEST_HPART_PASSES")) if (!hp_passes_override) passes = (uint32_t)std::max(1, atoi(env_passes));
        for (;;) {       // tables for one pass's share of a range at <= 70 % load; what the LDS cannot hold takes more passes
          const double load_g = knobs().hp_load_g, load_s = knobs().hp_load_s;
          const bool need_g = hp_passes_override ? true : groups_est / passes > load_g * gs, need_s = nb && (hp_passes_override ? true : ids_est / passes > load_s * ss);
          if (need_g && table_bytes(gs * 2, ss) <= budget && (!need_s || gs * 4 <= ss * 2 || table_bytes(gs, ss * 2) > budget)) { gs *= 2; continue; }
          if (need_s && table_bytes(gs, ss * 2) <= budget) { ss *= 2; continue; }
          if (need_g && table_bytes(gs * 2, ss) <= budget) { gs *= 2; continue; }
          if (hp_passes_override || (!need_g && !need_s) || passes >= 64) break;      // (a re-plan takes the biggest tables that fit, whatever the estimate said)
          passes *= 2;
        }
        P.hp_passes = (int32_t)passes; P.hp_gslots = (int32_t)gs; P.hp_sslots = (int32_t)ss;
        size_t off = 0;
        P.hp_keys_off = 0; off += (size_t)(P.hp_gslots + 1) * 8;
        for (int pass = 0; pass < 2; ++pass)
          for (int j = 0; j < P.nmetric; ++j) {
            const int b = P.m[j].sop() == SOP_BITSET ? 4 : vh_sop_bytes(P.m[j].sop());
            if ((pass == 0) != (b == 8)) continue;
            P.m[j].lds_off = (uint32_t)off; off += (size_t)(P.hp_gslots + 1) * b;
          }
        off = (off + 7) / 8 * 8;
        P.hp_set_off = (uint32_t)off; off += (size_t)P.hp_sslots * 8;
        lds_table = (off + 15) / 16 * 16;          // (of the aggregation kernel; the scan kernel keeps no table)
        P.lds_hash_slots = 0; P.lds_bytes = 0;
        // the list of group records: never more groups than tuples; blocks take it in chunks and leave a tail of their last one unused
        hp_bpp = vh_hpart_bpp(g_ctx.num_cu, lds_table);
        if (knobs().hp_bpp > 0 && HP_FAN % knobs().hp_bpp == 0) hp_bpp = knobs().hp_bpp;
        hp_chunk = 16384;
        while (hp_chunk > 256 && (uint64_t)hp_chunk * HP_FAN * hp_bpp * 4 > hp_tuple_cap) hp_chunk /= 2;
        capacity = hp_tuple_cap + (uint64_t)HP_FAN * hp_bpp * hp_chunk * 2;
        P.hmask = capacity - 1;
        P.present_carrier = -1;
      }
    }
  }
  if (P.nbitset && !hpart) jit_try = false;
  // the plain hash table is bound by random read-modify-writes, not by the scan: the pre-built kernel (smaller blocks, more of them per CU
  // next to the LDS front table) runs it a tenth faster than the compiled one (C5t: 3.99 vs 4.39 ms) — the compiled kernel is for plans it cannot hold
  if (mode == VH_MODE_HASH && !hpart && fast && !(p->flags & VH_PLAN_FORCE_JIT) && vh_jit_policy() != VH_JIT_FORCE) jit_try = false;
  return VH_OK;
}

int QueryBuild::choose_projection() {
  int rc = VH_OK; (void)rc;
  // ---------------- payload projection: when few rows pass, a survivor's group / metric values come out of ONE packed
  // record (vh_table_pack) instead of one line per column arena. Only the compacting kernels gather by row; the lanes
  // kernels read whole column ranges and keep the arenas.
  {
    std::vector<int32_t> gcols;
    for (int i = 0; i < p->ngroups; ++i) gcols.push_back(p->groups[i].col);
    for (int j = 0; j < P.nmetric; ++j) if (metric_col[j] >= 0) gcols.push_back(metric_col[j]);
    std::sort(gcols.begin(), gcols.end());
    gcols.erase(std::unique(gcols.begin(), gcols.end()), gcols.end());
    bool want = !lanes && !(p->flags & VH_PLAN_NO_PACK) && !gcols.empty() && gcols.size() <= VH_PACK_MAX_COLS && rows_to_scan &&
                P.nslots + (int)gcols.size() <= VH_MAX_SLOTS;
    const bool forced = (p->flags & VH_PLAN_FORCE_PACK) != 0;
    if (want && !forced) {
      // lines touched per survivor: one record vs one per column; the projection stops paying off once most lines of
      // the arenas are touched anyway (C3 columns: ~20 % of the rows passing)
      want = fastj && p->nfilter > 0;
      if (want) {
        double sel = 1.0;
        rc = probed_selectivity(&sel);
        if (rc) { return rc; }
        want = sel <= 0.15;
      }
    }
    VhPack* use = nullptr;
    if (want) {
      // (compressed records are read by the per-query compiled kernels only)
      for (auto& pk : t->packs) {
        bool all = !pk->compressed || jit_try;
        for (int c : gcols) all &= pk->col_index(c) >= 0;
        if (all && (!use || pk->rec_bytes < use->rec_bytes || (pk->rec_bytes == use->rec_bytes && pk->compressed && !use->compressed))) use = pk.get();
      }
      const int auto_after = g_preparing ? 1 : knobs().auto_pack;   // 0: never build one unasked
      if (!use && (forced || auto_after > 0)) {
        std::string sig = jit_try ? "c:" : "p:";
        for (int c : gcols) sig += std::to_string(c) + ",";
        bool build = forced || ++t->gather_seen[sig] >= (uint32_t)auto_after;
        if (build && !forced) {        // room: the projection must leave a quarter of the device free and not outgrow the table
          uint32_t bytes = 0; for (int c : gcols) bytes += (uint32_t)t->cols[c].esize;
          uint32_t rec = 8; while (rec < bytes) rec <<= 1;
          const size_t need = (size_t)t->cap_seg * ((t->segment_rows + 255) / 256 * 256) * rec;
          size_t free_b = 0, total_b = 0;
          build = bytes <= 64 && hipMemGetInfo(&free_b, &total_b) == hipSuccess && need <= t->device_bytes && free_b > need + total_b / 4;
          if (!build) t->gather_seen[sig] = 0;
        }
        if (build && table_pack_locked(t, gcols.data(), (int32_t)gcols.size(), !forced, &use, jit_try && !knobs().pack_plain) != VH_OK) use = nullptr;
      }
    }
    if (use) {
      rc = pack_refresh(t, use, 0, nseg);   // on the table's main stream, complete when it returns
      if (rc == VH_PACK_STALE) {            // a synced value outgrew its stored width: rebuilt at the widths the values need now
        const bool was_auto = use->automatic;
        pack_drop(t, use);
        use = nullptr;
        if (table_pack_locked(t, gcols.data(), (int32_t)gcols.size(), was_auto, &use, true) != VH_OK) use = nullptr;
        rc = VH_OK;
      }
      if (rc) { return rc; }
    }
    if (use) {
      int pslot_of[256];
      for (int i = 0; i < 256; ++i) pslot_of[i] = -1;
      auto pslot = [&](int col) {
        if (pslot_of[col] >= 0) return pslot_of[col];
        const int k = use->col_index(col);
        P.colbase[P.nslots] = use->base + use->off[k];
        P.colstride[P.nslots] = use->stride;
        P.colpitch[P.nslots] = use->rec_bytes;
        slot_rec[P.nslots] = 0; slot_recoff[P.nslots] = (int)use->off[k]; slot_stored[P.nslots] = (int)use->width[k];
        return pslot_of[col] = P.nslots++;
      };
      for (int i = 0; i < p->ngroups; ++i) P.g[i].set_slot((uint16_t)pslot(p->groups[i].col));
      for (int j = 0; j < P.nmetric; ++j) if (metric_col[j] >= 0) P.m[j].set_slot((uint16_t)pslot(metric_col[j]));
      packed = true;
      packed_compressed = use->compressed;
    }
  }
  return VH_OK;
}

int QueryBuild::compile_kernel() {
  int rc = VH_OK; (void)rc;
  // ---------------- the scan kernel compiled for this plan shape (vh_jit.hip), when there is to be one
  if (jit_try && lanes) jit_try = false;          // the no-compaction kernels are pre-built only
  if (jit_try) {
    VhJitShape& js = jshape;
    js.mode = mode;
    const size_t qw = (size_t)VJ_QUEUE_CAP * sizeof(uint32_t);       // per wave
    if (mode == VH_MODE_DENSE_LDS || (mode == VH_MODE_HASH && !hpart)) {          // an LDS table per block: the widest block whose table + queues stay within the 64 KB a module kernel may ask for
      jit_block = mode == VH_MODE_DENSE_LDS ? 1024 : 512;
      while (jit_block > 256 && lds_table + (size_t)(jit_block / 64) * qw > 64 * 1024) jit_block /= 2;
      if (lds_table + (size_t)(jit_block / 64) * qw > 64 * 1024) jit_try = false;
      if (mode == VH_MODE_HASH && !P.lds_hash_slots) jit_block = 256;
    }
    js.block = jit_block;
    js.ablate = knobs().jit_ablate;      // measurement only (profiles/r03/NOTES.md): 1 = no gathers, 2 = nothing behind the gathers
    js.xcd = nxcd > 1 ? 1 : 0;
    js.scope = (mode == VH_MODE_DENSE_GLOBAL || mode == VH_MODE_DENSE_LDS) && nxcd > 1 ? (int)__HIP_MEMORY_SCOPE_WORKGROUP : (int)__HIP_MEMORY_SCOPE_AGENT;
    js.carrier = P.present_carrier;
    js.tw = mode == VH_MODE_DENSE_PART ? P.tw : 1;
    js.key_words = mode == VH_MODE_HASH ? P.key_words : 1;
    js.lds_hash = P.lds_hash_slots ? 1 : 0;
    js.gid32 = mode != VH_MODE_HASH && G <= 0xFFFFFFFFull;
    const bool env_no_stage = knobs().no_stage;             // measurement: tuples appended piece by piece (vh_part_direct_add)
    js.gid_bits = mode == VH_MODE_DENSE_PART ? P.gid_bits : 0;
    js.stage = mode == VH_MODE_DENSE_PART && (P.tw == 2 || P.gid_bits) && !env_no_stage ? (P.npart <= VH_STAGE_PARTS ? VH_STAGE_PARTS : P.npart <= VH_STAGE_PARTS_MAX ? VH_STAGE_PARTS_MAX : 0) : 0;
    js.hpart = hpart ? 1 : 0;
    js.hp_pack = hp_pack ? 1 : 0; js.hp_pbits = hp_pbits; js.hp_idbits = hp_idbits;
    js.ng = P.ngroup; js.nm = P.nmetric;
    for (int i = 0; i < P.ngroup; ++i) {
      const VhGroupDev& g = P.g[i];
      VhJitCol& c = js.g[i];
      c.slot = (int)g.slot(); c.type = (int)g.type(); c.pitch = (int)P.colpitch[g.slot()];
      c.rec = slot_rec[g.slot()]; c.off = slot_recoff[g.slot()]; c.stored = slot_stored[g.slot()];
      c.sext = mode != VH_MODE_HASH;
      c.gran = (int)g.gran(); c.nroll = (int)g.nroll(); c.micro = (int)g.micro();
      c.key_word = (int)g.key_word(); c.key_shift = (int)g.key_shift();
      for (int k = 0; k < c.nroll; ++k) c.roll_unit[k] = (int)g.roll_unit(k);
      if (vh_elem_size(c.type) > 4 || c.type == VH_F32) js.gid32 = 0;
    }
    for (int j = 0; j < P.nmetric; ++j) {
      const VhMetricDev& m = P.m[j];
      VhJitCol& c = js.m[j];
      c.rowid = m.slot() == VH_SLOT_ROWID;
      c.bitset = m.sop() == SOP_BITSET;
      if (c.bitset) js.bitset_j = j;
      c.type = (int)m.type(); c.sop = (int)m.sop(); c.tword = (int)m.tword(); c.tshift = (int)m.tshift(); c.tbits = hp_pack || (mode == VH_MODE_DENSE_PART && P.gid_bits) ? (int)m.tbits : 0;
      c.sext = vh_sop_sext((int)m.sop());
      if (!c.rowid && !c.bitset) { c.slot = (int)m.slot(); c.pitch = (int)P.colpitch[m.slot()]; c.rec = slot_rec[m.slot()]; c.off = slot_recoff[m.slot()]; c.stored = slot_stored[m.slot()]; }
    }
    if (jit_try) {
      std::string jerr;
      jk = vh_jit_get(js, &jerr);
      if (!jk) {
        // no kernel for this shape (hipRTC missing, or the text did not compile): plan again for the pre-built kernels. The
        // failure is remembered per shape, so only the first query of the shape pays for the attempt.
        if (knobs().jit_verbose) fprintf(stderr, "vh: per-query kernel unavailable, falling back: %s\n", jerr.c_str());
        if ((p->flags & VH_PLAN_FORCE_JIT) || vh_jit_policy() == VH_JIT_FORCE) return vh_fail(VH_E_UNSUPPORTED, "per-query kernel requested (VH_PLAN_FORCE_JIT / VH_JIT=force) but unavailable: %s", jerr.c_str());
        vh_plan p2 = *p;
        p2.flags |= VH_PLAN_NO_JIT;
        // (the organisation must not depend on whether THIS rank could compile: partitioning chosen because a compiled scan makes tuples
        // cheap stays chosen — the pre-built kernels run it too — so that sharded ranks keep identically laid out partial tables)
        if (mode == VH_MODE_DENSE_PART) p2.flags |= VH_PLAN_FORCE_PART;
        holder.reset();
        done = true;
        return query_launch_locked(t, x, &p2, out, hash_capacity_override, force_hash, part_tuples_override, no_part, plan_only, summary_out, ag, device_rows, hp_passes_override, no_hpart);
      }
    }
  }
  if (!jk && (packed_compressed || (mode == VH_MODE_DENSE_PART && P.gid_bits))) {      // compressed records / one-word tuples and no compiled kernel to handle them after all: plan again for the pre-built ones
    vh_plan p2 = *p;
    p2.flags |= VH_PLAN_NO_JIT;
    holder.reset();
    done = true;
    return query_launch_locked(t, x, &p2, out, hash_capacity_override, force_hash, part_tuples_override, no_part, plan_only, summary_out, ag, device_rows, hp_passes_override, no_hpart);
  }
  return VH_OK;
}

void QueryBuild::scan_dispatch(int grid_, int* occ) {
  const size_t qb = (size_t)(BLOCK / 64) * VhScanCfg<256>::kQueueCap * sizeof(uint32_t);
  const size_t lds_ = ((mode == VH_MODE_DENSE_LDS || mode == VH_MODE_HASH) ? lds_table : 0) + qb;
  hipStream_t s_ = x->stream();
  if (jk) {
    const size_t jl = ((mode == VH_MODE_DENSE_LDS || (mode == VH_MODE_HASH && !hpart)) ? lds_table : 0) + (size_t)(BLOCK / 64) * (VJ_QUEUE_CAP * sizeof(uint32_t) + (size_t)VH_STAGE_BYTES(jshape.stage));
    if (occ) *occ = vh_jit_occupancy(jk, BLOCK, jl);
    else (void)vh_jit_launch(jk, P, grid_, BLOCK, jl, s_);
  }
  else if (mode == VH_MODE_DENSE_PART) {
    if (lanes) vh_launch_scan_lanes_part(P, grid_, 4 * vh_part_tile_bytes(P), s_, occ);
    else vh_launch_scan_fast_part(P, grid_, qb, s_, occ);   // the compacting form appends straight to the extents: no tile in LDS
  }
  else if (!fast) { if (!occ) vh_launch_scan_generic(mode, P, grid_, lds_, nxcd > 1, s_); }
  else if (lanes && mode == VH_MODE_HASH) vh_launch_scan_lanes_hash(P, grid_, lds_, s_, occ);
  else if (lanes) vh_launch_scan_lanes_lds(P, BLOCK, grid_, lds_, nxcd > 1, s_, occ);
  else if (mode == VH_MODE_DENSE_LDS) { if (!occ) vh_launch_scan_fast_lds(P, grid_, lds_, nxcd > 1, s_); }   // 1024-thread blocks: one per CU
  else if (mode == VH_MODE_DENSE_GLOBAL) vh_launch_scan_fast_global(P, grid_, lds_, nxcd > 1, s_, occ);
  else vh_launch_scan_fast_hash(P, grid_, lds_, s_, occ);
}

int QueryBuild::decompose_work() {
  int rc = VH_OK; (void)rc;
  // ---------------- work decomposition
  // The lanes kernels expose the latency of their payload loads (issued and consumed inside a sub-step), so they gain
  // from every extra resident wave: 256-thread blocks, as many per CU as registers and LDS allow (asked of the runtime
  // per instantiation: 5 for one predicate column today, 4 for two or more) — C2 at 1 B rows: 3.67 -> 3.39 ms. The LDS
  // variant pays for more blocks with more table merges at the end (blocks x groups x metrics global atomics), so
  // it keeps one 1024-thread block per CU unless the scan dwarfs that.
  const int env_lanes_block = knobs().lanes_block, env_bpc = knobs().blocks_per_cu, env_unit = knobs().unit_rows;
  BLOCK = jk ? jit_block : mode == VH_MODE_DENSE_LDS ? 1024 : 256;
  if (mode == VH_MODE_DENSE_LDS && lanes) {
    if (env_lanes_block == 256 || env_lanes_block == 512 || env_lanes_block == 1024) BLOCK = env_lanes_block;
    else if ((uint64_t)g_ctx.num_cu * 5 * G * std::max(1, P.nmetric) <= rows_to_scan / 32) BLOCK = 256;
  }
  // One place decides which scan kernel runs; with `occ` it only asks how many of its blocks fit a CU.
  {   // the kernel symbol(s) this query runs, as rocprofv3 prints them (vh_result_kernel: bench.py's roofline.kernel)
    const int np_ = std::max(1, (int)P.npred), scope = (mode == VH_MODE_DENSE_GLOBAL || mode == VH_MODE_DENSE_LDS) && nxcd > 1 ? (int)__HIP_MEMORY_SCOPE_WORKGROUP : (int)__HIP_MEMORY_SCOPE_AGENT;
    char nm[160];
    if (!fast) snprintf(nm, sizeof(nm), "scan_agg_kernel<%d, %d, %d>", mode, BLOCK, scope);
    else if (mode == VH_MODE_DENSE_PART && !lanes && P.shape) snprintf(nm, sizeof(nm), "scan_agg_shape_kernel<%d, %d, %d, %d, %d>", mode, BLOCK, (int)__HIP_MEMORY_SCOPE_AGENT, np_, P.shape);
    else snprintf(nm, sizeof(nm), "%s<%d, %d, %d, %d>", lanes ? "scan_agg_lanes_kernel" : "scan_agg_fast_kernel", mode, BLOCK,
                  (mode == VH_MODE_HASH || mode == VH_MODE_DENSE_PART) ? (int)__HIP_MEMORY_SCOPE_AGENT : scope, np_);
    r->kernel = jk ? jk->name : std::string(nm);
    if (hpart) {      // (the scatter kernel runs twice per query, level A and level B: named twice, so that per-query sums over the names count it twice)
      char hn[160];
      snprintf(hn, sizeof(hn), " + hp_scatter_kernel<1024, %d> + hp_scatter_kernel<1024, %d> + ", hp_units, hp_units);
      r->kernel += hn + jk->name + "_hpagg";
    }
    if (mode == VH_MODE_DENSE_PART) r->kernel += P.nlevel != 2 ? " + part_agg_kernel<1024>" : (P.tw == 2 && !getenv("VH_NO_SPLIT_TILE")) ? " + part_split_tile_kernel<256> + part_agg_kernel<1024>" : " + part_split_kernel<256> + part_agg_kernel<1024>";
  }
  int occupancy = 0;
  if (env_bpc <= 0) scan_dispatch(0, &occupancy);
  const uint32_t step = BLOCK * VH_LANE_ROWS;
  const uint64_t padded = (t->segment_rows + step - 1) / step * step;
  // all blocks co-resident (the compacting kernels need ~100-130 VGPRs: 4 waves/SIMD), units small enough
  // that the static round-robin leaves < 2 % imbalance
  // The compiled kernels that WRITE tuples (DENSE_PART phase 1, hashed partitioning) run best with fewer resident waves than their
  // 56-69 VGPRs allow: every wave keeps a line or an extent open per partition, and what eight blocks per CU keep open no longer
  // stays in L2 until it is complete (profiles/r03/NOTES.md, "Blocks per CU": a 125 M-row C3 shard 0.44 -> 0.39 ms with 3 instead of
  // 6, C5's scan 1.95 -> 1.6 ms with 4 instead of 8, and the scatter behind it finds fewer half-empty extents)
  const int occ_cap = jk && mode == VH_MODE_DENSE_PART ? 3 : jk && hpart ? 4 : 8;
  const int blocks_per_cu = env_bpc > 0 ? env_bpc : occupancy > 0 ? std::min(occupancy, occ_cap) : (BLOCK == 1024 ? 1 : 4);
  uint32_t unit_rows = step;
  const uint64_t want_units = (uint64_t)g_ctx.num_cu * blocks_per_cu * 64;
  while (unit_rows * 2 <= 65536 && unit_rows * 2 <= padded &&
         (uint64_t)nseg * ((padded + unit_rows * 2 - 1) / (unit_rows * 2)) >= want_units) unit_rows *= 2;
  if (env_unit >= (int)step) unit_rows = (uint32_t)env_unit / step * step;
  P.unit_rows = unit_rows;
  P.units_per_seg = (uint32_t)((t->segment_rows + unit_rows - 1) / unit_rows);
  P.nseg = nseg;
  P.total_units = nseg * P.units_per_seg;
  const int env_grid = knobs().grid;
  grid = env_grid > 0 ? env_grid : (int)std::max<uint64_t>(1, std::min<uint64_t>(P.total_units, (uint64_t)g_ctx.num_cu * blocks_per_cu));
  return VH_OK;
}

int QueryBuild::layout_scratch() {
  int rc = VH_OK; (void)rc;
  // ---------------- scratch layout
  ScratchPlan sp;
  // [counters | out_count | output key arrays | output state arrays] is one region: it is read back with a
  // single D2H copy when small, and its head is cleared with a single memset
  const size_t o_counters = sp.take(8 * sizeof(unsigned long long));
  const size_t o_outcount = sp.take(sizeof(unsigned long long));
  r->out_cap = mode == VH_MODE_HASH ? capacity + 1 : G;
  // a big result of the hashed partitioning whose groups nothing has to look at on the device first leaves in chunks, copied out while
  // the later chunks still aggregate: VH_HP_CHUNKS regions of the output columns, each with room for its share of the groups (the mixed
  // key spreads GROUPS evenly over the level-A partitions whatever the rows' skew) and a quarter more; a region that overflows all the
  // same voids the attempt like any pool that runs out
  r->hp_direct = hpart && r->nhaving == 0 && r->topk == 0 && !knobs().hp_list;
  if (r->hp_direct && !device_rows && (knobs().hp_stream > 0 ? capacity >= (1ull << 22) : getenv("VH_TEST_HP_STREAM") != nullptr)) {
    int nch = std::min(knobs().hp_stream > 0 ? knobs().hp_stream : 4, VH_HP_CHUNKS);
    while (HP_FAN % nch) --nch;
    r->hp_chunks = nch;
    r->hp_chunk_rows = capacity / nch + capacity / (4 * nch) + 4096;
    r->out_cap = r->hp_chunk_rows * nch;
    rc = exec_streaming(x);
    if (rc) return rc;
  }
  size_t o_okey[VH_MAX_GROUP], o_ostate[VH_MAX_METRIC];
  for (int i = 0; i < P.ngroup; ++i) o_okey[i] = sp.take(r->out_cap * vh_elem_size(P.g[i].type()));
  for (int j = 0; j < P.nmetric; ++j) o_ostate[j] = sp.take(r->out_cap * vh_elem_size(r->metric_elem[j]));
  r->out_region_off = o_counters;
  r->out_region_bytes = sp.off - o_counters;
  for (int i = 0; i < P.ngroup; ++i) r->off_key[i] = o_okey[i] - o_counters;
  for (int j = 0; j < P.nmetric; ++j) r->off_state[j] = o_ostate[j] - o_counters;
  o_segrows = sp.take(r->plan_words * sizeof(uint32_t));   // [segment snapshot | program | literals]
  size_t o_present = 0, o_hkeys = 0, o_htags = 0;
  size_t o_state[VH_MAX_METRIC];
  table_n = mode == VH_MODE_HASH ? capacity + 1 : P.xcd_stride * nxcd;
  // single-word keys: one record per slot = key + every metric state (8-byte states first), so that an insert and its
  // updates touch ONE line of a table that is far bigger than any cache
  // Only for tables far bigger than the caches: with few, hot groups three atomics on ONE line serialise more than on three
  // (C2 forced onto the hash table, 1 K groups: 1.75 ms with separate arrays, 2.21 ms with records).
  if (mode == VH_MODE_HASH && P.key_words == 1 && (hpart || ((capacity >= (1ull << 22) || (p->flags & VH_PLAN_FORCE_HASH_RECORDS)) && !(p->flags & VH_PLAN_NO_HASH_RECORDS)))) {
    size_t off = 8;
    for (int pass = 0; pass < 2; ++pass)
      for (int j = 0; j < P.nmetric; ++j) {
        const int b = vh_sop_bytes(P.m[j].sop());
        if ((pass == 0) != (b == 8)) continue;
        rec_off[j] = off; off += b;
      }
    off = (off + 7) / 8 * 8;
    if (off <= 64) P.hrec_bytes = (uint32_t)off;
  }
  if (mode == VH_MODE_HASH) {
    o_hkeys = sp.take(P.hrec_bytes ? table_n * P.hrec_bytes : table_n * P.key_words * sizeof(uint64_t));
    if (P.key_words > 1) o_htags = sp.take(table_n * sizeof(uint32_t));
  } else {
    o_present = sp.take(table_n);
  }
  // zero-identity states (every SUM) sit right behind the presence bytes: one memset clears them all
  zero_begin = mode == VH_MODE_HASH ? sp.off : o_present;
  if (P.hrec_bytes) { for (int j = 0; j < P.nmetric; ++j) o_state[j] = o_hkeys + rec_off[j]; }
  else for (int j = 0; j < P.nmetric; ++j) if (P.m[j].ident == 0) o_state[j] = sp.take(table_n * vh_sop_bytes(P.m[j].sop()));
  zero_end = sp.off;
  if (!P.hrec_bytes) for (int j = 0; j < P.nmetric; ++j) if (P.m[j].ident != 0) o_state[j] = sp.take(table_n * vh_sop_bytes(P.m[j].sop()));
  // device top-N: worth it only when the group table is big (small results are read back whole anyway)
  size_t o_tkkeys = 0, o_tkstate = 0, o_okey2[VH_MAX_GROUP] = {}, o_ostate2[VH_MAX_METRIC] = {};
  r->topk_active = r->topk > 0 && r->out_cap > 65536 && !knobs().no_topk;
  if (r->topk_active) {
    o_tkkeys = sp.take(r->out_cap * sizeof(uint64_t));
    o_tkstate = sp.take(sizeof(VhTopkState));
    for (int i = 0; i < P.ngroup; ++i) o_okey2[i] = sp.take(r->out_cap * vh_elem_size(P.g[i].type()));
    for (int j = 0; j < P.nmetric; ++j) o_ostate2[j] = sp.take(r->out_cap * vh_elem_size(r->metric_elem[j]));
  }
  // outputs
  size_t o_tuples = 0, o_emiss = 0, o_epart = 0, o_tuples2 = 0, o_emiss2 = 0, o_epart2 = 0, o_l2 = 0;
  if (hpart) part_tuple_cap = hp_tuple_cap;
  if (mode == VH_MODE_DENSE_PART || hpart) {
    // extent size: big enough that a wave allocates rarely (every allocation is a returning global
    // atomic = a full round trip the wave sits out), small enough that open extents do not waste HBM
    const uint64_t waves = (uint64_t)grid * 4;
    // ... and small enough that a wave fills about four of them per partition: the last extent of every (wave, partition) stays part
    // full, and phase 2 walks part-full extents at the price of full ones (C3: 1250 tuples per wave and partition — extents of 1024
    // were 61 % full on average, of 256 they are 90 %: kernels 2.07-2.12 -> 2.00-2.01 ms, an eighth of the table 0.36-0.38 -> 0.35-0.36)
    uint64_t et = 256;                 // a tile writes whole runs (<= VH_PART_TILE tuples) that must fit a fresh extent
    while (et < 4096 && et * 2 <= part_tuple_cap / (waves * P.npart) / 4) et *= 2;
    if (knobs().ext_tuples) et = std::max(256, knobs().ext_tuples);     // measurement
    if (hpart) et = HP_ET / hp_units;  // (the tiles of hp_scatter_kernel are whole source extents: 64 KB of tuples)
    const uint64_t ext_tuples = et;
    P.ext_tuples = (int32_t)ext_tuples;
    // extents of pool 1 start one 128-byte line further apart than they are long (not the stream pools of the hashed partitioning, whose
    // reader takes extents as whole tiles): see VhPlanDev::ext_stride
    const uint64_t ext_stride = hpart ? ext_tuples : ext_tuples + (P.gid_bits ? ((uint64_t)knobs().ext_pad + 15) / 16 * 16 : (uint64_t)knobs().ext_pad);      // (whole 128-byte lines: 8 two-word tuples, 16 one-word ones)
    P.ext_stride = (int32_t)ext_stride;
    uint64_t max_ext = part_tuple_cap / ext_tuples + waves * (P.npart + VH_EXT_CHUNK) + 64;
    if (max_ext > 0xFFFFFFF0ull) max_ext = 0xFFFFFFF0ull;
    if (!part_tuples_override && getenv("VH_TEST_PART_EXTENTS")) max_ext = std::max(1, atoi(getenv("VH_TEST_PART_EXTENTS")));   // tests: make the first attempt run out of extents
    P.max_extents = (uint32_t)max_ext;
    o_tuples = sp.take(max_ext * ext_stride * P.tw * 8);
    o_emiss = sp.take(max_ext * sizeof(uint16_t));
    o_epart = sp.take(max_ext);
    if (P.nlevel == 2) {
      // pool 2: small extents (4096 ranges x every splitting wave keep one open), sized like pool 1 plus what stays open
      split_bpp = std::max(1, 2 * g_ctx.num_cu / std::max(1, P.npart));     // 256-thread blocks, ~2 per CU whatever the partition count: more waves keep more extents open (4 per CU measured slower)
      // two-word tuples are split a block-wide tile at a time (part_split_tile_kernel): extents of one tile's size, one writer per block
      const bool tiled = P.tw == 2 && !getenv("VH_NO_SPLIT_TILE");
      if (tiled) split_bpp = std::max(1, knobs().split_bpc * g_ctx.num_cu / std::max(1, P.npart));   // a block is one writer: more of them cost less
      const uint64_t et2 = tiled ? VH_SPLIT_TILE_TUPLES : 256;
      P.ext_tuples2 = (int32_t)et2;
      uint64_t max2 = (part_tuple_cap + part_tuple_cap / 4) / et2 + (uint64_t)P.npart * ((uint64_t)split_bpp * (tiled ? 1 : 4) * (64 + VH_EXT_CHUNK) + 1) + 64;
      if (max2 > 0xFFFFFFF0ull) max2 = 0xFFFFFFF0ull;
      if (!part_tuples_override && getenv("VH_TEST_PART_EXTENTS2")) max2 = std::max(1, atoi(getenv("VH_TEST_PART_EXTENTS2")));   // tests: the second pool runs out first
      P.max_extents2 = (uint32_t)max2;
      o_tuples2 = sp.take(max2 * et2 * P.tw * 8);
      o_emiss2 = sp.take(max2 * sizeof(uint16_t));
      o_epart2 = sp.take(max2);
      o_l2 = sp.take((VH_L2_WORDS + VH_MAX_PART) * sizeof(uint32_t));
    }
  }
  // hashed partitioning: the two partitioned pools (vh_hpart.h), their fill / tag arrays and a block of small tables
  hp_meta_bytes = 8 + (size_t)HP_FAN * 4 + (size_t)(2 * HP_FAN + 2) * 4;      // [level-A cursor | tuples per digit | slices + their cursors]
  if (hpart) {
    for (int k = 0; k < 1; ++k) {
      const uint64_t cap = hp_tuple_cap, hp_et = HP_ET / hp_units, hp_es = hp_et + (uint64_t)knobs().ext_pad / hp_units;      // tuples per extent / between extent starts
      // level A: every block may hold an open extent per digit (+ one fresh one per tile boundary); level B: the slices hp_plan_kernel lays out
      uint64_t ma = ((cap / hp_et) / g_ctx.num_cu * 3 / 2 + 2 * HP_FAN + 16) * g_ctx.num_cu;      // one slab per block: its share of the tuples and half again, an open extent per digit, one more per digit for the tails
      uint64_t mb = cap / hp_et + (uint64_t)HP_FAN * (2 * HP_FAN + 9) + 64;
      if (!part_tuples_override && getenv("VH_TEST_PART_EXTENTS2")) mb = std::max(1, atoi(getenv("VH_TEST_PART_EXTENTS2")));   // tests: the last pool runs out
      hpo[k].maxa = ma; hpo[k].maxb = mb;
      hpo[k].ta = sp.take(ma * hp_es * 16 * hp_units); hpo[k].fa = sp.take(ma * 2); hpo[k].ga = sp.take(ma);
      hpo[k].tb = sp.take(mb * hp_es * 16 * hp_units); hpo[k].fb = sp.take(mb * 2); hpo[k].gb = sp.take(mb);
      hpo[k].meta = sp.take(hp_meta_bytes);
    }
    o_hpargs = sp.take(sizeof(VhHpArgs));
  }
  size_t o_fbs[VH_MAX_BITSET] = {};
  for (size_t k = 0; k < r->filter_bitset_cols.size(); ++k) o_fbs[k] = sp.take(std::max<uint32_t>(nseg, 1) * 8);
  size_t o_bsptr[VH_MAX_BITSET][2] = {}, o_dkeys[VH_MAX_BITSET] = {}, o_dtags[VH_MAX_BITSET] = {};
  for (int b = 0; b < P.nbitset; ++b) {
    o_bsptr[b][0] = sp.take(std::max<uint32_t>(nseg, 1) * 8);
    o_bsptr[b][1] = sp.take(std::max<uint32_t>(nseg, 1) * 8);
    if (hpart) continue;             // (its count-distinct lives in the LDS sets of hp_aggregate_kernel)
    // the (group, id) set can never hold more pairs than there are ids in the scanned segments
    uint64_t cap = 1024;
    while (cap < bitset_ids[b] * 2) cap <<= 1;
    P.dset_mask[b] = cap - 1;
    if (P.bs_wide[b]) { o_dkeys[b] = sp.take(cap * 16); o_dtags[b] = sp.take(cap * 4); }
    else {
      if (table_n >= 0xFFFFFFFFull) { return vh_fail(VH_E_UNSUPPORTED, "count-distinct over more than 2^32 group slots"); }
      o_dkeys[b] = sp.take(cap * 8);
    }
  }
  {
    VhPlaceHint ph;        // (only looked at when the scratch buffer has to be allocated anew)
    if ((mode == VH_MODE_DENSE_PART || hpart) && P.nslots > 0) {
      const int gs = P.ngroup > 0 ? (int)P.g[0].slot() : 0;
      for (int q = 0; q < P.npred && q < 4; ++q) { const int ps = (int)P.pred_slot[q]; ph.stream_src[q] = P.colbase[ps]; ph.stream_bytes[q] = (size_t)nseg * P.colstride[ps]; ph.nstream = q + 1; }
      if (!ph.nstream) { ph.stream_src[0] = P.colbase[gs]; ph.stream_bytes[0] = (size_t)nseg * P.colstride[gs]; ph.nstream = 1; }
      ph.gather_src = P.colbase[gs]; ph.gather_bytes = (size_t)nseg * P.colstride[gs];
      ph.gather_bytes -= std::min<size_t>(ph.gather_bytes, 256);      // (a projection's column starts inside its first record)
      ph.pool_off = o_tuples; ph.pool_bytes = (size_t)P.max_extents * (size_t)P.ext_stride * P.tw * 8;
    }
    if (sp.off > x->scratch_bytes && ph.pool_bytes >= ((size_t)128 << 20) && !t->derived_tried && g_preparing && knobs().place_trials >= 2 && (!t->packs.empty() || !t->narrows.empty())) {
      t->derived_tried = true;
      bool moved = false;
      rc = place_with_derived(t, x, sp.off, ph, &moved);
      if (rc) { return rc; }
      if (moved) {       // the plan built so far holds the old addresses of the derived layouts: once more from the top (the scratch buffer is in place)
        holder.reset();
        done = true;
        return query_launch_locked(t, x, p, out, hash_capacity_override, force_hash, part_tuples_override, no_part, plan_only, summary_out, ag, device_rows, hp_passes_override, no_hpart);
      }
    }
    rc = ensure_scratch(x, sp.off, &ph);
  }
  if (rc) { return rc; }
  S = x->scratch;
  P.counters = reinterpret_cast<unsigned long long*>(S + o_counters);
  P.seg_rows = reinterpret_cast<const uint32_t*>(S + o_segrows);
  P.prog = reinterpret_cast<const VhProgOp*>(S + o_segrows + r->seg_words * 4);
  P.lits = reinterpret_cast<const uint64_t*>(S + o_segrows + r->seg_words * 4 + r->h_prog.size() * sizeof(VhProgOp));
  if (mode == VH_MODE_HASH) {
    P.hkeys = reinterpret_cast<uint64_t*>(S + o_hkeys);
    P.htags = P.key_words > 1 ? reinterpret_cast<uint32_t*>(S + o_htags) : nullptr;
  } else {
    P.present = reinterpret_cast<uint8_t*>(S + o_present);
  }
  for (int j = 0; j < P.nmetric; ++j) P.m[j].state = S + o_state[j];
  r->d_out_count = reinterpret_cast<unsigned long long*>(S + o_outcount);
  if (mode == VH_MODE_DENSE_PART || hpart) {
    P.tuples = reinterpret_cast<uint64_t*>(S + o_tuples);
    P.extent_missing = reinterpret_cast<uint16_t*>(S + o_emiss);
    P.extent_part = reinterpret_cast<uint8_t*>(S + o_epart);
    if (P.nlevel == 2) {
      P.tuples2 = reinterpret_cast<uint64_t*>(S + o_tuples2);
      P.extent_missing2 = reinterpret_cast<uint16_t*>(S + o_emiss2);
      P.extent_part2 = reinterpret_cast<uint8_t*>(S + o_epart2);
      P.l2 = reinterpret_cast<uint32_t*>(S + o_l2);
    }
  }
  for (size_t k = 0; k < r->filter_bitset_cols.size(); ++k) {
    const VhColumn& c = t->cols[r->filter_bitset_cols[k]];
    for (uint32_t sgi : live) if (!c.bs_offsets[sgi]) return vh_fail(VH_E_INVALID, "bitset column %d of segment %u was never synced", r->filter_bitset_cols[k], sgi);
    P.fbs_offs[k] = reinterpret_cast<const uint64_t* const*>(S + o_fbs[k]);
    if (nseg) HIP_TRY(hipMemcpy(S + o_fbs[k], c.bs_offsets.data(), nseg * 8, hipMemcpyHostToDevice));
  }
  if (P.nbitset) {
    for (int b = 0; b < P.nbitset; ++b) {
      P.dset_keys[b] = hpart ? nullptr : reinterpret_cast<uint64_t*>(S + o_dkeys[b]);
      P.dset_tags[b] = P.bs_wide[b] && !hpart ? reinterpret_cast<uint32_t*>(S + o_dtags[b]) : nullptr;
      const VhColumn& c = t->cols[bitset_col[b]];
      P.bs_offs[b] = reinterpret_cast<const uint64_t* const*>(S + o_bsptr[b][0]);
      P.bs_vals[b] = reinterpret_cast<const void* const*>(S + o_bsptr[b][1]);
      if (nseg) {
        HIP_TRY(hipMemcpy(S + o_bsptr[b][0], c.bs_offsets.data(), nseg * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(S + o_bsptr[b][1], c.bs_values.data(), nseg * 8, hipMemcpyHostToDevice));
      }
    }
  }
  for (int i = 0; i < P.ngroup; ++i) r->d_out_key[i] = S + o_okey[i];
  for (int j = 0; j < P.nmetric; ++j) r->d_out_state[j] = S + o_ostate[j];
  if (r->topk_active) {
    r->d_topk_keys = reinterpret_cast<uint64_t*>(S + o_tkkeys);
    r->d_topk_state = reinterpret_cast<VhTopkState*>(S + o_tkstate);
    for (int i = 0; i < P.ngroup; ++i) r->d_out_key2[i] = S + o_okey2[i];
    for (int j = 0; j < P.nmetric; ++j) r->d_out_state2[j] = S + o_ostate2[j];
  }
  return VH_OK;
}

int QueryBuild::launch() {
  int rc = VH_OK; (void)rc;
  // ---------------- init + launch
  hipStream_t st = x->stream();
  HIP_TRY(hipEventRecord(x->ev[0], st));
  VhInitArgs IA{};              // everything that is cleared goes into one launch (init_regions_kernel)
  auto clear = [&](void* ptr, size_t bytes, uint32_t byte_pattern) {
    if (!bytes) return;
    const uint64_t units = (bytes + 15) / 16;                       // (regions are padded to 256 B: rounding up stays inside)
    if (IA.n == VH_INIT_MAX) { (void)hipMemsetAsync(ptr, (int)(byte_pattern & 0xFFu), bytes, st); return; }
    IA.p[IA.n] = static_cast<char*>(ptr); IA.end[IA.n] = (IA.n ? IA.end[IA.n - 1] : 0) + units; IA.pat[IA.n] = byte_pattern * 0x01010101u; ++IA.n;
  };
  clear(P.counters, 512, 0);   // counters + out_count (adjacent 256 B slots)
  HIP_TRY(hipMemcpyAsync(S + o_segrows, x->h_segrows, r->plan_words * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  if (mode == VH_MODE_HASH && !hpart) {      // (hashed partitioning writes its group records as a compact list: nothing to pre-fill)
    if (P.hrec_bytes) {          // records: empty key + the metrics' identities, one template for every slot
      VhRecordTemplate T{};
      T.w[0] = VH_HASH_EMPTY;
      for (int j = 0; j < P.nmetric; ++j)
        memcpy(reinterpret_cast<char*>(T.w) + rec_off[j], &P.m[j].ident, vh_sop_bytes(P.m[j].sop()));
      const uint64_t nwords = table_n * (P.hrec_bytes / 8);
      hipLaunchKernelGGL(fill_records_kernel, dim3((unsigned)std::min<uint64_t>((nwords + 255) / 256, (uint64_t)g_ctx.num_cu * 16)), dim3(256), 0, st,
                         P.hkeys, nwords, P.hrec_bytes / 8, T);
      HIP_TRY(hipGetLastError());
    }
    else if (P.key_words == 1) clear(P.hkeys, table_n * sizeof(uint64_t), 0xFF);
    else clear(P.htags, table_n * sizeof(uint32_t), 0);
  }
  // (DENSE_PART whose blocks each keep a private copy of their range store EVERY group of every copy, present or not: clearing 19
  // copies of C3's tables, 30 MB, before every query was two thirds of this launch's 17 us. A range's sole block stores present groups only.)
  const bool part_owned = mode == VH_MODE_DENSE_PART && part_bpp > 1 && nxcd == part_bpp && !knobs().skip_phase2 && P.total_units != 0;      // (no units: phase 2 does not run and nobody stores the copies — they are cleared like any table)
  if (zero_end > zero_begin && !part_owned) clear(S + zero_begin, zero_end - zero_begin, 0);
  r->zero_begin = S + zero_begin; r->zero_end = S + zero_end;
  for (int b = 0; b < P.nbitset && !hpart; ++b) {
    if (P.bs_wide[b]) clear(P.dset_tags[b], (P.dset_mask[b] + 1) * 4, 0);
    else clear(P.dset_keys[b], (P.dset_mask[b] + 1) * 8, 0xFF);
  }
  VhHpArgs* d_hpargs = nullptr;
  if (hpart) {          // the pools behind the scan (vh_hpart.h): descriptors for the kernels, fill arrays and small tables cleared with everything else
    VhHpArgs& HA = r->hp_args;
    memset(&HA, 0, sizeof(HA));
    HA.units = hp_units; HA.pk = hp_pack ? 1 : 0; HA.pk_pbits = hp_pbits; HA.pk_idbits = hp_idbits;
    HA.passes = P.hp_passes; HA.gslots = P.hp_gslots; HA.sslots = P.hp_sslots; HA.keys_off = P.hp_keys_off; HA.set_off = P.hp_set_off;
    HA.bitset_j = -1;
    for (int j = 0; j < P.nmetric; ++j) if (P.m[j].sop() == SOP_BITSET) HA.bitset_j = j;
    HA.list_cap = capacity; HA.chunk = hp_chunk; HA.ablate = knobs().hp_ablate;
    // no HAVING and no top-N to look at the groups first: the aggregation kernel emits them itself (C5: no 0.85 GB list, no 0.85 ms kernel)
    HA.direct = r->hp_direct ? 1 : 0; HA.ngroup = P.ngroup; HA.out_count = r->d_out_count;
    HA.nchunks = r->hp_chunks; HA.chunk_rows = r->hp_chunk_rows;
    for (int i = 0; i < P.ngroup; ++i) { HA.out_key[i] = r->d_out_key[i]; HA.gkey_shift[i] = P.g[i].key_shift(); HA.gesize[i] = (uint32_t)vh_elem_size(P.g[i].type()); }
    for (int j = 0; j < P.nmetric; ++j) { HA.out_state[j] = r->d_out_state[j]; HA.mesize[j] = (uint32_t)vh_elem_size(r->metric_elem[j]); }
    for (int k = 0; k < 1; ++k) {
      VhHpKind& K = HA.k[k];
      char* meta = S + hpo[k].meta;
      K.z.tuples = P.tuples; K.z.fill = P.extent_missing; K.z.tag = P.extent_part;
      K.z.max_extents = P.max_extents; K.z.stream = 1; K.z.cursor = P.counters + 5; K.z.stride = (uint32_t)(HP_ET / hp_units);
      K.a.stride = K.b.stride = (uint32_t)(HP_ET / hp_units) + (uint32_t)knobs().ext_pad / (uint32_t)hp_units;
      K.a.tuples = reinterpret_cast<uint64_t*>(S + hpo[k].ta); K.a.fill = reinterpret_cast<uint16_t*>(S + hpo[k].fa); K.a.tag = reinterpret_cast<uint8_t*>(S + hpo[k].ga);
      K.a.max_extents = (uint32_t)std::min<uint64_t>(hpo[k].maxa, 0xFFFFFFF0ull); K.a.cursor = nullptr;      // (handed out in one slab per block of level A)
      K.b.tuples = reinterpret_cast<uint64_t*>(S + hpo[k].tb); K.b.fill = reinterpret_cast<uint16_t*>(S + hpo[k].fb); K.b.tag = reinterpret_cast<uint8_t*>(S + hpo[k].gb);
      K.b.max_extents = (uint32_t)std::min<uint64_t>(hpo[k].maxb, 0xFFFFFFF0ull); K.b.cursor = nullptr;
      K.count = reinterpret_cast<uint32_t*>(meta + 8);
      K.slice = reinterpret_cast<uint32_t*>(meta + 8 + (size_t)HP_FAN * 4);
      clear(K.a.fill, (size_t)K.a.max_extents * 2, 0);
      clear(K.b.fill, (size_t)K.b.max_extents * 2, 0);
      clear(meta, hp_meta_bytes, 0);
    }
    d_hpargs = reinterpret_cast<VhHpArgs*>(S + o_hpargs);
    HIP_TRY(hipMemcpyAsync(d_hpargs, &HA, sizeof(HA), hipMemcpyHostToDevice, st));
  }
  if (mode == VH_MODE_DENSE_PART || hpart) {
    clear(P.extent_missing, (size_t)P.max_extents * sizeof(uint16_t), 0);
    clear(P.extent_part, (size_t)P.max_extents, 0xFF);
    if (P.nlevel == 2) {
      clear(P.extent_missing2, (size_t)P.max_extents2 * sizeof(uint16_t), 0);
      clear(P.extent_part2, (size_t)P.max_extents2, 0xFF);
      clear(P.l2, (VH_L2_WORDS + VH_MAX_PART) * sizeof(uint32_t), 0);
    }
  }
  for (int j = 0; j < P.nmetric; ++j) {
    if (P.m[j].ident == 0 || P.hrec_bytes || hpart) continue;
    rc = fill_states(P.m[j].state, table_n, vh_sop_bytes(P.m[j].sop()), P.m[j].ident, st);
    if (rc) { return rc; }
  }
  if (IA.n) {
    const uint64_t units = IA.end[IA.n - 1];
    hipLaunchKernelGGL(init_regions_kernel, dim3((unsigned)std::min<uint64_t>((units + 255) / 256, (uint64_t)g_ctx.num_cu * 16)), dim3(256), 0, st, IA);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipEventRecord(x->ev[1], st));
  if (lanes)       // the lanes kernels read 4-byte predicate columns only (vh_preload<NP, false>)
    for (int k = 0; k < P.npred; ++k) if (P.pred_width[k] != 4) { P.pred_slot[k] = (uint8_t)pred_wide_slot[k]; P.pred_width[k] = 4; }
  bool narrowed = false;
  for (int k = 0; k < P.npred; ++k) narrowed |= P.pred_width[k] != 4;
  if (jk) { narrowed = false; for (int k = 0; k < jshape.npred; ++k) narrowed |= jshape.pred[k].width != vh_elem_size(jshape.pred[k].type); }
  r->hpart = hpart;
  r->info.reserved = (hpart ? 64 : 0) | (fastj || jk ? 1 : 0) | (lanes ? 2 : 0) | (P.lds_hash_slots ? 4 : 0) | (packed ? 8 : 0) | (fastj && narrowed ? 16 : 0) | (jk ? 32 : 0) | (packed && packed_compressed ? 128 : 0) | (hpart && hp_pack ? 256 : 0) | ((mode == VH_MODE_DENSE_PART || hpart) && x->scratch_placed ? 512 : 0) | (mode == VH_MODE_DENSE_PART && P.gid_bits ? 1024 : 0);
  if (r->hp_chunks) memset(x->h_chunk, 0, VH_HP_CHUNKS * sizeof(unsigned long long));      // (what the context's previous query left there)
  if (P.total_units) {
    scan_dispatch(grid, nullptr);
    if (hpart) {
      vh_launch_hpart(P, d_hpargs, hp_units, g_ctx.num_cu, st);
      if (!r->hp_chunks) HIP_TRY(vh_jit_launch_hpagg(jk, P, d_hpargs, hp_bpp, 0, HP_FAN * hp_bpp, lds_table, st));
      else {
        // chunk c = level-A partitions [c * per, (c + 1) * per), alternately on the query's stream and on `aux` (both behind level B), each
        // followed by its row count into pinned memory and an event the host waits for (result_finalize)
        const int per = HP_FAN / r->hp_chunks;
        HIP_TRY(hipEventRecord(x->ev_fork, st));
        HIP_TRY(hipStreamWaitEvent(x->aux, x->ev_fork, 0));
        for (int c = 0; c < r->hp_chunks; ++c) {
          hipStream_t cs = (c & 1) ? x->aux : st;
          HIP_TRY(vh_jit_launch_hpagg(jk, P, d_hpargs, hp_bpp, c * per, per * hp_bpp, lds_table, cs));
          hipLaunchKernelGGL(publish_count_kernel, dim3(1), dim3(64), 0, cs, x->h_chunk + c, r->d_out_count + c);
          HIP_TRY(hipEventRecord(x->ev_chunk[c], cs));
        }
        for (int c = 0; c < r->hp_chunks; ++c) if (c & 1) HIP_TRY(hipStreamWaitEvent(st, x->ev_chunk[c], 0));      // the query's stream ends behind every chunk
      }
    }
    if (mode == VH_MODE_DENSE_PART) {
      const bool skip_phase2 = knobs().skip_phase2;     // measurement only (wrong results): phase 1 alone between the events
      if (P.nlevel == 2 && !skip_phase2) vh_launch_part_split(P, split_bpp, st);
      if (!skip_phase2) vh_launch_part_agg(P, part_bpp, lds_table, st);
    }
  }
  HIP_TRY(hipEventRecord(x->ev[2], st));
  HIP_TRY(hipGetLastError());
  if (mode != VH_MODE_HASH && nxcd > 1) {
    VhMergeArgs A{};
    A.nmetric = P.nmetric; A.nxcd = nxcd; A.G = G; A.xcd_stride = P.xcd_stride; A.present = P.present;
    A.present_carrier = P.present_carrier;
    for (int j = 0; j < P.nmetric; ++j) { A.state[j] = P.m[j].state; A.sop[j] = P.m[j].sop(); }
    hipLaunchKernelGGL(dense_merge_kernel, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, st, A);
    HIP_TRY(hipGetLastError());
  }
  *out = holder.release();
  done = true;
  return VH_OK;
}

static int query_launch_locked(vh_table* t, VhExec* x, const vh_plan* p, vh_result** out, uint64_t hash_capacity_override,
                               bool force_hash, uint64_t part_tuples_override = 0, bool no_part = false,
                               bool plan_only = false, VhSummary* summary_out = nullptr, const VhAgreed* ag = nullptr,
                               bool device_rows = false, uint32_t hp_passes_override = 0, bool no_hpart = false) {
  QueryBuild b(t, x, p, out, hash_capacity_override, force_hash, part_tuples_override, no_part, plan_only, summary_out, ag, device_rows,
               hp_passes_override, no_hpart);
  int (QueryBuild::* const steps[])() = {&QueryBuild::shape_filter, &QueryBuild::snapshot_segments, &QueryBuild::shape_groups, &QueryBuild::shape_metrics,
                                         &QueryBuild::choose_organisation, &QueryBuild::plan_hashed_partitioning, &QueryBuild::choose_projection,
                                         &QueryBuild::compile_kernel, &QueryBuild::decompose_work, &QueryBuild::layout_scratch, &QueryBuild::launch};
  for (auto step : steps) {
    if (int rc = (b.*step)()) return rc;
    if (b.done) return VH_OK;
  }
  return VH_OK;
}

extern "C" int vh_result_device_buffers(vh_result* r, vh_device_buffer* bufs, int32_t max_bufs, int32_t* nbufs) {
  if (!r || !bufs || !nbufs) return vh_fail(VH_E_INVALID, "null argument");
  if (r->plan.nbitset) return vh_fail(VH_E_UNSUPPORTED, "count-distinct partials are cardinalities: they cannot be reduced across GPUs");
  if (r->mode == VH_MODE_HASH) return vh_fail(VH_E_UNSUPPORTED, "hash-path partials are exchanged by key, not reduced in place");
  const VhPlanDev& P = r->plan;
  int n = 0;
  if (max_bufs < P.nmetric + 1) return vh_fail(VH_E_INVALID, "need %d buffers", P.nmetric + 1);
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
  hipStream_t st = r->exec->stream();
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
    hipStream_t st = r->exec->stream();
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
  hipStream_t st = r->exec->stream();
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

// returns VH_OK, or a positive "retry" request: 1 = grow hash table, 2 = fall back to hash
static int result_finalize(vh_result* r, int* retry) {
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
  if (r->hp_direct && r->nhaving == 0 && !r->topk_active) { /* hp_aggregate_kernel wrote the output columns and counted the rows */ }
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
    hipLaunchKernelGGL(publish_header_kernel, dim3(1), dim3(64), 0, st, head, reinterpret_cast<const unsigned long long*>(D));
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
  if (!one_shot) {
    if (!r->hp_chunks) {           // a big result in one piece: the staging buffer is sized for the rows there are
      L = packed_for(ng);
      if (int src = stage(L.bytes)) return src;
      if (ng) { if (int crc = copy_rows(L, 0, 0, ng, r->topk_active, st)) return crc; }
      HIP_TRY(hipEventRecord(x->ev[3], st));
      HIP_TRY(wait_event_spinning(x->ev[3]));
    }
    memcpy(x->h_out[slot], head, 512);
    for (int i = 0; i < P.ngroup; ++i) r->off_key[i] = L.key[i];      // (the host view: where vh_result_view finds the columns)
    for (int j = 0; j < P.nmetric; ++j) r->off_state[j] = L.state[j];
  }
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
    if (r->plan.hp_passes >= 64) rp->no_hpart = true;        // ... skewed beyond help: the plain hash table
    else rp->hp_passes = (uint32_t)r->plan.hp_passes * 4;
    if (rp->hp_passes > 64) rp->hp_passes = 64;
  }
  else if (retry == 3 && r->hpart) {                         // hashed partitioning ran out of tuple extents: size for the survivors it counted, then give up
    if (rp->part_override) rp->no_hpart = true;
    else rp->part_override = std::max<uint64_t>(r->info.passed_recs + r->info.passed_recs / 16 + 1024, 1ull << 16);
  }
  else if (retry == 3) {                                     // tuple extents exhausted: more room, then give up on partitioning
    // the attempt counted its survivors even though it dropped their tuples: the next one is sized for exactly that many
    const uint64_t had = (uint64_t)r->plan.max_extents * r->plan.ext_tuples;
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

extern "C" int vh_query_agg(vh_table* t, const vh_plan* plan, vh_result** out) {
  if (!t || !plan || !out) return vh_fail(VH_E_INVALID, "null argument");
  VH_ENTER();
  if (plan->nmetrics > VH_MAX_METRIC - 1 && plan->metrics) return query_agg_multipass(t, plan, out);
  VhExec* x = nullptr;
  if (int rc = exec_acquire(t, &x)) return rc;
  VhReplan rp;
  int rc = VH_OK;
  for (uint32_t attempt = 0; attempt < 12; ++attempt) {
    vh_result* r = nullptr;
    // planned and launched under the table lock; the wait for the device and the read-back happen outside it, so
    // queries of other threads on this table run meanwhile (each on its own context)
    { std::lock_guard<std::mutex> lk(t->mu); rc = query_launch_locked(t, x, plan, &r, rp.cap_override, rp.force_hash, rp.part_override, rp.no_part, false, nullptr, nullptr, false, rp.hp_passes, rp.no_hpart); }
    if (rc) break;
    r->exec = x;
    int retry = 0;
    rc = result_finalize(r, &retry);
    if (rc) { r->exec = nullptr; delete r; break; }
    if (!retry) {
      r->info.retries = attempt;
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

// First-use costs paid up front: the scan kernel compiled for the plan's shape (1-2 s of hipRTC, or milliseconds from the disk
// cache), the payload projection and the narrow predicate copies a selective query reads (built at once instead of after VH_AUTO_PACK /
// VH_AUTO_NARROW uses), and — only here — a tuple pool placed by measurement (place_search: bounded to half of the free memory / 48 GB, 8
// candidates; everything but the winner is released before the call returns). The reference's analogue is Compiler::Compile running when a
// query shape is first seen (src/codegen/compiler.cc:97-144, QueryStats::compile_time); a caller that knows its hot shapes at table-load
// time runs them through here. The plan is executed (up to three times: a narrow copy, then a projection, then the pool can appear) and
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
  return VH_OK;
}

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

extern "C" int vh_measure_read_bandwidth(uint64_t bytes, int32_t iters, double* bytes_per_sec) {
  if (!g_ctx.inited || !bytes_per_sec || iters <= 0) return vh_fail(VH_E_INVALID, "bad argument");
  VH_ENTER();
  bytes = bytes / 16 * 16;
  char* buf = nullptr; unsigned long long* sink = nullptr;
  HIP_TRY(hipMalloc(&buf, bytes));
  HIP_TRY(hipMalloc(&sink, 8));
  HIP_TRY(hipMemsetAsync(buf, 1, bytes, g_ctx.stream));
  HIP_TRY(hipMemsetAsync(sink, 0, 8, g_ctx.stream));
  hipEvent_t a, b;
  HIP_TRY(hipEventCreate(&a)); HIP_TRY(hipEventCreate(&b));
  const int grid = g_ctx.num_cu * knobs().bw_blocks_per_cu;   // 256-thread blocks: 8 per CU = 8 waves/SIMD
  hipLaunchKernelGGL(read_bw_kernel, dim3(grid), dim3(256), 0, g_ctx.stream, (const vh_u32x4*)buf, bytes / 16, sink);
  HIP_TRY(hipEventRecord(a, g_ctx.stream));
  for (int i = 0; i < iters; ++i)
    hipLaunchKernelGGL(read_bw_kernel, dim3(grid), dim3(256), 0, g_ctx.stream, (const vh_u32x4*)buf, bytes / 16, sink);
  HIP_TRY(hipEventRecord(b, g_ctx.stream));
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, a, b));
  *bytes_per_sec = (double)bytes * iters / (ms * 1e-3);
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  (void)hipFree(buf); (void)hipFree(sink);
  return VH_OK;
}

#include "vh_sharded.h"

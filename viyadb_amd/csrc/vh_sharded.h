// vh_sharded.h — vh_comm_* and vh_query_agg_sharded (include/viya_hip.h, "one query over a table sharded across GPUs").
// Included at the end of viya_hip.hip only: it works on the internals of that file (vh_table, vh_result, the planner).
//
// What it replaces in the reference: the cluster merge of src/cluster/query/agg_runner.cc:66-140 — worker queries, TSV over
// HTTP, re-upsert into a temporary table on the controller, the original query minus its filter on top. Same algebra
// (partial aggregates merge by re-aggregation: SUM of sums and counts, MIN, MAX, set union), but inside one node the
// partial tables never leave HBM: dense ones are reduced in place with RCCL, sparse ones are exchanged by key owner.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

// ------------------------------------------------------------------ RCCL, resolved at run time
// librccl.so.1 is dlopen'ed on first use: the library loads (and its CPU-side tests run) on hosts without RCCL, and in a
// process that already carries a copy of it (torch) the same instance is used.
struct VhRccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static VhRccl g_rccl;
static std::mutex g_rccl_mu;

static int rccl_load() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.handle) return VH_OK;
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return vh_fail(VH_E_UNSUPPORTED, "RCCL is not available: %s", dlerror());
#define VH_NCCL_SYM(field, name)                                                                   \
  *reinterpret_cast<void**>(&g_rccl.field) = dlsym(h, name);                                       \
  if (!g_rccl.field) return vh_fail(VH_E_UNSUPPORTED, "librccl has no symbol %s", name)
  VH_NCCL_SYM(GetUniqueId, "ncclGetUniqueId"); VH_NCCL_SYM(CommInitRank, "ncclCommInitRank"); VH_NCCL_SYM(CommDestroy, "ncclCommDestroy");
  VH_NCCL_SYM(AllGather, "ncclAllGather"); VH_NCCL_SYM(AllReduce, "ncclAllReduce"); VH_NCCL_SYM(Reduce, "ncclReduce");
  VH_NCCL_SYM(Send, "ncclSend"); VH_NCCL_SYM(Recv, "ncclRecv"); VH_NCCL_SYM(GroupStart, "ncclGroupStart"); VH_NCCL_SYM(GroupEnd, "ncclGroupEnd");
  VH_NCCL_SYM(GetErrorString, "ncclGetErrorString");
  VH_NCCL_SYM(CommCount, "ncclCommCount"); VH_NCCL_SYM(CommCuDevice, "ncclCommCuDevice"); VH_NCCL_SYM(CommUserRank, "ncclCommUserRank");
#undef VH_NCCL_SYM
  g_rccl.handle = h;
  return VH_OK;
}
#define NCCL_TRY(expr)                                                                                         \
  do {                                                                                                         \
    ncclResult_t e_ = (expr);                                                                                  \
    if (e_ != ncclSuccess) return vh_fail(VH_E_DEVICE, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// ------------------------------------------------------------------ communicator
#define VH_FLAG_WORDS 16
struct vh_comm {
  int rank = 0, world = 1;
  vh_comm_ops ops{};
  ncclComm_t nccl = nullptr;                         // RCCL transport (ops point at the rccl_* functions below, ctx = this)
  hipStream_t stream = nullptr;                      // RCCL transport: host-level all-gathers
  char* d_stage = nullptr; size_t d_stage_bytes = 0; // RCCL transport: device staging of host all-gathers
  unsigned long long* d_flags = nullptr;             // verdict words, all-reduced (SUM) after every attempt
  unsigned long long* h_flags = nullptr;             // pinned copy
  char* d_void = nullptr; size_t d_void_bytes = 0;   // stand-in state arrays of a rank whose attempt is void (fused_dense_step): as big as the largest recorded dense shape,
                                                     // allocated before the verdict of the query that records the shape (a rank that cannot says so IN the verdict:
                                                     // all ranks record, or none) — never inside a collective group
  std::mutex mu;                                     // one sharded query at a time per communicator (collectives must not interleave)
  // Agreements of earlier queries, by plan. Entries appear and disappear at collective points only, so every rank holds the
  // same set: a query whose plan is in the cache skips the all-gather of step 1 (a 128-byte verdict all-reduce is then the
  // only host-visible collective of a dense query). A rank whose table changed since still plans with the cached agreement
  // and says so in the verdict; all ranks then drop the entry and agree afresh.
  struct BufShape { uint64_t count; int32_t elem, reduce; };
  struct Agreement {
    VhAgreed ag; uint64_t local_state;                         // local_state: plan_local_state() of THIS rank when the agreement was made
    // the partial table of a DENSE query planned from this agreement, as vh_result_device_buffers lists it — the same on every rank (same
    // plan, same agreed digit ranges), recorded when a query of this plan first went through on all ranks. Known: the query runs as ONE
    // stream-ordered sequence (scan, verdict and state arrays in one collective group, emission) with a single host wait at its end.
    bool dense_known = false; int dense_mode = 0; std::vector<BufShape> dense;
  };
  std::map<std::string, Agreement> agreed;
};

// The cache key must be the SAME on every rank that issues the same query, or ranks disagree on whether step 1's all-gather
// happens and the collectives no longer match (all-gather against all-reduce on one communicator). So it holds only what all
// ranks necessarily share — the structure of the query: filter nodes without their padding, group columns without
// dictionary sizes or rollup boundaries, metrics, HAVING, flags, top-N. Everything a rank may see differently — its segment
// snapshot, its literals (a `now`-derived bound), its dictionaries' sizes, its table's contents — goes into plan_local_state
// instead: a rank whose local state differs from the one the agreement was made under still takes the cached agreement, says
// "changed" in the verdict, and ALL ranks then drop the entry and agree afresh.
static std::string plan_signature(const vh_plan* p) {
  std::string k;
  auto put = [&](long long v) { k += std::to_string(v); k.push_back(','); };
  put(p->nfilter);
  for (int i = 0; i < p->nfilter; ++i) { const vh_filter_node& n = p->filter[i]; put(n.kind); put(n.col); put(n.op); put(n.count); put(n.lit); }
  put(p->nlits); put(p->ngroups);
  for (int i = 0; i < p->ngroups; ++i) {
    const vh_group_col& g = p->groups[i];
    put(g.col); put(g.granularity); put(g.nrollup); put(g.micro);
    for (int r = 0; r < g.nrollup && r < VH_MAX_ROLLUP; ++r) put(g.rollup_unit[r]);
  }
  put(p->nmetrics);
  for (int j = 0; j < p->nmetrics; ++j) put(p->metrics[j]);
  put(p->nhaving);
  for (int i = 0; i < p->nhaving; ++i) { const vh_filter_node& n = p->having[i]; put(n.kind); put(n.col); put(n.op); put(n.count); put(n.lit); }
  put(p->flags); put((long long)p->groups_hint); put(p->top_col); put(p->top_desc); put((long long)p->top_k);
  return k;
}
static uint64_t plan_local_state(const vh_plan* p, const vh_table* t) {
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](const void* d, size_t n) { const unsigned char* c = static_cast<const unsigned char*>(d); for (size_t i = 0; i < n; ++i) h = (h ^ c[i]) * 1099511628211ull; };
  for (int i = 0; i < p->nlits; ++i) mix(&p->lits[i].u64, 8);
  for (int i = 0; i < p->ngroups; ++i) {
    mix(&p->groups[i].cardinality, 8);
    for (int r = 0; r < p->groups[i].nrollup && r < VH_MAX_ROLLUP; ++r) mix(&p->groups[i].rollup_before[r], 8);
  }
  const uint64_t ns = p->seg_rows ? p->nseg : ~0ull;
  mix(&ns, 8);
  if (p->seg_rows) mix(p->seg_rows, sizeof(uint64_t) * p->nseg);
  mix(&t->sync_epoch, 8);
  return h;
}

static int rccl_allgather_host(void* ctx, const void* send, void* recv, uint64_t bytes) {
  vh_comm* c = static_cast<vh_comm*>(ctx);
  const size_t need = (size_t)bytes * (c->world + 1);
  if (need > c->d_stage_bytes) {
    if (c->d_stage) HIP_TRY(hipFree(c->d_stage));
    c->d_stage = nullptr; c->d_stage_bytes = 0;
    HIP_TRY(hipMalloc(&c->d_stage, need * 2));
    c->d_stage_bytes = need * 2;
  }
  char* d_send = c->d_stage + (size_t)bytes * c->world;
  HIP_TRY(hipMemcpyAsync(d_send, send, bytes, hipMemcpyHostToDevice, c->stream));
  NCCL_TRY(g_rccl.AllGather(d_send, c->d_stage, bytes, ncclInt8, c->nccl, c->stream));
  HIP_TRY(hipMemcpyAsync(recv, c->d_stage, (size_t)bytes * c->world, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return VH_OK;
}
static int rccl_reduce_device(void* ctx, void* buf, uint64_t count, int32_t elem, int32_t op, int32_t root, void* stream) {
  vh_comm* c = static_cast<vh_comm*>(ctx);
  ncclDataType_t dt;
  switch (elem) {
    case VH_U8: dt = ncclUint8; break;   case VH_I8: dt = ncclInt8; break;
    case VH_U32: dt = ncclUint32; break; case VH_I32: dt = ncclInt32; break;
    case VH_U64: dt = ncclUint64; break; case VH_I64: dt = ncclInt64; break;
    case VH_F32: dt = ncclFloat32; break; case VH_F64: dt = ncclFloat64; break;
    default: return vh_fail(VH_E_UNSUPPORTED, "no RCCL data type for element type %d", elem);
  }
  const ncclRedOp_t ro = op == VH_RED_MIN ? ncclMin : op == VH_RED_MAX ? ncclMax : ncclSum;   // unsigned MIN / MAX are native: ncclUint32 / ncclUint64
  if (root < 0) NCCL_TRY(g_rccl.AllReduce(buf, buf, count, dt, ro, c->nccl, (hipStream_t)stream));
  else NCCL_TRY(g_rccl.Reduce(buf, buf, count, dt, ro, root, c->nccl, (hipStream_t)stream));
  return VH_OK;
}
static int rccl_alltoallv_device(void* ctx, int32_t ncols, const void* const* send, void* const* recv, const uint32_t* esize,
                                 const uint64_t* send_off, const uint64_t* recv_off, void* stream) {
  vh_comm* c = static_cast<vh_comm*>(ctx);
  // ONE group for every column and peer: grouped ncclSend / ncclRecv = one launch, every xGMI link busy at once
  NCCL_TRY(g_rccl.GroupStart());
  for (int32_t k = 0; k < ncols; ++k) {
    for (int p = 0; p < c->world; ++p) {
      const uint64_t ns = send_off[p + 1] - send_off[p], nr = recv_off[p + 1] - recv_off[p];
      if (ns) NCCL_TRY(g_rccl.Send(static_cast<const char*>(send[k]) + send_off[p] * esize[k], ns * esize[k], ncclInt8, p, c->nccl, (hipStream_t)stream));
      if (nr) NCCL_TRY(g_rccl.Recv(static_cast<char*>(recv[k]) + recv_off[p] * esize[k], nr * esize[k], ncclInt8, p, c->nccl, (hipStream_t)stream));
    }
  }
  NCCL_TRY(g_rccl.GroupEnd());
  return VH_OK;
}

static int comm_common_init(vh_comm* c) {
  HIP_TRY(hipMalloc((void**)&c->d_flags, VH_FLAG_WORDS * sizeof(unsigned long long)));
  HIP_TRY(hipHostMalloc((void**)&c->h_flags, VH_FLAG_WORDS * sizeof(unsigned long long), hipHostMallocDefault));
  return VH_OK;
}

extern "C" int vh_comm_unique_id(void* id_out) {
  if (!id_out) return vh_fail(VH_E_INVALID, "null argument");
  static_assert(sizeof(ncclUniqueId) == VH_COMM_ID_BYTES, "ncclUniqueId size");
  if (int rc = rccl_load()) return rc;
  VH_ENTER();
  ncclUniqueId id;
  NCCL_TRY(g_rccl.GetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return VH_OK;
}

extern "C" void vh_comm_destroy(vh_comm* c) {
  if (!c) return;
  VH_ENTER();
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->nccl) (void)g_rccl.CommDestroy(c->nccl);
  if (c->d_stage) (void)hipFree(c->d_stage);
  if (c->d_void) (void)hipFree(c->d_void);
  if (c->d_flags) (void)hipFree(c->d_flags);
  if (c->h_flags) (void)hipHostFree(c->h_flags);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" int vh_comm_init(const void* id, int32_t rank, int32_t world, vh_comm** out) {
  if (!id || !out || world < 1 || rank < 0 || rank >= world || world > 64) return vh_fail(VH_E_INVALID, "vh_comm_init: bad argument");
  if (!g_ctx.inited) return vh_fail(VH_E_INVALID, "vh_init has not been called");
  if (int rc = rccl_load()) return rc;
  VH_ENTER();
  std::unique_ptr<vh_comm> c(new vh_comm());
  c->rank = rank; c->world = world;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  NCCL_TRY(g_rccl.CommInitRank(&c->nccl, world, uid, rank));
  hipError_t he = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  int rc = he == hipSuccess ? comm_common_init(c.get()) : vh_fail(VH_E_DEVICE, "vh_comm_init: %s", hipGetErrorString(he));
  if (rc) { vh_comm_destroy(c.release()); return rc; }
  c->ops.ctx = c.get();
  c->ops.allgather_host = rccl_allgather_host;
  c->ops.reduce_device = rccl_reduce_device;
  c->ops.alltoallv_device = rccl_alltoallv_device;
  *out = c.release();
  return VH_OK;
}

extern "C" int vh_comm_init_custom(const vh_comm_ops* ops, int32_t rank, int32_t world, vh_comm** out) {
  if (!ops || !out || !ops->allgather_host || !ops->reduce_device || !ops->alltoallv_device || world < 1 || rank < 0 || rank >= world || world > 64)
    return vh_fail(VH_E_INVALID, "vh_comm_init_custom: bad argument");
  if (!g_ctx.inited) return vh_fail(VH_E_INVALID, "vh_init has not been called");
  VH_ENTER();
  std::unique_ptr<vh_comm> c(new vh_comm());
  c->rank = rank; c->world = world; c->ops = *ops;
  if (int rc = comm_common_init(c.get())) { vh_comm_destroy(c.release()); return rc; }
  *out = c.release();
  return VH_OK;
}

extern "C" int vh_comm_info(vh_comm* c, vh_comm_info_t* out) {
  if (!c || !out) return vh_fail(VH_E_INVALID, "null argument");
  VH_ENTER();
  memset(out, 0, sizeof(*out));
  int dev = 0;
  if (c->nccl) {
    out->transport = VH_COMM_RCCL;
    int n = 0, r = 0;
    NCCL_TRY(g_rccl.CommCount(c->nccl, &n));
    NCCL_TRY(g_rccl.CommUserRank(c->nccl, &r));
    NCCL_TRY(g_rccl.CommCuDevice(c->nccl, &dev));
    out->nranks = n; out->rank = r;
  } else {
    out->transport = VH_COMM_CALLBACKS;
    out->nranks = c->world; out->rank = c->rank;
    HIP_TRY(hipGetDevice(&dev));
  }
  out->device = dev;
  HIP_TRY(hipDeviceGetPCIBusId(out->pci_bus_id, (int)sizeof(out->pci_bus_id), dev));
  return VH_OK;
}

// ------------------------------------------------------------------ plan agreement
// Pure host arithmetic, also reachable from tests without a device (vh_plan_agree below is not part of the C ABI).
static void merge_summaries(const VhSummary* all, int world, int ngroups, VhAgreed* ag, VhReplan* rp, bool* fatal) {
  for (int i = 0; i < VH_MAX_GROUP; ++i) { ag->klo[i] = ~0ull; ag->khi[i] = 0; }
  uint64_t rows = 0, passed = 0, sampled = 0, rows_max = 0;
  *fatal = false;
  for (int r = 0; r < world; ++r) {
    const VhSummary& s = all[r];
    for (int i = 0; i < ngroups; ++i) { ag->klo[i] = std::min(ag->klo[i], s.klo[i]); ag->khi[i] = std::max(ag->khi[i], s.khi[i]); }
    rows += s.rows_to_scan; passed += s.probe_passed; sampled += s.probe_sampled;
    rows_max = std::max(rows_max, s.rows_to_scan);
    rp->cap_override = std::max(rp->cap_override, s.cap_override);
    rp->part_override = std::max(rp->part_override, s.part_override);
    rp->force_hash |= s.force_hash != 0; rp->no_part |= s.no_part != 0;
    *fatal |= s.fatal != 0;
  }
  ag->rows_to_scan = rows; ag->rows_max = rows_max;
  ag->sel = sampled ? (double)passed / (double)sampled : 0.0;
}

// verdict words (all-reduced with SUM): [0] range error, [1] hash table full, [2] tuple extents exhausted, [3] fatal,
// [4] scanned_recs, [5] scanned_segments, [6] passed rows, [7] table organisation, [8] its square, [9] groups (hash path),
// [10] this rank's table changed since the cached agreement was made, [11] / [12] hashed partitioning full / ids too wide,
// [13] no stand-in arrays for a void attempt on this rank (see vh_comm::d_void)
__global__ void sharded_flags_kernel(const unsigned long long* counters, unsigned long long* flags, unsigned long long host_err,
                                     unsigned long long fatal, unsigned long long scanned_recs, unsigned long long scanned_segments,
                                     unsigned long long mode, unsigned long long ngroups, unsigned long long changed, unsigned long long novoid = 0) {
  if (threadIdx.x != 0) return;
  const unsigned long long err = (counters ? counters[2] : 0ull) | host_err;
  flags[0] = (err & VH_ERR_RANGE) ? 1 : 0;
  flags[1] = (err & VH_ERR_HASH_FULL) ? 1 : 0;
  flags[2] = (err & VH_ERR_PART_FULL) ? 1 : 0;
  flags[3] = fatal;
  flags[4] = scanned_recs; flags[5] = scanned_segments;
  flags[6] = counters ? counters[0] : 0ull;
  flags[7] = mode; flags[8] = mode * mode;
  flags[9] = ngroups;
  flags[10] = changed;
  flags[11] = (err & VH_ERR_HPART_FULL) ? 1 : 0;
  flags[12] = (err & VH_ERR_HP_WIDE) ? 1 : 0;
  flags[13] = novoid;                              // this rank could not allocate the stand-in arrays of a void attempt: nobody records the dense shape
  for (int i = 14; i < VH_FLAG_WORDS; ++i) flags[i] = 0;
}

// vh_query_agg with the rows kept in device memory (they are exchanged or gathered next). Same re-plan loop.
static int query_agg_device_rows(vh_table* t, const vh_plan* plan, vh_result** out) {
  VhExec* x = nullptr;
  if (int rc = exec_acquire(t, &x)) return rc;
  VhReplan rp;
  int rc = VH_OK;
  for (uint32_t attempt = 0; attempt < 12; ++attempt) {
    vh_result* r = nullptr;
    { std::lock_guard<std::mutex> lk(t->mu); rc = query_launch_locked(t, x, plan, &r, rp.cap_override, rp.force_hash, rp.part_override, rp.no_part, false, nullptr, nullptr, true, rp.hp_passes, rp.no_hpart); }
    if (rc) break;
    r->exec = x;
    int retry = 0;
    rc = result_finalize(r, &retry);
    if (rc) { r->exec = nullptr; delete r; break; }
    if (!retry) { r->info.retries = attempt; *out = r; return VH_OK; }
    replan_after(t, r, retry, &rp);
    r->exec = nullptr;
    delete r;
    rc = vh_fail(VH_E_NOMEM, "merge table kept overflowing");
  }
  (void)hipStreamSynchronize(x->stream());
  exec_release(t, x);
  return rc;
}

static int merge_kind_of(int kind) {   // aggregation that merges two partial states ("count" -> "long_sum" in the reference: agg_runner.cc:66-76)
  return kind == VH_METRIC_MAX ? VH_METRIC_MAX : kind == VH_METRIC_MIN ? VH_METRIC_MIN : VH_METRIC_SUM;
}

// A bitset column of a segment of the merge table: one id per row (ids already in HBM), or no ids at all.
static int merge_table_bitset(vh_table* tt, uint32_t seg, int col, uint64_t nrows, const void* d_ids, hipStream_t st) {
  VhColumn& c = tt->cols[col];
  const size_t vsz = c.elem == VH_BITSET32 ? 4 : 8;
  HIP_TRY(hipMalloc((void**)&c.bs_offsets[seg], (nrows + 1) * sizeof(uint64_t)));
  HIP_TRY(hipMalloc((void**)&c.bs_values[seg], std::max<size_t>((d_ids ? nrows : 0) * vsz, 8)));
  if (d_ids) {
    hipLaunchKernelGGL(iota_kernel, dim3((unsigned)std::min<uint64_t>((nrows + 256) / 256, 65535)), dim3(256), 0, st, c.bs_offsets[seg], nrows + 1);
    HIP_TRY(hipGetLastError());
    if (nrows) HIP_TRY(hipMemcpyAsync(c.bs_values[seg], d_ids, nrows * vsz, hipMemcpyDeviceToDevice, st));
  } else {
    HIP_TRY(hipMemsetAsync(c.bs_offsets[seg], 0, (nrows + 1) * sizeof(uint64_t), st));
  }
  c.bs_nvalues[seg] = d_ids ? nrows : 0;
  return VH_OK;
}

// Collective status point. Between two collectives of one sharded query a rank may fail on its own (an allocation, a kernel
// launch, a full pool): returning there would leave its peers blocked in the next collective, holding their communicator's
// lock. Instead every rank carries its local status to the next status point, where all of them learn of it and all of them
// return — the failing rank its own error, the others "failed on rank p".
static int agree_status(vh_comm* comm, int lrc, const char* what) {
  char own[sizeof(g_err)];
  snprintf(own, sizeof(own), "%s", g_err);
  const int32_t mine = lrc;
  std::vector<int32_t> all((size_t)comm->world, 0);
  if (int rc = comm->ops.allgather_host(comm->ops.ctx, &mine, all.data(), sizeof(int32_t)))
    return rc < 0 ? rc : vh_fail(VH_E_DEVICE, "status exchange before %s failed (%d)", what, rc);
  if (lrc) return vh_fail(lrc, "%s", own);
  for (int p = 0; p < comm->world; ++p)
    if (all[p]) return vh_fail(VH_E_DEVICE, "%s: failed on rank %d (status %d)", what, p, (int)all[p]);
  return VH_OK;
}

// 4b of the header comment: key-partitioned exchange of the partial groups (and of the distinct (group, id) pairs of every
// bitset metric), merge by re-aggregation on the owner, gather on root. `r` is this rank's finalised partial result.
static int sharded_exchange(vh_table* t, const vh_plan* plan, vh_comm* comm, int root, vh_result* r, const unsigned long long* gflags,
                            vh_result** out) {
  const int W = comm->world, R = comm->rank;
  const VhPlanDev& P = r->plan;
  const int nk = P.ngroup, nm = (int)r->user_metric.size();
  const bool has_hidden = r->info.has_hidden_count != 0;
  hipStream_t st = r->stream_for_work();
  std::vector<int> bitset_js;
  for (int j = 0; j < nm; ++j) if (P.m[r->user_metric[j]].sop() == SOP_BITSET) bitset_js.push_back(j);
  const int nbs = (int)bitset_js.size();
  const int ncols = nk + nm + (has_hidden ? 1 : 0);

  // Local failures between collectives are carried to the next status point (agree_status / the status word of the counts
  // all-gather), never returned on the spot: see agree_status.
  int lrc = VH_OK;
  char lerr[sizeof(g_err)] = "";
  auto keep = [&](int rc) { if (rc && !lrc) { lrc = rc; snprintf(lerr, sizeof(lerr), "%s", g_err); } return rc; };
  auto own_error = [&]() { return vh_fail(lrc, "%s", lerr); };
  // ---- 1. regroup by owner, in HBM
  std::vector<uint64_t> goffs(W + 1, 0);
  std::vector<vh_device_buffer> gbufs(ncols);
  int32_t nb = 0;
  keep(vh_result_partition(r, (uint32_t)W, goffs.data(), gbufs.data(), ncols, &nb));
  std::vector<std::vector<uint64_t>> poffs(nbs, std::vector<uint64_t>(W + 1, 0));
  std::vector<std::vector<vh_device_buffer>> pbufs(nbs, std::vector<vh_device_buffer>(nk + 1));
  for (int s = 0; s < nbs && !lrc; ++s) {
    int32_t n = 0;
    keep(vh_result_partition_pairs(r, bitset_js[s], (uint32_t)W, poffs[s].data(), pbufs[s].data(), nk + 1, &n));
  }
  // ---- 2. who sends how much to whom (+ one status word per rank)
  const int sets = 1 + nbs;
  const size_t cw = (size_t)sets * W + 1;
  std::vector<uint64_t> mine(cw, 0), all((size_t)W * cw);
  for (int p = 0; p < W && !lrc; ++p) {
    mine[p] = goffs[p + 1] - goffs[p];
    for (int s = 0; s < nbs; ++s) mine[(size_t)(1 + s) * W + p] = poffs[s][p + 1] - poffs[s][p];
  }
  mine[cw - 1] = lrc ? 1 : 0;
  if (int rc = comm->ops.allgather_host(comm->ops.ctx, mine.data(), all.data(), cw * sizeof(uint64_t))) return rc < 0 ? rc : vh_fail(VH_E_DEVICE, "all-gather of the exchange counts failed (%d)", rc);
  if (lrc) return own_error();
  for (int p = 0; p < W; ++p) if (all[(size_t)p * cw + cw - 1]) return vh_fail(VH_E_DEVICE, "regrouping the partial groups failed on rank %d", p);
  std::vector<std::vector<uint64_t>> roff(sets, std::vector<uint64_t>(W + 1, 0));
  uint64_t maxrows = 1;
  for (int s = 0; s < sets; ++s) {
    for (int p = 0; p < W; ++p) roff[s][p + 1] = roff[s][p] + all[(size_t)p * cw + (size_t)s * W + R];
    maxrows = std::max(maxrows, roff[s][W]);
  }
  if (maxrows > 0xFFFF0000ull) keep(vh_fail(VH_E_UNSUPPORTED, "a rank would own %llu partial rows", (unsigned long long)maxrows));

  // ---- 3. the merge table: [key columns, the plan's metrics (value metrics as their merge aggregation, bitset metrics as
  // bitsets), hidden count]; segment 0 = the group rows received, segment 1 + s = the pairs of bitset metric s
  std::vector<vh_col_desc> cd(ncols);
  for (int i = 0; i < nk; ++i) cd[i] = vh_col_desc{VH_DIM_NUMERIC, (int32_t)P.g[i].type()};
  for (int j = 0; j < nm; ++j) {
    const int col = plan->metrics[j];
    if (col == VH_COL_ROWID) { keep(vh_fail(VH_E_UNSUPPORTED, "search (VH_COL_ROWID) over a sharded table: storage positions are per rank")); cd[nk + j] = vh_col_desc{VH_METRIC_SUM, VH_U64}; continue; }
    const VhColumn& c = t->cols[col];
    cd[nk + j] = c.kind == VH_METRIC_BITSET ? vh_col_desc{VH_METRIC_BITSET, c.elem} : vh_col_desc{merge_kind_of(c.kind), c.elem};
  }
  if (has_hidden) cd[nk + nm] = vh_col_desc{VH_METRIC_SUM, VH_U64};
  vh_table* tt = nullptr;
  if (!lrc) keep(vh_table_create(cd.data(), ncols, maxrows, (uint32_t)sets, &tt));
  struct TableGuard { vh_table* t; ~TableGuard() { if (t) vh_table_destroy(t); } } guard{tt};
  std::vector<char*> d_ids(nbs, nullptr);
  struct IdsGuard { std::vector<char*>& v; ~IdsGuard() { for (char* p : v) if (p) (void)hipFree(p); } } ids_guard{d_ids};
  for (int s = 0; s < nbs && !lrc; ++s) {
    const size_t idsz = tt->cols[nk + bitset_js[s]].elem == VH_BITSET32 ? 4 : 8;
    if (hipMalloc((void**)&d_ids[s], std::max<size_t>(roff[1 + s][W] * idsz, 8)) != hipSuccess) keep(vh_fail(VH_E_NOMEM, "no memory for %llu received ids", (unsigned long long)roff[1 + s][W]));
  }
  if (int rc = agree_status(comm, lrc, "building the merge table")) return rc;      // every receive buffer of the exchange exists on every rank

  // ---- 4. exchange, straight into the merge table's column arenas
  {
    std::vector<const void*> send; std::vector<void*> recv; std::vector<uint32_t> es;
    for (int c = 0; c < ncols; ++c) {
      if (gbufs[c].reduce == -2) continue;              // a bitset metric's cardinalities do not merge: its pairs travel below
      send.push_back(gbufs[c].ptr); recv.push_back(tt->cols[c].base); es.push_back((uint32_t)tt->cols[c].esize);
    }
    if (int rc = comm->ops.alltoallv_device(comm->ops.ctx, (int32_t)send.size(), send.data(), recv.data(), es.data(), goffs.data(), roff[0].data(), st))
      return rc < 0 ? rc : vh_fail(VH_E_DEVICE, "exchange of the partial groups failed (%d)", rc);
  }
  for (int s = 0; s < nbs; ++s) {
    const int bcol = nk + bitset_js[s];
    const size_t idsz = tt->cols[bcol].elem == VH_BITSET32 ? 4 : 8;
    std::vector<const void*> send; std::vector<void*> recv; std::vector<uint32_t> es;
    for (int i = 0; i < nk; ++i) { send.push_back(pbufs[s][i].ptr); recv.push_back(tt->cols[i].base + (size_t)(1 + s) * tt->cols[i].stride); es.push_back((uint32_t)tt->cols[i].esize); }
    send.push_back(pbufs[s][nk].ptr); recv.push_back(d_ids[s]); es.push_back((uint32_t)idsz);
    if (int rc = comm->ops.alltoallv_device(comm->ops.ctx, (int32_t)send.size(), send.data(), recv.data(), es.data(), poffs[s].data(), roff[1 + s].data(), st))
      return rc < 0 ? rc : vh_fail(VH_E_DEVICE, "exchange of the distinct pairs failed (%d)", rc);
  }
  // ---- 5. what was not received: identities (a pair row leaves every value state unchanged), empty sets (a group row adds no id)
  std::vector<vh_group_col> mg(nk);
  std::vector<int32_t> mm;
  vh_result* rm = nullptr;
  auto merge_locally = [&]() -> int {
    for (int s = 0; s < sets; ++s) {
      const uint64_t n = roff[s][W];
      for (int c = nk; c < ncols; ++c) {
        VhColumn& col = tt->cols[c];
        if (is_bitset_elem(col.elem)) {
          const bool own = s > 0 && c == nk + bitset_js[s - 1];
          if (int rc = merge_table_bitset(tt, (uint32_t)s, c, n, own ? d_ids[s - 1] : nullptr, st)) return rc;
          continue;
        }
        if (s == 0 || !n) continue;
        int sop; uint64_t ident;
        if (sop_for(col.kind, col.elem, &sop, &ident)) return vh_fail(VH_E_DEVICE, "merge table: no identity for column %d", c);
        char* dst = col.base + (size_t)s * col.stride;
        const unsigned grid = (unsigned)std::min<uint64_t>(2048, (n + 255) / 256);
        VH_ELEM_SWITCH(col.elem, (fill_kernel<T><<<dim3(grid), dim3(256), 0, st>>>(reinterpret_cast<T*>(dst), n, vh_lit_host<T>(ident))));
      }
      tt->seg_rows[s] = n;
      table_note_change(tt, (uint32_t)s, 0, n);
    }
    HIP_TRY(hipGetLastError());
    tt->nseg = (uint32_t)sets;
    HIP_TRY(hipStreamSynchronize(st));
    if (int rc = refresh_stats(tt, 0, (uint32_t)sets)) return rc;
    // ---- 6. merge by re-aggregation; HAVING and top-N see merged groups
    for (int i = 0; i < nk; ++i) { memset(&mg[i], 0, sizeof(vh_group_col)); mg[i].col = i; mg[i].granularity = VH_T_NONE; }
    for (int j = 0; j < nm + (has_hidden ? 1 : 0); ++j) mm.push_back(nk + j);
    vh_plan mp{};
    mp.groups = mg.data(); mp.ngroups = nk; mp.metrics = mm.data(); mp.nmetrics = (int32_t)mm.size();
    mp.lits = plan->lits; mp.nlits = plan->nlits; mp.having = plan->having; mp.nhaving = plan->nhaving;
    mp.top_col = plan->top_col; mp.top_desc = plan->top_desc; mp.top_k = plan->top_k;
    mp.groups_hint = roff[0][W];
    return query_agg_device_rows(tt, &mp, &rm);
  };
  keep(merge_locally());
  std::unique_ptr<vh_result> rm_holder(rm);
  if (rm) {
    if (has_hidden) { rm->user_metric.resize(nm); rm->info.has_hidden_count = 1; rm->info.nmetrics = nm; }
    rm->info.scanned_recs = gflags[4]; rm->info.scanned_segments = gflags[5]; rm->info.passed_recs = gflags[6];
    rm->info.path = VH_PATH_HASH;
    rm->info.scan_kernel_ms = r->info.scan_kernel_ms; rm->info.algorithmic_bytes = r->info.algorithmic_bytes; rm->info.retries = r->info.retries;
    rm->kernel = r->kernel;
  }

  // ---- 7. how many groups everywhere; leave the rows with their owners or gather them on root
  uint64_t cnt[3] = {rm ? rm->ngroups_host : 0, rm ? rm->info.ngroups : 0, lrc ? 1ull : 0ull};      // (+ this rank's status)
  std::vector<uint64_t> cnts((size_t)W * 3);
  if (int rc = comm->ops.allgather_host(comm->ops.ctx, cnt, cnts.data(), sizeof(cnt))) return rc < 0 ? rc : vh_fail(VH_E_DEVICE, "all-gather of the group counts failed (%d)", rc);
  if (lrc) return own_error();
  uint64_t total_rows = 0, total_groups = 0;
  for (int p = 0; p < W; ++p) {
    if (cnts[(size_t)p * 3 + 2]) return vh_fail(VH_E_DEVICE, "merging the exchanged groups failed on rank %d", p);
    total_rows += cnts[(size_t)p * 3]; total_groups += cnts[(size_t)p * 3 + 1];
  }
  rm->info.ngroups = total_groups;
  if (root < 0) {
    rm->owned_table = tt; guard.t = nullptr;
    *out = rm_holder.release();
    return VH_OK;
  }
  // gather: every owner's rows to root, column by column in one grouped exchange
  std::unique_ptr<vh_result> rf(new vh_result());
  rf->table = t;
  rf->info = rm->info;
  rf->info.returned_groups = R == root ? total_rows : 0;
  rf->plan.ngroup = nk; rf->plan.nmetric = rm->plan.nmetric; rf->plan.key_words = rm->plan.key_words;
  for (int i = 0; i < nk; ++i) rf->plan.g[i] = rm->plan.g[i];
  rf->user_metric = rm->user_metric; rf->metric_elem = rm->metric_elem; rf->group_elem = rm->group_elem;
  rf->mode = VH_MODE_HASH;
  const int ndev = rm->plan.nmetric;
  std::vector<const void*> send; std::vector<void*> recv; std::vector<uint32_t> es;
  size_t bytes = 0;
  const uint64_t cap_rows = R == root ? std::max<uint64_t>(total_rows, 1) : 1;
  for (int i = 0; i < nk; ++i) { rf->off_key[i] = bytes; bytes += (cap_rows * vh_elem_size(rm->plan.g[i].type()) + 255) / 256 * 256; }
  for (int u = 0; u < ndev; ++u) { rf->off_state[u] = bytes; bytes += (cap_rows * vh_elem_size(rm->metric_elem[u]) + 255) / 256 * 256; }
  if (hipMalloc((void**)&rf->d_own, bytes) != hipSuccess || host_alloc_near_device((void**)&rf->h_own, bytes, hipHostMallocDefault) != hipSuccess)
    keep(vh_fail(VH_E_NOMEM, "no memory for %llu gathered groups", (unsigned long long)total_rows));
  if (int rc = agree_status(comm, lrc, "gathering the merged groups")) return rc;
  for (int i = 0; i < nk; ++i) {
    send.push_back(rm->topk_active ? rm->d_out_key2[i] : rm->d_out_key[i]); recv.push_back(rf->d_own + rf->off_key[i]); es.push_back((uint32_t)vh_elem_size(rm->plan.g[i].type()));
  }
  for (int u = 0; u < ndev; ++u) {
    send.push_back(rm->topk_active ? rm->d_out_state2[u] : rm->d_out_state[u]); recv.push_back(rf->d_own + rf->off_state[u]); es.push_back((uint32_t)vh_elem_size(rm->metric_elem[u]));
  }
  std::vector<uint64_t> soff(W + 1, 0), goff(W + 1, 0);
  for (int p = 0; p <= W; ++p) soff[p] = p > root ? rm->ngroups_host : 0;          // everything goes to root
  if (R == root) for (int p = 0; p < W; ++p) goff[p + 1] = goff[p] + cnts[(size_t)p * 3];
  hipStream_t st2 = rm->stream_for_work();
  if (int rc = comm->ops.alltoallv_device(comm->ops.ctx, (int32_t)send.size(), send.data(), recv.data(), es.data(), soff.data(), goff.data(), st2))
    return rc < 0 ? rc : vh_fail(VH_E_DEVICE, "gather of the merged groups failed (%d)", rc);
  if (R == root && total_rows) HIP_TRY(hipMemcpyAsync(rf->h_own, rf->d_own, bytes, hipMemcpyDeviceToHost, st2));
  HIP_TRY(hipStreamSynchronize(st2));
  rf->h_base = rf->h_own;
  rf->ngroups_host = R == root ? total_rows : 0;
  rf->finalized = true;
  *out = rf.release();
  return VH_OK;   // rm (and with it the merge table's context), then the merge table itself, go out of scope here
}

static bool knobs_no_fused_sharded() { return false; }   // (true: every dense sharded query in the two-collective form — what round 5 measured the fused step against)

// The steady state of a dense sharded query — its plan has an agreement in the cache and the shape of its partial table is known on every
// rank — as ONE stream-ordered sequence behind the scan: verdict words -> ONE collective group (the verdict's all-reduce and the reduce of
// every state array: one launch, one ring set-up) -> on root the emission into pinned host memory -> ONE host wait. The verdict is read
// AFTER everything ran: when any rank overflowed a pool, met a digit outside the agreed range or saw its table change, the reduced arrays
// are garbage of the agreed size — nobody looks at them, all ranks re-plan together exactly as they would have after the separate
// verdict. A rank that could not even plan joins the group with scratch arrays of the recorded shape (the collectives always match).
// *again = 1: this attempt is void, the caller's loop goes round. Replaces two collectives and two host round trips per query by one
// and one (what the reference's controller does per query: src/cluster/query/agg_runner.cc:83-140).
static int fused_dense_step(vh_table* t, vh_comm* comm, int root, VhExec* x, vh_result* r, int lrc, bool changed, const vh_comm::Agreement& agr,
                            const std::string& sig, char* local_err, VhReplan* rp, int* again) {
  hipStream_t st = x->stream();
  const int W = comm->world, R = comm->rank;
  *again = 0;
  const bool mine_ok = r && !lrc && r->mode == agr.dense_mode && r->plan.nbitset == 0;
  vh_device_buffer bufs[VH_MAX_METRIC + 1];
  int32_t nb = 0;
  bool shape_ok = mine_ok;
  if (mine_ok) {
    if (int rc = vh_result_device_buffers(r, bufs, VH_MAX_METRIC + 1, &nb)) { shape_ok = false; if (!local_err[0]) snprintf(local_err, sizeof(g_err), "%s", g_err); (void)rc; }
    if (shape_ok && (size_t)nb != agr.dense.size()) shape_ok = false;
    for (int32_t b = 0; shape_ok && b < nb; ++b)
      shape_ok = bufs[b].count == agr.dense[b].count && bufs[b].elem == agr.dense[b].elem && bufs[b].reduce == agr.dense[b].reduce;
  }
  int alloc_rc = VH_OK;
  if (!shape_ok) {
    // this rank has no partial of the agreed shape: arrays that only keep the collectives matched — slices of the communicator's stand-in
    // buffer, which was sized for this shape when the shape was recorded (dense_shape_bytes; no allocation can fail here)
    nb = (int32_t)agr.dense.size();
    size_t off = 0;
    for (int32_t b = 0; b < nb; ++b) {
      const size_t bytes = (std::max<uint64_t>(agr.dense[b].count, 1) * vh_elem_size(agr.dense[b].elem) + 255) / 256 * 256;
      bufs[b] = vh_device_buffer{comm->d_void + off, agr.dense[b].count, agr.dense[b].elem, agr.dense[b].reduce};
      off += bytes;
    }
    if (off > comm->d_void_bytes) {      // cannot happen (recorded shapes never outgrow the buffer): say so rather than write past it — before any collective is posted
      alloc_rc = vh_fail(VH_E_NOMEM, "sharded query: the stand-in arrays of a void attempt are smaller than the recorded shape");
      if (!local_err[0]) snprintf(local_err, sizeof(g_err), "%s", g_err);
      return alloc_rc;      // (before anything is posted; peers learn of it through the communicator's failure, as with any rank that dies)
    }
  }
  const bool fatal = (lrc && !(changed)) || alloc_rc;             // (a stale agreement that does not even plan is not an error: everyone re-agrees)
  // a rank whose partial does not have the recorded shape although it planned: its table changed under the agreement — say so
  const bool void_mine = !shape_ok;
  hipLaunchKernelGGL(sharded_flags_kernel, dim3(1), dim3(64), 0, st, mine_ok && shape_ok ? r->plan.counters : nullptr, comm->d_flags, 0ull,
                     (unsigned long long)(fatal ? 1 : 0), r ? r->info.scanned_recs : 0ull, r ? r->info.scanned_segments : 0ull,
                     (unsigned long long)agr.dense_mode, 0ull, (unsigned long long)((changed || void_mine) ? 1 : 0));
  HIP_TRY(hipGetLastError());
  // Between GroupStart and GroupEnd nothing returns: a rank that left the group open, or posted fewer collectives than its peers, would hang
  // all of them. Every rank posts the verdict and every recorded array; errors are collected and reported after the group is closed.
  const bool grouped = comm->nccl != nullptr;
  ncclResult_t gs = ncclSuccess, ge = ncclSuccess;
  if (grouped) gs = g_rccl.GroupStart();
  int red_rc = comm->ops.reduce_device(comm->ops.ctx, comm->d_flags, VH_FLAG_WORDS, VH_U64, VH_RED_SUM, -1, st);
  for (int32_t b = 0; b < nb; ++b) {
    if (!bufs[b].count) continue;      // (a zero-length array is skipped by every rank alike: counts are part of the recorded shape)
    const int rc1 = comm->ops.reduce_device(comm->ops.ctx, bufs[b].ptr, bufs[b].count, bufs[b].elem, bufs[b].reduce, root, st);
    if (rc1 && !red_rc) red_rc = rc1;
  }
  if (grouped) ge = g_rccl.GroupEnd();
  if (gs != ncclSuccess || ge != ncclSuccess) return vh_fail(VH_E_DEVICE, "ncclGroup%s failed: %s", gs != ncclSuccess ? "Start" : "End", g_rccl.GetErrorString(gs != ncclSuccess ? gs : ge));
  if (red_rc) return red_rc < 0 ? red_rc : vh_fail(VH_E_DEVICE, "reduce of a partial state array failed (%d)", red_rc);
  HIP_TRY(hipMemcpyAsync(comm->h_flags, comm->d_flags, VH_FLAG_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  int fin_rc = VH_OK, rt = 0;
  const bool emits = mine_ok && shape_ok && (root < 0 || R == root);
  if (emits) {
    r->device_rows = false;                                       // small results go straight into pinned host memory again
    fin_rc = result_finalize(r, &rt);                             // emission + header behind the reduce; its wait is the query's one host wait
    if (fin_rc && !local_err[0]) snprintf(local_err, sizeof(g_err), "%s", g_err);
  }
  if (!emits || fin_rc) HIP_TRY(hipStreamSynchronize(st));        // the reduce has read this rank's partial; the verdict is on the host
  const unsigned long long* f = comm->h_flags;
  if (f[10]) { comm->agreed.erase(sig); *again = 1; return VH_OK; }          // some rank's table changed: the agreement is void for all
  if (f[3]) { comm->agreed.clear(); return fatal ? vh_fail(lrc ? lrc : alloc_rc, "%s", local_err) : vh_fail(VH_E_DEVICE, "the query failed on another rank"); }
  const int verdict = f[12] ? 6 : f[11] ? 4 : f[1] ? 1 : f[2] ? 3 : f[0] ? 2 : 0;
  if (verdict) {
    // (a plan that overflows from the agreed sizes would do so on every query: its later queries take the two-step form, which re-plans
    // before anything is reduced, until one of them goes through at the first attempt again)
    auto hit = comm->agreed.find(sig);
    if (hit != comm->agreed.end()) hit->second.dense_known = false;
    r->info.passed_recs = f[6];
    replan_after(t, r, verdict, rp);
    *again = 1;
    return VH_OK;
  }
  if (fin_rc) { comm->agreed.clear(); return vh_fail(fin_rc, "%s", local_err); }     // (this rank's emission failed on its own: peers already hold their results)
  r->info.retries = 0;
  r->info.scanned_recs = f[4]; r->info.scanned_segments = f[5]; r->info.passed_recs = f[6];
  if (!emits) {
    r->h_base = reinterpret_cast<char*>(x->h_counters);           // no rows here: any readable address
    r->ngroups_host = 0; r->info.returned_groups = 0; r->info.ngroups = 0;
    r->finalized = true;
  }
  return VH_OK;
}

extern "C" int vh_query_agg_sharded(vh_table* t, const vh_plan* plan, vh_comm* comm, int32_t root, vh_result** out) {
  if (!t || !plan || !comm || !out) return vh_fail(VH_E_INVALID, "null argument");
  if (root >= comm->world || root < -1) return vh_fail(VH_E_INVALID, "root %d of %d ranks", root, comm->world);
  if (comm->world == 1 && !test_env("VH_TEST_SHARDED_WORLD1")) return vh_query_agg(t, plan, out);   // (the test knob sends one rank through the whole protocol: RCCL with a single GPU)
  VH_ENTER();
  std::lock_guard<std::mutex> comm_lk(comm->mu);
  VhExec* x = nullptr;
  if (int rc = exec_acquire(t, &x)) return rc;     // (a rank that fails here leaves its peers waiting in the first collective: nothing to agree on yet)
  struct ExecGuard { vh_table* t; VhExec* x; ~ExecGuard() { if (x) { (void)hipStreamSynchronize(x->stream()); exec_release(t, x); } } } xg{t, x};
  hipStream_t st = x->stream();
  const int W = comm->world, R = comm->rank;
  VhReplan rp;
  char local_err[sizeof(g_err)] = "";
  const std::string sig = plan_signature(plan);
  const uint64_t local_state = plan_local_state(plan, t);       // (the table cannot change under a sharded query's first attempt: syncs take comm-independent locks, but this rank's caller is here)
  for (uint32_t attempt = 0; attempt < 12; ++attempt) {
    // ---- 1. agree on what to plan with (or take the agreement this plan got last time)
    VhAgreed ag{};
    int lrc = VH_OK;
    bool cached = false, changed = false;
    if (attempt == 0) {
      auto hit = comm->agreed.find(sig);
      if (hit != comm->agreed.end()) { ag = hit->second.ag; cached = true; changed = hit->second.local_state != local_state; }
    }
    if (!cached) {
      VhSummary mine{};
      { std::lock_guard<std::mutex> lk(t->mu); vh_result* none = nullptr; lrc = query_launch_locked(t, x, plan, &none, 0, false, 0, false, false, &mine); }
      if (lrc && !local_err[0]) snprintf(local_err, sizeof(local_err), "%s", g_err);
      mine.cap_override = rp.cap_override; mine.part_override = rp.part_override;
      mine.force_hash = rp.force_hash; mine.no_part = rp.no_part; mine.fatal = lrc ? 1 : 0;
      std::vector<VhSummary> all(W);
      if (int rc = comm->ops.allgather_host(comm->ops.ctx, &mine, all.data(), sizeof(VhSummary)))
        return rc < 0 ? rc : vh_fail(VH_E_DEVICE, "all-gather of the plan summaries failed (%d)", rc);
      bool fatal = false;
      merge_summaries(all.data(), W, plan->ngroups, &ag, &rp, &fatal);
      if (fatal) { comm->agreed.clear(); return lrc ? vh_fail(lrc, "%s", local_err) : vh_fail(VH_E_INVALID, "another rank rejected the plan"); }
      if (comm->agreed.size() >= 64) comm->agreed.clear();
      comm->agreed[sig] = vh_comm::Agreement{ag, plan_local_state(plan, t)};
    }

    // ---- 2. scan this rank's shard with the agreed plan
    vh_result* r = nullptr;
    { std::lock_guard<std::mutex> lk(t->mu); lrc = query_launch_locked(t, x, plan, &r, rp.cap_override, rp.force_hash, rp.part_override, rp.no_part, false, nullptr, &ag, true, rp.hp_passes, rp.no_hpart); }
    if (lrc && cached && changed) { lrc = VH_OK; r = nullptr; }          // a stale agreement may not even plan: not an error, everyone re-agrees below
    if (lrc && !local_err[0]) snprintf(local_err, sizeof(local_err), "%s", g_err);
    std::unique_ptr<vh_result> holder(r);
    struct Detach { vh_result* r; ~Detach() { if (r) r->exec = nullptr; } } detach{r};   // unless handed out, the partial never owns the context (the guard does)
    if (r) r->exec = x;
    const bool sparse = r && (r->mode == VH_MODE_HASH || r->plan.nbitset > 0);   // partials exchanged by key, not reduced in place
    int retry = 0;
    if (r && sparse) {
      // partial groups: the HAVING / top-N of the query apply to MERGED groups (step 6 of sharded_exchange)
      r->nhaving = 0; r->topk = 0; r->topk_active = false;
      lrc = result_finalize(r, &retry);
      if (lrc && !local_err[0]) snprintf(local_err, sizeof(local_err), "%s", g_err);
    }
    // ---- 2a. steady state of a dense query: everything that follows as ONE stream-ordered sequence (see fused_dense_step)
    if (attempt == 0 && cached) {
      auto hit = comm->agreed.find(sig);
      if (hit != comm->agreed.end() && hit->second.dense_known) {
        const vh_comm::Agreement agr = hit->second;                // (a copy: the entry may be dropped inside)
        int again = 0;
        const int frc = fused_dense_step(t, comm, root, x, r, lrc, changed, agr, sig, local_err, &rp, &again);
        if (frc) return frc;
        if (again) continue;
        xg.x = nullptr; detach.r = nullptr;                       // the result owns the context from here
        *out = holder.release();
        return VH_OK;
      }
    }
    // ---- 3. verdict: error flags, row counters and the table organisation of every rank, all-reduced
    // the stand-in arrays a rank with a void attempt joins the fused group with (fused_dense_step) are made HERE, before the verdict, so that
    // a rank that cannot have them says so in the verdict all ranks read: the dense shape is recorded by all ranks or by none, without a
    // collective of its own, and never inside a collective group
    unsigned long long novoid = 0;
    if (attempt == 0 && r && !sparse && !lrc && !knobs_no_fused_sharded()) {
      vh_device_buffer vb[VH_MAX_METRIC + 1];
      int32_t vn = 0;
      size_t need = 0;
      if (vh_result_device_buffers(r, vb, VH_MAX_METRIC + 1, &vn) == VH_OK)
        for (int32_t b = 0; b < vn; ++b) need += (std::max<uint64_t>(vb[b].count, 1) * vh_elem_size(vb[b].elem) + 255) / 256 * 256;
      if (need > comm->d_void_bytes) {
        (void)hipStreamSynchronize(st);                           // (nothing enqueued may still use the old one)
        if (comm->d_void) { (void)hipFree(comm->d_void); comm->d_void = nullptr; comm->d_void_bytes = 0; }
        if (hipMalloc((void**)&comm->d_void, need) == hipSuccess) comm->d_void_bytes = need; else { (void)hipGetLastError(); novoid = 1; }
      }
    }
    const unsigned long long host_err = retry == 1 ? VH_ERR_HASH_FULL : retry == 2 ? VH_ERR_RANGE : retry == 3 ? VH_ERR_PART_FULL : retry == 4 ? VH_ERR_HPART_FULL : retry == 6 ? VH_ERR_HP_WIDE : 0ull;
    hipLaunchKernelGGL(sharded_flags_kernel, dim3(1), dim3(64), 0, st, r && !sparse ? r->plan.counters : nullptr, comm->d_flags, host_err,
                       (unsigned long long)(lrc ? 1 : 0), r ? r->info.scanned_recs : 0ull, r ? r->info.scanned_segments : 0ull,
                       (unsigned long long)(r ? (sparse ? 100 + (r->mode == VH_MODE_HASH ? 0 : r->mode) : r->mode) : 0), r && sparse ? r->info.ngroups : 0ull,
                       (unsigned long long)(changed ? 1 : 0), novoid);
    HIP_TRY(hipGetLastError());
    if (r && sparse) {   // the row counter of a finalised result is on the host already
      HIP_TRY(hipMemcpyAsync(comm->d_flags + 6, &r->info.passed_recs, sizeof(uint64_t), hipMemcpyHostToDevice, st));
    }
    if (int rc = comm->ops.reduce_device(comm->ops.ctx, comm->d_flags, VH_FLAG_WORDS, VH_U64, VH_RED_SUM, -1, st))
      return rc < 0 ? rc : vh_fail(VH_E_DEVICE, "all-reduce of the verdict failed (%d)", rc);
    HIP_TRY(hipMemcpyAsync(comm->h_flags, comm->d_flags, VH_FLAG_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const unsigned long long* f = comm->h_flags;
    if (f[10]) { comm->agreed.erase(sig); continue; }             // some rank's table changed: the agreement is void for all, this attempt is dropped
    if (f[3]) { comm->agreed.clear(); return lrc ? vh_fail(lrc, "%s", local_err) : vh_fail(VH_E_DEVICE, "the query failed on another rank"); }
    const unsigned long long my_mode = r ? (sparse ? 100 + (r->mode == VH_MODE_HASH ? 0 : r->mode) : r->mode) : 0;
    if (f[7] != (unsigned long long)W * my_mode || f[8] != (unsigned long long)W * my_mode * my_mode)
      return vh_fail(VH_E_DEVICE, "ranks planned different table organisations for one query (this rank: %llu): plans or table shapes differ between ranks", my_mode);
    const int verdict = f[12] ? 6 : f[11] ? 4 : f[1] ? 1 : f[2] ? 3 : f[0] ? 2 : 0;      // same priority as a single-GPU query's re-plan
    if (verdict) {
      if (r) r->info.passed_recs = f[6];                          // all ranks' survivors: an upper bound for the tuple pools of any one of them
      replan_after(t, r, verdict, &rp);                           // every rank re-plans; the requests are merged in step 1 of the next attempt
      continue;
    }
    r->info.retries = attempt;
    if (sparse) {
      vh_result* merged = nullptr;
      int rc = sharded_exchange(t, plan, comm, root, r, f, &merged);   // the partial is dropped afterwards, its context returns to the pool through the guard
      if (rc) return rc;
      { std::lock_guard<std::mutex> lk(t->mu); t->groups_seen[r->group_sig] = r->info.ngroups; }
      *out = merged;
      return VH_OK;
    }
    // ---- 4a. dense: identically indexed partial tables, reduced in place
    vh_device_buffer bufs[VH_MAX_METRIC + 1];
    int32_t nb = 0;
    if (int rc = vh_result_device_buffers(r, bufs, VH_MAX_METRIC + 1, &nb)) return rc;
    // (RCCL: the state arrays of one query go out as ONE group — one launch, one ring set-up — instead of one collective per array)
    const bool grouped = comm->nccl != nullptr && nb > 1;
    if (grouped) NCCL_TRY(g_rccl.GroupStart());
    int red_rc = VH_OK;
    for (int32_t b = 0; b < nb && !red_rc; ++b)
      red_rc = comm->ops.reduce_device(comm->ops.ctx, bufs[b].ptr, bufs[b].count, bufs[b].elem, bufs[b].reduce, root, st);
    if (grouped) NCCL_TRY(g_rccl.GroupEnd());
    if (red_rc) return red_rc < 0 ? red_rc : vh_fail(VH_E_DEVICE, "reduce of a partial state array failed (%d)", red_rc);
    if (attempt == 0 && !knobs_no_fused_sharded()) {               // every rank is here together: all of them record, or none
      auto hit = comm->agreed.find(sig);
      const bool everyone = f[13] == 0;                            // every rank holds stand-in arrays of this shape (made before the verdict)
      if (hit != comm->agreed.end() && everyone) {
        hit->second.dense.clear();
        for (int32_t b = 0; b < nb; ++b) hit->second.dense.push_back(vh_comm::BufShape{bufs[b].count, bufs[b].elem, bufs[b].reduce});
        hit->second.dense_mode = r->mode; hit->second.dense_known = true;
      }
    }
    r->info.scanned_recs = f[4]; r->info.scanned_segments = f[5];
    if (root < 0 || R == root) {
      r->device_rows = false;                                     // small results go straight into pinned host memory again
      int rt = 0;
      if (int rc = result_finalize(r, &rt)) return rc;
      r->info.scanned_recs = f[4]; r->info.scanned_segments = f[5]; r->info.passed_recs = f[6];
    } else {
      HIP_TRY(hipStreamSynchronize(st));                           // the reduce has read this rank's partial
      r->info.passed_recs = f[6];
      r->h_base = reinterpret_cast<char*>(x->h_counters);          // no rows here: any readable address
      r->ngroups_host = 0; r->info.returned_groups = 0; r->info.ngroups = 0;
      r->finalized = true;
    }
    xg.x = nullptr; detach.r = nullptr;                           // the result owns the context from here
    *out = holder.release();
    return VH_OK;
  }
  return vh_fail(VH_E_NOMEM, "aggregate table kept overflowing on some rank");
}

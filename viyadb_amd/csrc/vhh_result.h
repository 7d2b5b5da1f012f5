// vhh_result.h — host side of libviya_hip, part of viya_hip.hip's translation unit (included there, in order; not a stand-alone header):
// typed host helpers, vh_segment_stats, struct vh_result and its views.
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
  std::lock_guard<std::mutex> lk(t->mu);
  if (int src = sync_resolve(t)) return src;
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
  uint32_t part_blocks = 0;         // DENSE_PART: blocks of the phase-2 launch (what vh_part_shares shares out)
  bool hp_direct = false;           // ... or its aggregation kernel already wrote the output columns (no emission kernel to run)
  int hp_chunks = 0;                // ... in this many chunk launches, each with a region of `hp_chunk_rows` rows of the output columns: delivered chunk by chunk
  uint64_t hp_chunk_rows = 0;
  bool hp_one_launch = false;       // ... regions only: ONE launch of the aggregation, the regions' rows packed behind it (no overlap of copies and kernels)
  VhHpArgs hp_args;                 // ... and the pool descriptors its kernels were given
  VhHpArgs* d_hp_args = nullptr;    // ... where they lie on the device; hp_scan_blocks: the scan blocks that wrote level A (a second pass over heavy partitions reads pool a: vhh_finalize.h)
  int hp_scan_blocks = 0;
  vh_table* table = nullptr;
  uint64_t launch_epoch = 0;        // the table's sync_epoch when the query was planned (a second pass over the table's rows must see the same rows)
  vh_result_info info{};
  int mode = 0;
  bool finalized = false;
  bool stream_quiet = false;   // vh_query_agg waited for the last event it recorded on the context's stream and nothing was enqueued since: the destructor need not wait again (~10 us per query)
  // the ONE way work gets onto a finished result's stream (exchange, partitioning, device buffers ...): whoever enqueues behind the query
  // takes the stream from here, which is also what makes the destructor wait for it before the context goes to the next query
  hipStream_t stream_for_work() { stream_quiet = false; return exec->stream(); }
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
  bool small_tail = false;             // a small dense result that goes straight into pinned host memory: merge + emission + header are ONE launch (small_tail_kernel)
  bool unmerged = false;               // ... which also adds the private copies up: until then (or until merge_copies_now, in front of a cross-GPU reduce) nobody has
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
    if (exec) { if (!stream_quiet) (void)hipStreamSynchronize(exec->stream()); exec_release(table, exec); }   // nothing of this query may still run on a context the next one takes
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


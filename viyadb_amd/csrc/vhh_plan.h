// vhh_plan.h — host side of libviya_hip, part of viya_hip.hip's translation unit (included there, in order; not a stand-alone header):
// the planner: QueryBuild — filter program, segment snapshot + skipping, group / metric shapes, table organisation, hashed partitioning, projection choice.
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
  // (called once per segment and query — a thousand times for C3 —: the evaluation stack lives on the caller's stack, not on the heap: 17 of the 36 us
  // a C3 query spent planning were these allocations)
  char small[256];
  std::vector<char> big;
  char* st = small;
  if ((size_t)p->nfilter + 1 > sizeof(small)) { big.resize((size_t)p->nfilter + 1); st = big.data(); }
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
  if (int frc = derived_fence(t, st)) return frc;
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

// Heavy ranges of the hashed partitioning (VhPlanDev::heavy_mark / heavy_only), per calling thread: vh_query_agg lets its plans mark heavy ranges
// (it can run the second pass: it holds the caller's plan when the verdict is in); the second pass itself plans with `only` = the first pass's bitmap
// and sizes its (group, id) set for the ids those ranges can hold. Everything else (split launch / finalize, sharded queries) plans without either.
// `tuples_of`: every marked range belongs to a level-A partition that hp_plan_kernel left out whole — the second pass then reads that partition's
// tuples out of the first pass's pool a (hp_heavy_tuples_kernel) instead of scanning the table for its rows.
struct VhHeavyCtx { bool allow_mark = false; const uint32_t* only = nullptr; uint64_t ids_bound = 0; const vh_result* tuples_of = nullptr; uint64_t epoch = 0; };
static thread_local VhHeavyCtx g_heavy;

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
  int slot_bits[VH_MAX_SLOTS];                   // bit-field records (VhPack::bits): the record's bytes (4 / 8), slot_recoff = the field's bit offset, slot_stored = its bits; 0 otherwise
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
  bool heavy_marks = false; size_t o_heavy = 0;         // hashed partitioning: heavy ranges are marked for a second pass (layout_scratch)
  bool part_balanced = false; size_t o_pcount = 0;      // phase 2's blocks by the partitions' tuple counts (layout_scratch)
  uint64_t capacity = 0;
  bool hpart = false;
  uint64_t hp_tuple_cap = 0;
  int hp_units = 1;                 // 16-byte units per tuple of the hashed partitioning: 2 when the tuples carry the ids of a bitset metric
  int hp_ring_nb = 1;
  bool part_ring = false;           // DENSE_PART: phase 1 writes its tuples through the block's ring writer (vj_part_ring_add)
  bool split_ring = false;          // DENSE_PART, two levels: the second split through the ring writer, extents by position (part_split_ring_kernel)
  bool hp_fan = false;              // ... whose scan writes the level-A pool itself (vj_fan_add): no stream pool, no level-A scatter
  bool hp_off32 = false;            // ... and reads a row's CSR offsets from the 32-bit copies (every scanned segment has one: VhColumn::bs_offsets32)
  bool hp_pack = false;             // ... or 1 all the same: payload, two ids and their count packed into the tuple's second word (VhHpArgs::pk)
  int hp_pbits = 0, hp_idbits = 0;
  int hp_bpp = 1;
  uint32_t hp_chunk = 256;
  bool packed = false, packed_compressed = false;
  int qpay = 0, qpay_slot = -1;                  // streamed payload (VhJitShape::qpay): the record's bytes, the slot the records come from
  std::vector<int> pp_cols;                      // the filter's columns (ascending) when every leaf reads a fixed-width column: what a predicate projection must hold
  int pp_boff[VJ_MAX_PRED] = {}, pp_bbits[VJ_MAX_PRED] = {}, pp_soff[VJ_MAX_PRED] = {}, pp_sbits[VJ_MAX_PRED] = {};      // predicate column k's bit field in the byte-plane / bit-sliced projection noted in jshape
  void predpack_auto(bool want_sliced);          // counts this query towards an unasked predicate projection of the form it would have used
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
  bool jit_pagg() const { return jk && jk->fn_pagg && mode == VH_MODE_DENSE_PART && !knobs().no_jit_pagg; }      // phase 2 of DENSE_PART runs as the plan's compiled `_pagg` kernel
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

  r->table = t; r->launch_epoch = t->sync_epoch;
  memset(&P, 0, sizeof(P));

  // ---------------- column slots
  for (int i = 0; i < 256; ++i) slot_of[i] = -1;
  for (int i = 0; i < VH_MAX_SLOTS; ++i) { slot_col[i] = -1; slot_rec[i] = -1; slot_recoff[i] = 0; slot_stored[i] = 0; slot_bits[i] = 0; }

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

  // Bit-packed predicate projections (vh_table_predpack) of the filter's columns: the byte-plane form serves every compiled kernel form,
  // the bit-sliced one the compacting kernels. Both are noted here when they exist; compile_kernel picks (the no-compaction form and the
  // streamed payload want rows, i.e. byte planes; everything else the bit-sliced planes) and asks for the one it lacked to be built
  // for the queries to come (predpack_auto).
  bool jit_predpack = false;
  pp_cols.clear();
  if (jit_try && jshape.npred > 0 && !(p->flags & (VH_PLAN_NO_NARROW | VH_PLAN_NO_PREDPACK))) {
    bool all_cols = true;
    for (int k = 0; k < jshape.npred; ++k) { all_cols &= jit_pred_col[k] >= 0; pp_cols.push_back(jit_pred_col[k]); }
    std::sort(pp_cols.begin(), pp_cols.end());
    pp_cols.erase(std::unique(pp_cols.begin(), pp_cols.end()), pp_cols.end());
    if (!all_cols) pp_cols.clear();
    VhPredPack* pb = all_cols ? predpack_usable(t, pp_cols, 0) : nullptr;
    VhPredPack* ps = all_cols ? predpack_usable(t, pp_cols, 1) : nullptr;
    if (pb && P.nslots + pb->nplanes <= VH_MAX_SLOTS) {
      jshape.pp_nplanes = pb->nplanes;
      for (int q = 0; q < pb->nplanes; ++q) {
        P.colbase[P.nslots] = pb->pbase[q]; P.colstride[P.nslots] = pb->pstride[q]; P.colpitch[P.nslots] = (uint32_t)pb->pwidth[q];
        jshape.pp_plane[q].slot = P.nslots++; jshape.pp_plane[q].width = pb->pwidth[q]; jshape.pp_plane[q].pos = pb->ppos[q];
      }
      for (int k = 0; k < jshape.npred; ++k) {
        const size_t at = (size_t)(std::find(pb->cols.begin(), pb->cols.end(), jit_pred_col[k]) - pb->cols.begin());
        pp_boff[k] = pb->bitoff[at]; pp_bbits[k] = pb->bitw[at];
      }
      jit_predpack = true;
    }
    if (ps && P.nslots < VH_MAX_SLOTS) {
      P.colbase[P.nslots] = ps->pbase[0]; P.colstride[P.nslots] = ps->pstride[0]; P.colpitch[P.nslots] = (uint32_t)ps->pitch;
      jshape.pp_sliced = 1; jshape.pp_slot = P.nslots++;
      for (int k = 0; k < jshape.npred; ++k) {
        const size_t at = (size_t)(std::find(ps->cols.begin(), ps->cols.end(), jit_pred_col[k]) - ps->cols.begin());
        pp_soff[k] = ps->bitoff[at]; pp_sbits[k] = ps->bitw[at];
      }
      jit_predpack = true;
    }
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
      if (jit_predpack && !have) return narrow_slot[col] = -1;       // (the compiled kernel reads the predicate projection: no copy of its own for this column)
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
    if (jit_try)      // (with a predicate projection at hand these are the fallback of the kernel forms it does not serve: existing copies only, see narrow_for)
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
  // (the pre-built kernel: at most VH_LANES_COLS group and VH_LANES_COLS metric columns; the compiled one keeps a step's payload in registers — up
  // to 40 bytes per row, 160 VGPRs, with the two blocks per CU it runs at)
  uint32_t lanes_bytes = 0;
  for (int i = 0; i < P.ngroup; ++i) lanes_bytes += std::max<uint32_t>(4, vh_elem_size(P.g[i].type()));
  for (int j = 0; j < P.nmetric; ++j) lanes_bytes += std::max<uint32_t>(4, vh_elem_size(P.m[j].type()));
  const bool lanes_cols_ok = jit_try ? lanes_bytes <= 40 : P.ngroup <= VH_LANES_COLS && P.nmetric <= VH_LANES_COLS;
  if (mode == VH_MODE_DENSE_LDS && (fast || (jit_try && fastj)) && !(p->flags & VH_PLAN_NO_LANES) && lanes_cols_ok &&
      P.nmetric >= 1 && rows_to_scan) {
    bool ok = true;
    // (the pre-built kernel loads 4- and 8-byte columns only; the compiled form of it any width. No compiled kernel after all: the query is planned
    // again with VH_PLAN_NO_JIT and comes back here with jit_try off)
    const int min_elem = jit_try ? 1 : 4;
    for (int i = 0; i < P.ngroup; ++i) ok &= vh_elem_size(P.g[i].type()) >= min_elem && P.g[i].gran() == VH_T_NONE && P.g[i].nroll() == 0;
    for (int j = 0; j < P.nmetric; ++j) ok &= P.m[j].slot() != VH_SLOT_ROWID && P.m[j].sop() != SOP_BITSET && vh_elem_size(P.m[j].type()) >= min_elem;
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
    const size_t part_table_bytes = test_env("VH_PART_TABLE_KB") ? (size_t)atoi(test_env("VH_PART_TABLE_KB")) * 1024 : 128 * 1024;      // (tests shrink it between two queries to force many ranges)   // one 1024-thread block per CU in phase 2 (160 KB LDS)
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
      // Round 3: with the scan compiled for the plan and two-word tuples leaving as whole lines (then the per-wave writer, now the block's ring writer) a tuple costs ~10 ps
      // against ~86 ps for its two direct atomics, and phase 2's fixed cost is ~0.1 ms: an eighth of C3 (125 M rows, 6.2 M survivors) runs
      // 0.47 ms partitioned against 0.60 ms direct, a quarter 0.78 against 1.12 (profiles/r03/NOTES.md). From 2 M survivors on, one level.
      if (!want_part && !two_level && jit_try && np <= VH_RING_PARTS) {
        int words = 1, halves = 1;           // tuple words this plan would need: 64-bit states own one, 32-bit ones pair up (word 0 has one half free)
        for (int j = 0; j < P.nmetric; ++j) { if (vh_sop_bytes(P.m[j].sop()) == 8) ++words; else if (halves) --halves; else { ++words; halves = 1; } }
        if (words == 2 && shard_rows * sel >= 2e6 && sel >= 0.015) want_part = true;      // (at 1 % of 1 B rows the atomics still hide behind the scan: 1.18 ms direct, 1.31 partitioned)
      }
    }
    // ONE-word tuples (decided here, used below): gid and every metric value fit 63 bits together — what the values need is known from the
    // columns' recorded min / max (refresh_stats keeps them for metric columns too). C3's (gid 17 bits, SUM value 10, COUNT 2) is 8 bytes
    // instead of 16: half the tuple bytes written by phase 1, moved by the split level and read back by phase 2. Only the compiled scan
    // with the whole-line writer packs them.
    int nt_gb = 0, nt_mb[VH_MAX_METRIC] = {};
    bool narrow_tuples = false, tuple4 = false;
    if (want_part && jit_try && (two_level ? (np + 63) / 64 : np) <= VH_RING_PARTS_MAX && !(p->flags & (VH_PLAN_NO_NARROW_TUPLES | VH_PLAN_FORCE_LANES)) && !test_env("VH_NO_NARROW_TUPLES")) {
      auto bits_of = [](uint64_t v) { int b = 1; while (b < 64 && (v >> b)) ++b; return b; };
      nt_gb = bits_of(G - 1);
      int used = nt_gb;
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
        nt_mb[j] = bits_of(sgn ? (khi ^ (1ull << 63)) : bits_of_order_key(c.elem, khi));
        used += nt_mb[j];
      }
      narrow_tuples = fits && used <= 63;
      // ... and FOUR bytes when they fit 32 bits. With two levels the tuple carries the gid RELATIVE to its level-1 partition (the low part_shift
      // bits: the partition is where the tuple lies) — 4 M groups in 8 partitions: 19 bits instead of 22, C3's values behind them: 31
      tuple4 = narrow_tuples && (two_level ? used - nt_gb + shift + 6 : used) <= 32 && !test_env("VH_NO_TUPLE4") && !(two_level && test_env("VH_NO_TUPLE4_TWO"));
      if (tuple4 && two_level) nt_gb = shift + 6;
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
            // (... nor when the tuples are one word: half the bytes beat the saved compaction at every selectivity, two levels included —
            // 4 M groups, every row passing: 23.1 ms through the no-compaction form, see profiles/r04/NOTES.md for the one-word figure)
            lanes = s2 >= 0.5 && !(jit_try && !two_level && np <= VH_RING_PARTS) && !narrow_tuples;
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
      P.gid_bits = 0; P.tuple4 = 0;
      if (narrow_tuples && !lanes) {
        P.gid_bits = nt_gb; P.tw = 1; P.tuple4 = tuple4 ? 1 : 0;
        int at = nt_gb;
        for (int j = 0; j < P.nmetric; ++j) { P.m[j].set_tword(0); P.m[j].set_tshift((uint8_t)at); P.m[j].tbits = (uint32_t)nt_mb[j]; at += nt_mb[j]; }
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

  // what the table remembers about a query shape goes by its group columns: groups of the last query (hash sizing), clustered survivors (tuple extents)
  std::string sig;
  if (mode == VH_MODE_HASH || mode == VH_MODE_DENSE_PART) {
    for (int i = 0; i < p->ngroups; ++i) sig += std::to_string(p->groups[i].col) + ":" + std::to_string(P.g[i].gran()) + ":" + std::to_string(P.g[i].nroll()) + ",";
    r->group_sig = sig;
  }
  if (mode == VH_MODE_HASH) {
    // sizing: explicit override (regrow) > caller's hint > what the same group columns produced last time > 1 M
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
  if (mode == VH_MODE_HASH && jit_try && (!lanes || (p->flags & VH_PLAN_FORCE_HPART)) && !no_hpart && !(p->flags & VH_PLAN_NO_HPART) && !(t->hpart_hopeless.count(r->group_sig) && !(p->flags & VH_PLAN_FORCE_HPART)) && P.key_words == 1 && P.nmetric >= 1 && rows_to_scan) {
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
          bool fits = !(p->flags & VH_PLAN_NO_HP_PACK) && !test_env("VH_NO_HP_PACK");
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
          if (const char* e = test_env("VH_TEST_HP_IDBITS")) idbits = std::max(1, atoi(e));      // tests: ids that do NOT fit -> VH_ERR_HP_WIDE -> the plain hash table
          if (fits && idbits <= 32 && pbits + 2 * idbits <= 61) {
            hp_pack = true; hp_units = 1; hp_pbits = pbits; hp_idbits = idbits;
            for (int j = 0; j < P.nmetric; ++j) P.m[j].tbits = (uint32_t)mb[j];
          }
        }
        if (nb) {
          hp_off32 = !test_env("VH_NO_OFF32");
          for (uint32_t sgi : live) hp_off32 = hp_off32 && t->cols[bitset_col[0]].bs_offsets32[sgi] != nullptr;
        }
        // the scan partitions 256 ways by itself (one 1024-thread block per CU sharing the digits' waiting lines: vj_fan_add)
        hp_fan = true;
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
        if (const char* env_passes = test_env("VH_TEST_HPART_PASSES")) if (!hp_passes_override) passes = (uint32_t)std::max(1, atoi(env_passes));
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
    // A bit-field projection of 4-byte records that holds every value of the plan can be STREAMED by the compiled compacting scan (the
    // survivor's record goes into the wave's queue, nothing is gathered): 4 bytes per row whatever passes. Measured on C3 (tools/qpay_probe.py,
    // profiles/r05/NOTES.md): gathers win up to ~8 % selectivity, streaming from there on — 12 % faster at 25 %, 40 % at 50 %, 30 % at 100 %
    // (against the arenas, which is what a query above 15 % used to read).
    bool all_plain = P.nbitset == 0;
    for (int j = 0; j < P.nmetric; ++j) all_plain &= metric_col[j] >= 0;
    const bool qpay_can = jit_try && !lanes && all_plain && want && !(p->flags & VH_PLAN_NO_QPAY);
    bool want_gather = want, want_stream = false;
    if (want) {
      // lines touched per survivor: one record vs one per column; gathering stops paying off once most lines of
      // the arenas are touched anyway (C3 columns: ~20 % of the rows passing)
      if (!forced) want_gather = fastj && p->nfilter > 0;
      double sel = 1.0;
      if ((want_gather && !forced) || qpay_can) { rc = probed_selectivity(&sel); if (rc) { return rc; } }
      if (!forced) want_gather = want_gather && sel <= 0.15;
      want_stream = qpay_can && ((p->flags & VH_PLAN_FORCE_QPAY) || p->nfilter == 0 || sel >= knobs().qpay_min_sel);
      if (!forced) want = want_gather || want_stream;
    }
    const bool qpay_want = want_stream;
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
    if (use && !want_gather && !forced && !(use->bits && use->rec_bytes == 4)) use = nullptr;      // (asked for to be streamed, and it cannot be: the arenas)
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
        if (use->bits) { slot_bits[P.nslots] = (int)use->rec_bytes; slot_recoff[P.nslots] = (int)use->bitoff[k]; slot_stored[P.nslots] = (int)use->bitw[k]; }
        return pslot_of[col] = P.nslots++;
      };
      for (int i = 0; i < p->ngroups; ++i) P.g[i].set_slot((uint16_t)pslot(p->groups[i].col));
      for (int j = 0; j < P.nmetric; ++j) if (metric_col[j] >= 0) P.m[j].set_slot((uint16_t)pslot(metric_col[j]));
      packed = true;
      packed_compressed = use->compressed;
      if (qpay_want && use->bits && use->rec_bytes == 4) { qpay = 4; qpay_slot = pslot(gcols[0]); }      // (a bit-field record's members all start at the record: any member's slot is the record's)
    }
  }
  return VH_OK;
}


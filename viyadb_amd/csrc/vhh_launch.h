// vhh_launch.h — host side of libviya_hip, part of viya_hip.hip's translation unit (included there, in order; not a stand-alone header):
// QueryBuild continued: kernel compile, scan dispatch, work decomposition, scratch layout, launch; query_launch_locked.
// Adds a dense result's private copies (one per XCD, or one per block of a partition's range) into copy 0.
static void merge_args_of(const vh_result* r, VhMergeArgs* A) {
  const VhPlanDev& P = r->plan;
  *A = VhMergeArgs{};
  A->nmetric = P.nmetric; A->nxcd = r->nxcd; A->G = P.G; A->xcd_stride = P.xcd_stride; A->present = P.present;
  A->present_carrier = P.present_carrier;
  for (int j = 0; j < P.nmetric; ++j) { A->state[j] = P.m[j].state; A->sop[j] = P.m[j].sop(); }
  if (P.part_count && P.nlevel == 1) { A->part_count = P.part_count; A->npart = P.npart; A->agg_shift = P.agg_shift; A->blocks = r->part_blocks; }
}
static int merge_copies_now(vh_result* r, hipStream_t st) {
  if (!r->unmerged) return VH_OK;
  VhMergeArgs A;
  merge_args_of(r, &A);
  hipLaunchKernelGGL(dense_merge_kernel, dim3((unsigned)((r->plan.G + 255) / 256)), dim3(256), 0, st, A);
  HIP_TRY(hipGetLastError());
  r->unmerged = false;
  return VH_OK;
}

// A compiled scan ran (or will run) without the predicate projection that serves its form: count the query; at the VH_AUTO_NARROW-th one
// (the first, inside vh_table_prepare) build it for the queries to come, memory permitting. Only where kernels get compiled at all.
void QueryBuild::predpack_auto(bool want_sliced) {
  const int auto_after = g_preparing ? 1 : knobs().auto_narrow;
  if (pp_cols.empty() || auto_after <= 0) return;
  uint64_t rows = 0;
  for (uint32_t sgi = 0; sgi < t->nseg; ++sgi) rows += t->seg_rows[sgi];
  if (!(vh_jit_policy() == VH_JIT_FORCE || (p->flags & VH_PLAN_FORCE_JIT) || rows >= vh_jit_min_rows())) return;
  const bool sliced = want_sliced && !knobs().predpack_bytes;
  for (auto& q : t->predpacks) if (q->cols == pp_cols && q->sliced == sliced) return;
  std::string key = sliced ? "s:" : "b:";
  for (int c : pp_cols) { key += std::to_string(c); key.push_back(','); }
  if (++t->ppred_seen[key] < (uint32_t)auto_after) return;
  size_t free_b = 0, total_b = 0;
  const size_t need = (size_t)t->cap_seg * t->padded_rows * 4;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > need + total_b / 4) (void)table_predpack_locked(t, pp_cols, true, nullptr, sliced);
  else t->ppred_seen[key] = 0;
}

int QueryBuild::compile_kernel() {
  int rc = VH_OK; (void)rc;
  // ---------------- the scan kernel compiled for this plan shape (vh_jit.hip), when there is to be one
  // the no-compaction kernels are pre-built, except DENSE_LDS's over plain 4- / 8-byte arena columns (C2's shape), which has a compiled form
  if (jit_try && lanes && (mode != VH_MODE_DENSE_LDS || test_env("VH_TEST_NO_JIT_LANES"))) jit_try = false;
  if (jit_try) {
    VhJitShape& js = jshape;
    js.mode = mode;
    const size_t qw = (size_t)VJ_QUEUE_CAP * sizeof(uint32_t);       // per wave
    js.lanes = lanes ? 1 : 0;
    if (mode == VH_MODE_DENSE_LDS || (mode == VH_MODE_HASH && !hpart)) {          // an LDS table per block: the widest block whose table + queues stay within the 64 KB a module kernel may ask for
      jit_block = mode == VH_MODE_DENSE_LDS ? 1024 : 512;
      // (the no-compaction form lives off resident waves like its pre-built twin: 256-thread blocks, as many per CU as fit, unless the blocks'
      // table flushes — blocks x groups x metrics atomics at the end — would weigh more than that; same rule as below)
      if (lanes && (uint64_t)g_ctx.num_cu * 5 * G * std::max(1, P.nmetric) <= rows_to_scan / 32) jit_block = 256;
      const size_t lds_room = 64 * 1024 - 1024;          // (the kernels' own static LDS — vh_scan_block_end's counters — shares the 64 KB)
      while (jit_block > 256 && lds_table + (size_t)(jit_block / 64) * qw > lds_room) jit_block /= 2;
      if (lds_table + (size_t)(jit_block / 64) * qw > lds_room) jit_try = false;
      if (mode == VH_MODE_HASH && !P.lds_hash_slots) jit_block = 256;
    }
    if (hpart) jit_block = 1024;      // (one block per CU shares the 256 digits' waiting lines: vj_fan_add)
    js.block = jit_block;
    js.hp_fan = hpart ? 1 : 0;
    js.ablate = knobs().jit_ablate;      // measurement only (profiles/r03/NOTES.md): 1 = no gathers, 2 = nothing behind the gathers
    js.xcd = nxcd > 1 ? 1 : 0;
    js.scope = (mode == VH_MODE_DENSE_GLOBAL || mode == VH_MODE_DENSE_LDS) && nxcd > 1 ? (int)__HIP_MEMORY_SCOPE_WORKGROUP : (int)__HIP_MEMORY_SCOPE_AGENT;
    js.carrier = P.present_carrier;
    js.tw = mode == VH_MODE_DENSE_PART ? P.tw : 1;
    js.key_words = mode == VH_MODE_HASH ? P.key_words : 1;
    js.lds_hash = P.lds_hash_slots ? 1 : 0;
    js.gid32 = mode != VH_MODE_HASH && G <= 0xFFFFFFFFull;
    js.gid_bits = mode == VH_MODE_DENSE_PART ? P.gid_bits : 0;
    js.tuple4 = mode == VH_MODE_DENSE_PART && P.gid_bits && P.tuple4 ? 1 : 0;
    // one- and two-word tuples of up to 64 partitions leave through the BLOCK's ring writer (vj_part_ring_add): whole lines, extents by position with
    // the pool's shared overflow region behind them; wider tuples are appended piece by piece (vh_part_direct_add)
    part_ring = mode == VH_MODE_DENSE_PART && (P.tw == 2 || P.gid_bits) && P.npart <= VH_RING_PARTS_MAX;
    js.part_ring = part_ring ? (P.npart <= VH_RING_PARTS ? VH_RING_PARTS : VH_RING_PARTS_MAX) : 0;
    js.hpart = hpart ? 1 : 0;
    js.bs_off32 = hpart && hp_off32 ? 1 : 0;
    js.hp_pack = hp_pack ? 1 : 0; js.hp_pbits = hp_pbits; js.hp_idbits = hp_idbits;
    // (measurement, VH_HP_AGG_WAVES=6: bound the aggregation's registers so that three of its blocks fit a CU where its LDS tables allow
    // them — measured: 81 -> 73 registers, the same 1.33 ms; occupancy is not what it waits for. Off.)
    if (hpart && hp_pack && knobs().hp_agg_waves > 0) js.hp_agg_waves = knobs().hp_agg_waves;
    js.ng = P.ngroup; js.nm = P.nmetric;
    js.qpay = !lanes ? qpay : 0; js.qpay_slot = js.qpay ? qpay_slot : -1;
    {   // which predicate projection, if any (shape_filter noted what exists): rows -> byte planes, everything else -> bit-sliced planes
      const bool rows_form = lanes || js.qpay != 0;
      const bool have_sliced = js.pp_sliced != 0, have_bytes = js.pp_nplanes != 0;
      if (have_sliced && !rows_form && !(p->flags & VH_PLAN_NO_SLICED)) { js.pp_nplanes = 0; for (int k = 0; k < js.npred; ++k) { js.pp_off[k] = pp_soff[k]; js.pp_bits[k] = pp_sbits[k]; } }
      else { js.pp_sliced = 0; js.pp_slot = -1; for (int k = 0; k < js.npred; ++k) { js.pp_off[k] = pp_boff[k]; js.pp_bits[k] = pp_bbits[k]; } }
      if (!(p->flags & (VH_PLAN_NO_NARROW | VH_PLAN_NO_PREDPACK)) && (rows_form ? !have_bytes : !have_sliced)) predpack_auto(!rows_form);
    }
    {   // (the records' registers count against the packed predicate registers' budget)
      int nv = 0;
      if (js.pp_sliced) for (int k = 0; k < js.npred; ++k) nv += js.pp_bits[k];
      else if (js.pp_nplanes) for (int q = 0; q < js.pp_nplanes; ++q) nv += VH_SUBSTEPS * js.pp_plane[q].width;
      else for (int k = 0; k < js.npred; ++k) nv += VH_SUBSTEPS * js.pred[k].width;
      if (js.pp_sliced && nv > 64) { js.pp_sliced = 0; js.pp_slot = -1; }      // (more planes than registers to hold them: the columns themselves)
      if (js.qpay && nv + VH_SUBSTEPS * js.qpay > VJ_MAX_NV) { js.qpay = 0; js.qpay_slot = -1; }
    }
    for (int i = 0; i < P.ngroup; ++i) {
      const VhGroupDev& g = P.g[i];
      VhJitCol& c = js.g[i];
      c.slot = (int)g.slot(); c.type = (int)g.type(); c.pitch = (int)P.colpitch[g.slot()];
      c.rec = slot_rec[g.slot()]; c.off = slot_recoff[g.slot()]; c.stored = slot_stored[g.slot()]; c.bits = slot_bits[g.slot()];
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
      if (!c.rowid && !c.bitset) { c.slot = (int)m.slot(); c.pitch = (int)P.colpitch[m.slot()]; c.rec = slot_rec[m.slot()]; c.off = slot_recoff[m.slot()]; c.stored = slot_stored[m.slot()]; c.bits = slot_bits[m.slot()]; }
    }
    if (jit_try) {
      std::string jerr;
      jk = vh_jit_get(js, &jerr);
      if (!jk) {
        // no kernel for this shape (hipRTC missing, or the text did not compile): plan again for the pre-built kernels. The
        // failure is remembered per shape, so only the first query of the shape pays for the attempt.
        if (knobs().jit_verbose) fprintf(stderr, "vh: per-query kernel unavailable, falling back: %s\n", jerr.c_str());
        else {      // said ONCE per process, whatever the verbosity: a maintainer must be able to see this cliff (large scans run about half as fast)
          static std::once_flag told;
          std::call_once(told, [&] { fprintf(stderr, "viya_hip: no per-query compiled scan kernels in this process (%s): the pre-built interpreting kernels answer instead; "
                                                     "vh_result_info.reserved bit 5 tells per query (VH_JIT_VERBOSE=1 for every occurrence)\n", jerr.substr(0, 200).c_str()); });
        }
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
  bool lanes_narrow = false;          // the no-compaction form over 1- / 2-byte group or metric columns, or more than VH_LANES_COLS of either: only its compiled kernel takes those
  if (lanes && mode == VH_MODE_DENSE_LDS) {
    lanes_narrow = P.ngroup > VH_LANES_COLS || P.nmetric > VH_LANES_COLS;
    for (int i = 0; i < P.ngroup; ++i) lanes_narrow |= vh_elem_size(P.g[i].type()) < 4;
    for (int j = 0; j < P.nmetric; ++j) lanes_narrow |= vh_elem_size(P.m[j].type()) < 4;
  }
  if (!jk && (packed_compressed || (mode == VH_MODE_DENSE_PART && P.gid_bits) || lanes_narrow || hpart)) {      // compressed records / one-word tuples / narrow lanes columns / the hashed partitioning (its scan writes level A itself) and no compiled kernel to handle them after all: plan again for the pre-built ones
    vh_plan p2 = *p;
    p2.flags |= VH_PLAN_NO_JIT;
    holder.reset();
    done = true;
    return query_launch_locked(t, x, &p2, out, hash_capacity_override, force_hash, part_tuples_override, no_part, plan_only, summary_out, ag, device_rows, hp_passes_override, no_hpart || hpart);
  }
  return VH_OK;
}

void QueryBuild::scan_dispatch(int grid_, int* occ) {
  const size_t qb = (size_t)(BLOCK / 64) * VhScanCfg<256>::kQueueCap * sizeof(uint32_t);
  const size_t lds_ = ((mode == VH_MODE_DENSE_LDS || mode == VH_MODE_HASH) ? lds_table : 0) + qb;
  hipStream_t s_ = x->stream();
  if (jk) {
    const size_t jl = ((mode == VH_MODE_DENSE_LDS || (mode == VH_MODE_HASH && !hpart)) ? lds_table : 0) + (size_t)(BLOCK / 64) * (VJ_QUEUE_CAP * sizeof(uint32_t)) +
                      (jshape.hp_fan ? VJ_FAN_LDS_BYTES(BLOCK) : 0) + (jshape.part_ring ? VH_RING_LDS_BYTES(jshape.part_ring, 2, BLOCK) : 0);
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
      if (hp_fan) snprintf(hn, sizeof(hn), " + hp_ring_scatter_kernel<1024, %d> + ", hp_units);      // (the scan wrote level A itself; level B without barriers)
      else snprintf(hn, sizeof(hn), " + hp_scatter_kernel<1024, %d> + hp_scatter_kernel<1024, %d> + ", hp_units, hp_units);
      r->kernel += hn + jk->name + "_hpagg";
    }
    const std::string pagg = jit_pagg() ? " + " + jk->name + "_pagg" : std::string(" + part_agg_kernel<1024>");
    if (mode == VH_MODE_DENSE_PART) r->kernel += P.nlevel != 2 ? pagg : (P.tw == 2 || P.gid_bits) ? (std::string(" + part_split_ring_kernel<256, ") + (P.tuple4 ? "4>" : P.gid_bits ? "8>" : "16>") + pagg) : " + part_split_kernel<256>" + pagg;
  }
  int occupancy = 0;
  if (env_bpc <= 0) scan_dispatch(0, &occupancy);
  const uint32_t step = BLOCK * (jk && jshape.pp_sliced ? 32u : (uint32_t)VH_LANE_ROWS);      // (bit-sliced predicates: a lane owns 32 consecutive rows per step)
  const uint64_t padded = (t->segment_rows + step - 1) / step * step;
  // all blocks co-resident (the compacting kernels need ~100-130 VGPRs: 4 waves/SIMD), units small enough
  // that the static round-robin leaves < 2 % imbalance
  // The compiled kernels that WRITE tuples (DENSE_PART phase 1, hashed partitioning) run best with fewer resident waves than their
  // 56-69 VGPRs allow: every wave keeps a line or an extent open per partition, and what eight blocks per CU keep open no longer
  // stays in L2 until it is complete (profiles/r03/NOTES.md, "Blocks per CU": a 125 M-row C3 shard 0.44 -> 0.39 ms with 3 instead of
  // 6, C5's scan 1.95 -> 1.6 ms with 4 instead of 8, and the scatter behind it finds fewer half-empty extents)
  // (the compiled no-compaction kernel has a whole step's payload in flight per wave — 16 rows x every column per lane: two 256-thread blocks per
  // CU already stream at full rate, and every block fewer is a table flush fewer: C2 0.315 -> 0.308 ms, 400 M rows 1.217 -> 1.179)
  // (through the block's ring writer a block keeps its partitions' lines open, not every wave: four blocks per CU — over four fresh processes each
  // 1.296-1.312 ms per C3 query against 1.265-1.405 with three and 1.31-1.48 with five; an eighth of the table 0.283 against 0.294)
  const int occ_cap = jk && mode == VH_MODE_DENSE_PART ? (jshape.part_ring ? 4 : 3) : jk && hpart ? 4 : jk && lanes ? 2 : 8;
  int blocks_per_cu = env_bpc > 0 ? env_bpc : occupancy > 0 ? std::min(occupancy, occ_cap) : (BLOCK == 1024 ? 1 : 4);
  if (const char* e = test_env("VH_TEST_BLOCKS_PER_CU")) { if (atoi(e) > 0) blocks_per_cu = occupancy > 0 ? std::min(occupancy, atoi(e)) : atoi(e); }   // (measurement: switched between two queries of one process)
  uint32_t unit_rows = step;
  // (units of 4 096 rows lose to longer ones — an eighth of C3 0.342 -> 0.332 ms with 16 K rows, a quarter 0.569 -> 0.542 and the whole table
  // 1.867 -> 1.841 with 32 K, 64 K no better — so a block only needs about eight of them to keep the round-robin even: tools/env_ab_probe.py)
  const uint64_t want_units = (uint64_t)g_ctx.num_cu * blocks_per_cu * (jk ? 8 : 64);
  while (unit_rows * 2 <= (jk ? 32768u : 65536u) && unit_rows * 2 <= padded &&
         (uint64_t)nseg * ((padded + unit_rows * 2 - 1) / (unit_rows * 2)) >= want_units) unit_rows *= 2;
  if (env_unit >= (int)step) unit_rows = (uint32_t)env_unit / step * step;
  if (const char* e = test_env("VH_TEST_UNIT_ROWS")) { if (atoi(e) >= (int)step) unit_rows = (uint32_t)atoi(e) / step * step; }   // (measurement: switched between two queries of one process)
  P.unit_rows = unit_rows;
  P.units_per_seg = (uint32_t)((t->segment_rows + unit_rows - 1) / unit_rows);
  P.nseg = nseg;
  P.total_units = nseg * P.units_per_seg;
  const int env_grid = knobs().grid;
  grid = env_grid > 0 ? env_grid : (int)std::max<uint64_t>(1, std::min<uint64_t>(P.total_units, (uint64_t)g_ctx.num_cu * blocks_per_cu));
  // The second pass over heavy ranges runs WHILE the first pass's rows cross PCIe (result_finalize): the 64 blocks that push them take wave slots
  // on 64 CUs, and a grid that fills the chip exactly then leaves some of its blocks waiting for a whole other block's share (measured: the pass
  // 8.3 -> 10.7-14.7 ms). Eight times the blocks, an eighth of the units each: the dispatcher hands them out as slots come free (9.0-11.9 ms; what is left is the PCIe traffic's
  // own cost to everything that is dispatched beside it).
  if (g_heavy.only && env_grid <= 0) grid = (int)std::max<uint64_t>(1, std::min<uint64_t>(P.total_units, (uint64_t)g_ctx.num_cu * blocks_per_cu * 8));
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
  r->hp_direct = hpart && r->nhaving == 0 && r->topk == 0;
  // ... or, in ONE launch, only takes its rows' places off the counter of its region (VH_HP_REGIONS, default off): every range of the aggregation
  // ends with a returning atomic on the result's row counter — 65 536 of them per query, each a block-wide wait. Measured: spreading them over
  // eight words buys nothing (profiles/r05/NOTES.md)
  const bool hp_streamed = knobs().hp_stream > 0 ? capacity >= (1ull << 22) : test_env("VH_TEST_HP_STREAM") != nullptr;
  const bool hp_regions = !hp_streamed && knobs().hp_regions > 1 && capacity >= (1ull << 22);
  if (r->hp_direct && !device_rows && (hp_streamed || hp_regions)) {
    int nch = std::min(hp_regions ? knobs().hp_regions : knobs().hp_stream > 0 ? knobs().hp_stream : 4, VH_HP_CHUNKS);
    while (HP_FAN % nch) --nch;
    r->hp_chunks = nch;
    r->hp_one_launch = hp_regions;
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
  // Phase 2's blocks by the partitions' tuple counts (vh_part_shares): one-level DENSE_PART whose phase 1 goes through the ring writer (it counts), with a
  // private copy of its range per phase-2 block — then a hot partition may take most of the blocks, and as many copies: room for up to 128 of them
  // (256 MB of states at most; the merge only reads the copies a partition's blocks wrote). Not for results small enough for the one-block tail.
  part_balanced = mode == VH_MODE_DENSE_PART && P.nlevel == 1 && jk && jshape.part_ring != 0 && part_bpp > 1 && nxcd == part_bpp && !knobs().skip_phase2 &&
                  !test_env("VH_NO_PART_BALANCE") &&
                  !(r->out_cap <= VH_SMALL_TAIL_MAX && (uint64_t)r->out_cap * (uint64_t)nxcd * (uint64_t)std::max(1, (int)P.nmetric) <= VH_SMALL_TAIL_STATES);
  if (part_balanced) {
    size_t per_copy = P.xcd_stride;      // presence bytes + states
    for (int j = 0; j < P.nmetric; ++j) per_copy += P.xcd_stride * (size_t)vh_sop_bytes(P.m[j].sop());
    const int room = (int)std::max<size_t>(1, ((size_t)256 << 20) / std::max<size_t>(per_copy, 1));
    nxcd = std::max(nxcd, std::min(std::min(128, room), P.npart * part_bpp));
    P.nxcd = nxcd; r->nxcd = nxcd;
  }
  r->part_blocks = (uint32_t)(P.nfine * part_bpp);
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
  if (part_balanced) o_pcount = sp.take(VH_MAX_PART * VH_PART_COUNT_WAYS * sizeof(uint32_t));
  // hashed partitioning whose groups leave straight into the output columns, planned by a caller that can run a second pass (vh_query_agg): ranges
  // that are too heavy for a block's LDS tables are marked in a bitmap of the 65 536 ranges instead of voiding the attempt
  heavy_marks = hpart && r->hp_direct && !r->hp_chunks && !device_rows && g_heavy.allow_mark && !g_heavy.only && P.hp_passes == 1 && !test_env("VH_NO_HEAVY_PASS");
  if (heavy_marks) o_heavy = sp.take(65536 / 8);
  if (hpart) part_tuple_cap = hp_tuple_cap;
  if (mode == VH_MODE_DENSE_PART || hpart) {
    // extent size: big enough that a wave allocates rarely (every allocation is a returning global
    // atomic = a full round trip the wave sits out), small enough that open extents do not waste HBM
    const uint64_t waves = (uint64_t)grid * (uint64_t)(BLOCK / 64);
    // ... and small enough that a wave fills about four of them per partition: the last extent of every (wave, partition) stays part
    // full, and phase 2 walks part-full extents at the price of full ones (C3: 1250 tuples per wave and partition — extents of 1024
    // were 61 % full on average, of 256 they are 90 %: kernels 2.07-2.12 -> 2.00-2.01 ms, an eighth of the table 0.36-0.38 -> 0.35-0.36)
    uint64_t et = 256;                 // a tile writes whole runs (<= VH_PART_TILE tuples) that must fit a fresh extent
    while (et < 4096 && et * 2 <= part_tuple_cap / (waves * P.npart) / 4) et *= 2;
    if (knobs().ext_tuples) et = std::max(256, knobs().ext_tuples);     // measurement
    if (hpart) et = HP_ET / hp_units;  // (the tiles of hp_scatter_kernel are whole source extents: 64 KB of tuples)
    const bool ring1 = jk && jshape.part_ring != 0;      // phase 1 through the block's ring writer: (block, partition) streams, extents by position
    const uint64_t per1 = (uint64_t)grid * (uint64_t)std::max(1, (int)P.npart);
    if (ring1) { et = 256; while (et < 4096 && et * 2 <= part_tuple_cap / per1 / 4) et *= 2; }      // (a power of two: the writer shifts)
    const uint64_t ext_tuples = et;
    P.ext_tuples = (int32_t)ext_tuples;
    // extents of pool 1 start one 128-byte line further apart than they are long (not the stream pools of the hashed partitioning, whose
    // reader takes extents as whole tiles): see VhPlanDev::ext_stride
    const uint64_t ext_stride = hpart ? ext_tuples : ext_tuples + (P.tuple4 ? ((uint64_t)knobs().ext_pad + 31) / 32 * 32 : P.gid_bits ? ((uint64_t)knobs().ext_pad + 15) / 16 * 16 : (uint64_t)knobs().ext_pad);      // (whole 128-byte lines: 8 two-word tuples, 16 one-word ones, 32 four-byte ones)
    P.ext_stride = (int32_t)ext_stride;
    uint64_t max_ext = part_tuple_cap / ext_tuples + waves * (P.npart + VH_EXT_CHUNK) + 64;
    if (max_ext > 0xFFFFFFF0ull) max_ext = 0xFFFFFFF0ull;
    if (hpart) max_ext = 64;      // (the scan writes the level-A pool itself — the plan's second pool —: nothing goes into this one)
    // the ring writer's pool: every (block, partition) stream its share of evenly spread tuples and one more extent BY POSITION, and behind those the
    // shared overflow region — room for all the expected tuples once more, so that ANY skew between the streams fits (one partition taking everything
    // included); only more survivors than estimated void the attempt, and the re-run is sized for the survivors it counted
    uint64_t pos1 = 0;
    if (ring1) { pos1 = part_tuple_cap / per1 / ext_tuples + 1; max_ext = std::min<uint64_t>(pos1 * per1 + part_tuple_cap / ext_tuples + per1 + 64, 0xFFFFFFF0ull); }      // (+ per1: every stream's last, part-filled overflow extent)
    if (!part_tuples_override && test_env("VH_TEST_PART_EXTENTS")) max_ext = std::max(1, atoi(test_env("VH_TEST_PART_EXTENTS")));   // tests: make the first attempt run out of extents
    const uint32_t test_levels = test_env("VH_TEST_POS_LEVELS") ? (uint32_t)std::max(0, atoi(test_env("VH_TEST_POS_LEVELS"))) : ~0u;      // tests: few (or no) positional levels — the tuples go through the overflow regions
    P.slice_levels_cap = test_levels;
    if (ring1) P.pos_levels = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(pos1, max_ext / per1), test_levels);
    P.max_extents = (uint32_t)max_ext;
    // the scan's waves take their extent chunks by position (no shared cursor, no returning atomics: VhPlanDev::ext_waves). The pool above
    // holds that whenever the waves' tuple counts agree within the 25 % the estimate leaves; a re-run after VH_ERR_PART_FULL goes back to
    // the cursor, which packs the chunks whatever the imbalance
    P.ext_waves = ring1 || part_tuples_override || test_env("VH_TEST_EXT_CURSOR") || t->part_clustered.count(r->group_sig) ? 0u : (uint32_t)grid * (uint32_t)(BLOCK / 64);
    o_tuples = sp.take(max_ext * ext_stride * (P.tuple4 ? 4 : P.tw * 8));
    o_emiss = sp.take(max_ext * sizeof(uint16_t));
    o_epart = sp.take(max_ext);
    if (P.nlevel == 2) {
      // pool 2: small extents (4096 ranges x every splitting wave keep one open), sized like pool 1 plus what stays open
      split_bpp = std::max(1, 2 * g_ctx.num_cu / std::max(1, P.npart));     // 256-thread blocks, ~2 per CU whatever the partition count: more waves keep more extents open (4 per CU measured slower)
      // one- and two-word tuples are split through the ring writer (part_split_ring_kernel): a block is one writer, more of them cost less
      const bool tiled = P.tw == 2 || P.gid_bits;
      if (tiled) split_bpp = std::max(1, 4 * g_ctx.num_cu / std::max(1, P.npart));
      const uint64_t et2 = tiled ? VH_SPLIT_TILE_TUPLES : 256;
      P.ext_tuples2 = (int32_t)et2;
      uint64_t max2 = (part_tuple_cap + part_tuple_cap / 4) / et2 + (uint64_t)P.npart * ((uint64_t)split_bpp * (tiled ? 1 : 4) * (64 + VH_EXT_CHUNK) + 1) + 64;
      split_ring = tiled;
      // (slices laid out on the device from the partitions' counted tuples — vh_slice_extents: positional extents + an overflow region as big as the count)
      if (split_ring) max2 = 2 * (part_tuple_cap / et2) + (uint64_t)P.npart * (2 * 64 * split_bpp + 2) + 64;
      if (max2 > 0xFFFFFFF0ull) max2 = 0xFFFFFFF0ull;
      if (!part_tuples_override && test_env("VH_TEST_PART_EXTENTS2")) max2 = std::max(1, atoi(test_env("VH_TEST_PART_EXTENTS2")));   // tests: the second pool runs out first
      P.max_extents2 = (uint32_t)max2;
      o_tuples2 = sp.take(max2 * et2 * (P.tuple4 ? 4 : P.tw * 8));
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
      // level A: extents by position — extent k of (scan block, digit) is k * blocks * 256 + block * 256 + digit: a (block, digit)'s share and one more —,
      // then the shared overflow region: all the tuples once more (any skew between the digits fits)
      const uint64_t per = (uint64_t)grid * HP_FAN, posa = cap / per / hp_et + 1;
      uint64_t ma = posa * per + cap / hp_et + per + 64;
      if (!part_tuples_override && test_env("VH_TEST_PART_EXTENTS")) ma = std::max(1, atoi(test_env("VH_TEST_PART_EXTENTS")));      // tests: the first attempt's pool is too small
      P.pos_levels2 = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(posa, ma / per), test_env("VH_TEST_POS_LEVELS") ? (uint64_t)std::max(0, atoi(test_env("VH_TEST_POS_LEVELS"))) : ~0ull);
      // level B: hp_plan_kernel's slices — positional extents + an overflow region as big as the partition's count (vh_slice_extents).
      hp_ring_nb = 1;      // (level-B blocks per partition. Measured, C5: two or four of them write level B no faster — 0.425 / 0.421 / 0.397 ms — and leave the
                           //  aggregation two or four extents per range to walk: 1.08 / 1.24 / 1.69 ms)
      uint64_t mb = 2 * (cap / hp_et) + (uint64_t)HP_FAN * (2 * HP_FAN * hp_ring_nb + 2) + 64;
      if (!part_tuples_override && test_env("VH_TEST_PART_EXTENTS2")) mb = std::max(1, atoi(test_env("VH_TEST_PART_EXTENTS2")));   // tests: the last pool runs out
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
    const uint64_t ids = g_heavy.only ? std::min<uint64_t>(bitset_ids[b], g_heavy.ids_bound) : bitset_ids[b];      // (a heavy pass: the ids its ranges can hold)
    while (cap < ids * 2) cap <<= 1;
    P.dset_mask[b] = cap - 1;
    if (P.bs_wide[b]) { o_dkeys[b] = sp.take(cap * 16); o_dtags[b] = sp.take(cap * 4); }
    else {
      if (table_n >= 0xFFFFFFFFull) { return vh_fail(VH_E_UNSUPPORTED, "count-distinct over more than 2^32 group slots"); }
      o_dkeys[b] = sp.take(cap * 8);
    }
  }
  rc = ensure_scratch(x, sp.off);
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
  P.part_count = part_balanced ? reinterpret_cast<uint32_t*>(S + o_pcount) : nullptr;
  P.heavy_mark = heavy_marks ? reinterpret_cast<uint32_t*>(S + o_heavy) : nullptr;
  P.heavy_only = g_heavy.only;
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
        // (the compiled scan of the hashed partitioning reads the 32-bit copies of the offsets: the same table of pointers, other arrays)
        HIP_TRY(hipMemcpy(S + o_bsptr[b][0], hpart && hp_off32 ? (const void*)c.bs_offsets32.data() : (const void*)c.bs_offsets.data(), nseg * 8, hipMemcpyHostToDevice));
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
  if (int frc = derived_fence(t, st)) return frc;      // derived layouts refreshed for this query are still on the table's stream
  HIP_TRY(hipEventRecord(x->ev[0], st));
  VhInitArgs IA{};              // everything that is cleared goes into one launch (init_regions_kernel)
  auto clear = [&](void* ptr, size_t bytes, uint32_t byte_pattern) {
    if (!bytes) return;
    const uint64_t units = (bytes + 15) / 16;                       // (regions are padded to 256 B: rounding up stays inside)
    if (IA.n == VH_INIT_MAX) { (void)hipMemsetAsync(ptr, (int)(byte_pattern & 0xFFu), bytes, st); return; }
    IA.p[IA.n] = static_cast<char*>(ptr); IA.end[IA.n] = (IA.n ? IA.end[IA.n - 1] : 0) + units; IA.pat[IA.n] = byte_pattern * 0x01010101u; ++IA.n;
  };
  clear(P.counters, 512, 0);   // counters + out_count (adjacent 256 B slots)
  IA.cp_src = x->h_segrows; IA.cp_dst = reinterpret_cast<uint32_t*>(S + o_segrows); IA.cp_words = (uint32_t)r->plan_words;      // (copied by init_regions_kernel below: the counters' clear is always there)
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
  const bool part_owned = mode == VH_MODE_DENSE_PART && part_bpp > 1 && (nxcd == part_bpp || part_balanced) && !knobs().skip_phase2 && P.total_units != 0;      // (no units: phase 2 does not run and nobody stores the copies — they are cleared like any table)
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
    HA.list_cap = capacity; HA.chunk = hp_chunk; HA.ablate = knobs().hp_ablate; HA.slice_levels_cap = P.slice_levels_cap; HA.heavy_mark = P.heavy_mark;
    // no HAVING and no top-N to look at the groups first: the aggregation kernel emits them itself (C5: no 0.85 GB list, no 0.85 ms kernel)
    HA.direct = r->hp_direct ? 1 : 0; HA.ngroup = P.ngroup; HA.out_count = r->d_out_count;
    HA.nchunks = r->hp_chunks; HA.chunk_rows = r->hp_chunk_rows;
    for (int i = 0; i < P.ngroup; ++i) { HA.out_key[i] = r->d_out_key[i]; HA.gkey_shift[i] = P.g[i].key_shift(); HA.gesize[i] = (uint32_t)vh_elem_size(P.g[i].type()); }
    for (int j = 0; j < P.nmetric; ++j) { HA.out_state[j] = r->d_out_state[j]; HA.mesize[j] = (uint32_t)vh_elem_size(r->metric_elem[j]); }
    for (int k = 0; k < 1; ++k) {
      VhHpKind& K = HA.k[k];
      char* meta = S + hpo[k].meta;
      K.a.stride = K.b.stride = (uint32_t)(HP_ET / hp_units) + (uint32_t)knobs().ext_pad / (uint32_t)hp_units;
      K.a.tuples = reinterpret_cast<uint64_t*>(S + hpo[k].ta); K.a.fill = reinterpret_cast<uint16_t*>(S + hpo[k].fa); K.a.tag = reinterpret_cast<uint8_t*>(S + hpo[k].ga);
      K.a.max_extents = (uint32_t)std::min<uint64_t>(hpo[k].maxa, 0xFFFFFFF0ull);
      K.b.tuples = reinterpret_cast<uint64_t*>(S + hpo[k].tb); K.b.fill = reinterpret_cast<uint16_t*>(S + hpo[k].fb); K.b.tag = reinterpret_cast<uint8_t*>(S + hpo[k].gb);
      K.b.max_extents = (uint32_t)std::min<uint64_t>(hpo[k].maxb, 0xFFFFFFF0ull);
      K.b.ovf_base = K.b.max_extents; K.b.ovf_cursor = nullptr;      // (pool b's slices carry their own overflow regions and cursors)
      // the scan kernel's view of pool a (the second pool's fields of the plan: DENSE_PART's two-level plans are the other user); behind its positional
      // levels the shared overflow region the scan's writer takes extents from (counters[10])
      P.tuples2 = K.a.tuples; P.extent_missing2 = K.a.fill; P.extent_part2 = K.a.tag; P.max_extents2 = K.a.max_extents; P.ext_tuples2 = (int32_t)K.a.stride;
      K.a.ovf_base = (uint32_t)std::min<uint64_t>((uint64_t)P.pos_levels2 * (uint64_t)grid * HP_FAN, K.a.max_extents);
      K.a.ovf_cursor = P.counters + 10;
      K.count = reinterpret_cast<uint32_t*>(meta + 8);
      K.slice = reinterpret_cast<uint32_t*>(meta + 8 + (size_t)HP_FAN * 4);
      clear(K.a.fill, (size_t)K.a.max_extents * 2, 0);
      clear(K.b.fill, (size_t)K.b.max_extents * 2, 0);
      clear(meta, hp_meta_bytes, 0);
    }
    d_hpargs = reinterpret_cast<VhHpArgs*>(S + o_hpargs);
    r->d_hp_args = d_hpargs; r->hp_scan_blocks = grid;
    HIP_TRY(hipMemcpyAsync(d_hpargs, &HA, sizeof(HA), hipMemcpyHostToDevice, st));
  }
  if (part_balanced) clear(P.part_count, VH_MAX_PART * VH_PART_COUNT_WAYS * sizeof(uint32_t), 0);
  if (heavy_marks) clear(P.heavy_mark, 65536 / 8, 0);
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
    if (P.m[j].ident == 0 || P.hrec_bytes || hpart || part_owned) continue;      // (part_owned: phase 2's blocks store every group of every copy that is read)
    rc = fill_states(P.m[j].state, table_n, vh_sop_bytes(P.m[j].sop()), P.m[j].ident, st);
    if (rc) { return rc; }
  }
  if (IA.n) {
    const uint64_t units = IA.end[IA.n - 1];
    hipLaunchKernelGGL(init_regions_kernel, dim3((unsigned)std::min<uint64_t>((units + 255) / 256, (uint64_t)g_ctx.num_cu * 16)), dim3(256), 0, st, IA);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipEventRecord(x->ev[1], st));
  if (lanes && !jk)       // the pre-built lanes kernels read 4-byte predicate columns only (vh_preload<NP, false>)
    for (int k = 0; k < P.npred; ++k) if (P.pred_width[k] != 4) { P.pred_slot[k] = (uint8_t)pred_wide_slot[k]; P.pred_width[k] = 4; }
  bool narrowed = false;
  for (int k = 0; k < P.npred; ++k) narrowed |= P.pred_width[k] != 4;
  if (jk) { narrowed = jshape.pp_nplanes || jshape.pp_sliced; for (int k = 0; k < jshape.npred; ++k) narrowed |= jshape.pred[k].width != vh_elem_size(jshape.pred[k].type); }
  r->hpart = hpart;
  r->info.reserved = (hpart ? 64 : 0) | (fastj || jk ? 1 : 0) | (lanes ? 2 : 0) | (P.lds_hash_slots ? 4 : 0) | (packed ? 8 : 0) | (fastj && narrowed ? 16 : 0) | (jk ? 32 : 0) | (packed && packed_compressed ? 128 : 0) | (hpart && hp_pack ? 256 : 0) | (mode == VH_MODE_DENSE_PART && P.gid_bits ? 1024 : 0) | (jk && (jshape.pp_nplanes || jshape.pp_sliced) ? 2048 : 0) | (jk && jshape.qpay ? 4096 : 0) | (jk && jshape.pp_sliced ? 8192 : 0);
  if (r->hp_chunks) memset(x->h_chunk, 0, VH_HP_CHUNKS * sizeof(unsigned long long));      // (what the context's previous query left there)
  // a second pass over heavy level-A partitions of a hashed partitioning: from the first pass's tuples when this plan's table can take them as
  // they are (one key word laid out as the first pass's, narrow ids) — else from the table's rows behind the bitmap, like any heavy range
  bool from_tuples = false;
  if (const vh_result* m = g_heavy.only ? g_heavy.tuples_of : nullptr) {
    const VhPlanDev& M = m->plan;
    from_tuples = mode == VH_MODE_HASH && !hpart && !jk && P.key_words == 1 && M.key_words == 1 && P.ngroup == M.ngroup && P.nmetric == M.nmetric && P.nbitset <= 1 && m->d_hp_args;
    for (int i = 0; i < P.ngroup && from_tuples; ++i) from_tuples = P.g[i].key_shift() == M.g[i].key_shift() && P.g[i].key_word() == M.g[i].key_word();
    for (int b = 0; b < P.nbitset && from_tuples; ++b) from_tuples = !P.bs_wide[b];
    if (from_tuples) {
      VhHeavyTuples HT{};
      HT.HA = m->d_hp_args; HT.src_blocks = (uint32_t)m->hp_scan_blocks;
      HT.units = m->hp_args.units; HT.pk = m->hp_args.pk; HT.pbits = m->hp_args.pk_pbits; HT.idbits = m->hp_args.pk_idbits;
      HT.nmetric = P.nmetric; HT.bitset_j = -1;
      for (int j = 0; j < P.nmetric; ++j) {
        if (P.m[j].sop() == SOP_BITSET) HT.bitset_j = j;
        from_tuples = from_tuples && (P.m[j].sop() == SOP_BITSET) == (M.m[j].sop() == SOP_BITSET);
        HT.tshift[j] = (uint8_t)M.m[j].tshift(); HT.tbits[j] = M.m[j].tbits; HT.tbytes[j] = (uint8_t)vh_sop_bytes(M.m[j].sop()); HT.tsext[j] = vh_sop_sext(M.m[j].sop()) ? 1 : 0;
      }
      if (from_tuples) { vh_launch_heavy_tuples(P, HT, g_ctx.num_cu, st); r->kernel = HT.units == 2 ? "hp_heavy_tuples_kernel<2, false>" : HT.pk ? "hp_heavy_tuples_kernel<1, true>" : "hp_heavy_tuples_kernel<1, false>"; }
    }
  }
  if (g_heavy.only && !from_tuples && g_heavy.epoch != t->sync_epoch)       // the table has moved on since the first pass: its rows are no longer the rows that pass saw
    return vh_fail(VH_E_RANGE, "second pass over heavy ranges: the table changed since the first pass");      // (heavy_pass_finish gives up; the caller re-plans the whole query)
  if (from_tuples) {
  } else if (P.total_units) {
    scan_dispatch(grid, nullptr);
    if (hpart) {
      vh_launch_hpart(P, d_hpargs, hp_units, g_ctx.num_cu, grid, hp_ring_nb, st);
      if (!r->hp_chunks) HIP_TRY(vh_jit_launch_hpagg(jk, P, d_hpargs, hp_bpp, 0, HP_FAN * hp_bpp, lds_table, st));
      else if (r->hp_one_launch) {      // regions without streaming: one launch, every region's row count into pinned memory behind it
        HIP_TRY(vh_jit_launch_hpagg(jk, P, d_hpargs, hp_bpp, 0, HP_FAN * hp_bpp, lds_table, st));
        for (int c = 0; c < r->hp_chunks; ++c) hipLaunchKernelGGL(publish_count_kernel, dim3(1), dim3(64), 0, st, x->h_chunk + c, r->d_out_count + c);
        for (int c = 0; c < r->hp_chunks; ++c) HIP_TRY(hipEventRecord(x->ev_chunk[c], st));
      } else {
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
      if (P.nlevel == 2 && !skip_phase2) vh_launch_part_split(P, split_bpp, split_ring, st);
      if (!skip_phase2) {
        if (jit_pagg()) HIP_TRY(vh_jit_launch_pagg(jk, P, part_bpp, lds_table, st));      // phase 2 compiled for this plan's tuple layout
        else vh_launch_part_agg(P, part_bpp, lds_table, st);
      }
    }
  }
  HIP_TRY(hipEventRecord(x->ev[2], st));
  HIP_TRY(hipGetLastError());
  // a small dense result that will go straight into pinned host memory (result_finalize's `direct`): its whole tail is one launch there
  // ... bounded by the states that ONE block then reads (entries x private copies x metrics), not by the entries alone: beyond ~64 K of them
  // the three parallel launches it replaces are the faster tail
  r->small_tail = mode != VH_MODE_HASH && r->out_cap <= VH_SMALL_TAIL_MAX && r->out_region_bytes <= (8u << 20) && !r->topk_active && !r->hp_chunks &&
                  (uint64_t)r->out_cap * (uint64_t)std::max(1, nxcd) * (uint64_t)std::max(1, (int)P.nmetric) <= VH_SMALL_TAIL_STATES &&
                  !device_rows && !knobs().no_direct_emit && !test_env("VH_TEST_NO_SMALL_TAIL");
  if (mode != VH_MODE_HASH && nxcd > 1) {
    r->unmerged = true;
    if (!r->small_tail) { if (int mrc = merge_copies_now(r, st)) return mrc; }
  }
  *out = holder.release();
  done = true;
  return VH_OK;
}

static int query_launch_locked(vh_table* t, VhExec* x, const vh_plan* p, vh_result** out, uint64_t hash_capacity_override,
                               bool force_hash, uint64_t part_tuples_override = 0, bool no_part = false,
                               bool plan_only = false, VhSummary* summary_out = nullptr, const VhAgreed* ag = nullptr,
                               bool device_rows = false, uint32_t hp_passes_override = 0, bool no_hpart = false) {
  if (int src = sync_resolve(t)) return src;      // a batched sync may still be on its way: its rows and its share of the stats, before anything is planned
  QueryBuild b(t, x, p, out, hash_capacity_override, force_hash, part_tuples_override, no_part, plan_only, summary_out, ag, device_rows,
               hp_passes_override, no_hpart);
  int (QueryBuild::* const steps[])() = {&QueryBuild::shape_filter, &QueryBuild::snapshot_segments, &QueryBuild::shape_groups, &QueryBuild::shape_metrics,
                                         &QueryBuild::choose_organisation, &QueryBuild::plan_hashed_partitioning, &QueryBuild::choose_projection,
                                         &QueryBuild::compile_kernel, &QueryBuild::decompose_work, &QueryBuild::layout_scratch, &QueryBuild::launch};
  const bool timed = knobs().times;
  double us[12] = {};
  int k = 0;
  for (auto step : steps) {
    const auto s0 = timed ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
    if (int rc = (b.*step)()) return rc;
    if (timed) us[k] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - s0).count();
    ++k;
    if (b.done) break;
  }
  if (timed) fprintf(stderr, "vh plan steps (us): filter %.1f, segments %.1f, groups %.1f, metrics %.1f, organisation %.1f, hpart %.1f, projection %.1f, kernel %.1f, work %.1f, scratch %.1f, launch %.1f\n",
                     us[0], us[1], us[2], us[3], us[4], us[5], us[6], us[7], us[8], us[9], us[10]);
  return VH_OK;
}


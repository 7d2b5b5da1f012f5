// vh_jit_body.h — the hand-written frame of the per-query scan kernels.
//
// The reference compiles every query into straight-line C++ (filter: ComparisonBuilder, src/codegen/query/filter.cc:206-261;
// loop: src/codegen/query/scan.cc:57-65; key / Update: src/codegen/db/store.cc:31-169) and caches the shared object
// (src/codegen/compiler.cc:97-144). vh_jit.hip does the same for the GPU: per plan SHAPE (column types and widths, the
// filter tree, the table organisation — not the literals, which stay run-time arguments like the reference's `fargs`) it
// writes a small translation unit — a traits struct `VJ` with the shape as compile-time constants, the packed predicate
// loads, the filter as one C++ expression per row slot and the survivor's gathers — which includes THIS file for everything
// that does not depend on the query: scan geometry, compaction, the drain and the table organisations (the same device
// functions the pre-built kernels of vh_kernels.h use, now called with constants). hipRTC compiles it for gfx950.
//
// Differences to scan_agg_fast_kernel that the generated form makes possible:
//   * predicate columns stay PACKED in registers (a 1-byte column: 4 VGPRs per 16 rows, not 16) and are compared where they
//     sit — the compiler selects the byte / half-word with SDWA operand selectors, one v_cmp per row slot and predicate;
//   * a comparison's result IS the wave's ballot (an SGPR pair): conjunctions are s_and_b64, compaction ranks come from
//     v_mbcnt on that pair — no per-lane 16-bit masks, no second round of ballots;
//   * the drain knows column types, record layout, digit arithmetic and tuple layout.
#pragma once
#include "vh_kernels.h"

// (the wave's survivor queue holds VJ_QUEUE_CAP rows: vh_internal.h)
#ifndef VJ_ABL
#define VJ_ABL 0           // measurement builds only (VH_JIT_FLAGS=-DVJ_ABL=n): 8 = tuples are built, nothing is appended
#endif
#ifndef VJ_NT_GATHER
#define VJ_NT_GATHER 0     // 1: a survivor's record is fetched with non-temporal loads
#endif
typedef uint64_t vj_u64x2_a8 __attribute__((ext_vector_type(2), aligned(8)));
typedef uint32_t vj_u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
// (a pointer that was itself loaded from memory is "generic" to the compiler and gets flat_load, which waits on the LDS counter too;
// the bitset arrays are device allocations: say so)
#define VJ_GLOBAL(T, p) ((const T __attribute__((address_space(1)))*)(p))
#ifndef VJ_DRAIN2
#define VJ_DRAIN2 1        // 0 (measurement): the hashed-partitioning scan drains one survivor per lane at a time, as before
#endif
#ifndef VJ_BS_MERGE
#define VJ_BS_MERGE 1      // 0 (measurement): a row's offsets and first two ids with four scalar-width loads, as before
#endif
template <typename T> __device__ __forceinline__ T vj_gload(const T* p) {
  if (VJ_NT_GATHER) return __builtin_nontemporal_load(p);
  return *p;
}

template <class J, int I>
__device__ __forceinline__ uint64_t vj_time_rollup(uint64_t ts, const VhGroupDev& g) {
  constexpr bool micro = J::g_micro[I] != 0;
  // a `time` column proper: 32-bit seconds (src/db/column.cc:328-333), truncated in 32-bit arithmetic (vh_time.h: a MONTH is ~60 instructions, not
  // ~250). The C ABI also allows a non-micro time group column of 8 bytes (VH_U64 seconds): that one takes the 64-bit form, like the pre-built
  // kernels' vh_trunc_secs, so the two paths agree above 2^32
  constexpr bool secs32 = !micro && (J::g_type[I] == VH_U32 || J::g_type[I] == VH_I32);
  if constexpr (secs32) {
    uint32_t secs = (uint32_t)ts;
    bool done = false;         // the FIRST rule whose boundary lies beyond ts truncates (rollup.cc:77-95)
#pragma unroll
    for (int k = 0; k < J::g_nroll[I]; ++k) {
      if (!done && ts < g.roll_before[k]) { secs = vh_trunc_secs32(secs, J::g_roll_unit[I][k]); done = true; }
    }
    if (J::g_gran[I] != VH_T_NONE) secs = vh_trunc_secs32(secs, J::g_gran[I]);
    return (uint64_t)secs;
  } else {
    uint64_t secs = micro ? ts / 1000000ull : ts;
    uint64_t micros = micro ? ts % 1000000ull : 0ull;
    bool done = false;
#pragma unroll
    for (int k = 0; k < J::g_nroll[I]; ++k) {
      if (!done && ts < g.roll_before[k]) { secs = vh_trunc_secs(secs, J::g_roll_unit[I][k]); micros = 0; done = true; }
    }
    if (J::g_gran[I] != VH_T_NONE) { secs = vh_trunc_secs(secs, J::g_gran[I]); micros = 0; }
    return micro ? secs * 1000000ull + micros : secs;
  }
}

template <class J, int I = 0>
__device__ __forceinline__ void vj_rollup_all(const VhPlanDev& P, uint64_t (&gv)[J::NG ? J::NG : 1]) {
  if constexpr (I < J::NG) {
    if constexpr (J::g_gran[I] != VH_T_NONE || J::g_nroll[I] != 0) gv[I] = vj_time_rollup<J, I>(gv[I], P.g[I]);
    vj_rollup_all<J, I + 1>(P, gv);
  }
}

// What a wave carries through the scan besides its registers of predicate values: partition cursors, the waiting lines of the
// whole-line writer, its luck with the LDS front table.
struct VjWave {
  VhPartWave W;
  VhPartTile T;
  VhLdsHashWave H;
  VhRing F;
};


// ------------------------------------------------- hashed partitioning: the scan partitions by itself (J::HPART)
// The first level of vh_hpart.h's partitioning — a stream of tuples re-read and scattered 256 ways by the top byte of the mixed key — costs
// a write and a read of every tuple (C5: 2 of the 10 GB the query moves, 0.53 of its kernels' 3.2 ms). Here the scan block writes the
// level-A pool itself through the ring writer of vh_kernels.h (vh_ring_add): one 1024-thread block per CU, the digits' waiting lines in
// LDS behind the block's queues, and a (block, digit)'s k-th extent at k * (blocks * 256) + block * 256 + digit — by POSITION, no
// allocation. The pool is the plan's second pool (tuples2, extent_missing2 = tuples in the extent, extent_part2 = digit).
struct VjFanDest {
  uint64_t per, first; uint32_t stride, kmax;
  VhRingOvf ovf;
  // kmax: the pool's positional levels (VhPlanDev::pos_levels2); behind them its shared overflow region (cursor: counters[10])
  __device__ __forceinline__ VjFanDest(const VhPlanDev& P) : per((uint64_t)gridDim.x * VH_RING_FAN), first((uint64_t)blockIdx.x * VH_RING_FAN), stride((uint32_t)P.ext_tuples2), kmax(P.pos_levels2) {
    const uint64_t pos = (uint64_t)kmax * per;
    ovf.base = (uint32_t)(pos < (uint64_t)P.max_extents2 ? pos : (uint64_t)P.max_extents2); ovf.cap = P.max_extents2 - ovf.base;
    ovf.cur32 = nullptr; ovf.cur64 = P.counters + 10; ovf.fill = P.extent_missing2; ovf.tag = P.extent_part2;
  }
  // extent of the digit's k-th extent (~0: beyond its positions)
  __device__ __forceinline__ uint64_t extent(uint32_t d, uint32_t k) const { return k < kmax ? (uint64_t)k * per + first + d : ~0ull; }
};
template <int U>
__device__ __forceinline__ void vj_fan_add(const VhPlanDev& P, const VhRing& F, bool active, const uint64_t (&w)[2 * U], int lane) {
  const VjFanDest D(P);
  vh_ring_add<U>(F, reinterpret_cast<vh_u64x2*>(P.tuples2), D.stride, active, w, (uint32_t)(w[0] >> 56), lane, D, P.counters + 2);
}

// ------------------------------------------------- DENSE_PART phase 1 through the ring writer (J::PART_RING = digits the block keeps lines for: 16 or 64)
// Rounds 3-5 gave every WAVE a waiting line and an open extent per partition (3 072 waves x 13 partitions on C3: extents opened with chunk
// reservations, four line-flush passes per drain at 64 partitions, part-full extents that phase 2 walked at the price of full ones). Here the
// BLOCK shares the partitions' waiting lines (vh_ring_add_tb) and a (block, partition) stream's first extents of pool 1 lie at
// k * (blocks * npart) + block * npart + partition: a quarter of the open streams, no allocation, whole lines; what a stream holds beyond its
// positions comes out of the pool's shared overflow region. Tags and `missing` as phase 2 and the second split expect them.
struct VjPartDest {
  uint64_t per, first; uint32_t kmax;
  VhRingOvf ovf;
  __device__ __forceinline__ VjPartDest(const VhPlanDev& P) : per((uint64_t)gridDim.x * (uint32_t)P.npart), first((uint64_t)blockIdx.x * (uint32_t)P.npart), kmax(P.pos_levels) {
    const uint64_t pos = (uint64_t)kmax * per;
    ovf.base = (uint32_t)(pos < (uint64_t)P.max_extents ? pos : (uint64_t)P.max_extents); ovf.cap = P.max_extents - ovf.base;
    ovf.cur32 = nullptr; ovf.cur64 = P.counters + 9; ovf.fill = P.extent_missing; ovf.tag = P.extent_part;
  }
  __device__ __forceinline__ uint64_t extent(uint32_t d, uint32_t k) const { return k < kmax ? (uint64_t)k * per + first + d : ~0ull; }
};
template <class J, int TW>
__device__ __forceinline__ void vj_part_ring_add(const VhPlanDev& P, const VhRing& F, bool active, const uint64_t (&w)[TW], uint32_t part, int lane) {
  const VjPartDest D(P);
  vh_ring_add_tb<(J::TUPLE4 && TW == 1) ? 4 : TW * 8, VjPartDest, J::PART_RING, 2, true>(F, reinterpret_cast<char*>(P.tuples), (uint32_t)P.ext_stride, 31u - (uint32_t)__builtin_clz((uint32_t)P.ext_tuples), active, w, part, lane, D, P.counters + 2);
}

// Where a row's ids lie (hashed partitioning with a bitset metric) and the first two of them: loaded in two dependent steps, which a drain
// of two survivors per lane (vj_drain2) takes for both rows before it looks at either.
struct VjBits {
  uint64_t bk = 0, bk1 = 0;
  const uint32_t* bids = nullptr;
  uint32_t bid0 = 0, bid1 = 0;
};
template <class J>
__device__ __forceinline__ void vj_sink(const VhPlanDev& P, uint32_t seg, uint32_t row, bool active, uint64_t (&gv)[J::NG ? J::NG : 1], uint64_t (&mv)[J::NM ? J::NM : 1],
                                        char* lds, uint64_t xoff, unsigned long long& nfresh, VjWave& V, const VjBits* pre = nullptr);
// One surviving row per active lane: AggTuple key, then Metrics::Update (store.cc:131-161) into the plan's table organisation.
template <class J>
__device__ __forceinline__ void vj_drain(const VhPlanDev& P, uint32_t seg, uint32_t row, bool active, char* lds, uint64_t xoff,
                                         unsigned long long& nfresh, VjWave& V) {
  constexpr int NG = J::NG, NM = J::NM;
  if (!active) row = 0;
  uint64_t gv[NG ? NG : 1], mv[NM ? NM : 1];
  if constexpr (J::ABLATE & 1) {       // measurement builds (VH_JIT_ABLATE): no gathers, values made up from the row number
#pragma unroll
    for (int i = 0; i < NG; ++i) gv[i] = P.g[i].lo + (row & 63u);
#pragma unroll
    for (int j = 0; j < NM; ++j) mv[j] = row;
  } else if constexpr (J::QPAY != 0) {
    J::unpack(row, gv, mv);            // `row` IS the survivor's record (vj_slot queued it): nothing to fetch
  } else {
    J::gather(P, seg, row, gv, mv);    // every load of the survivor is issued before the first value is looked at
  }
  if constexpr (J::ABLATE & 2) {       // ... the gathers and nothing behind them
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < NG; ++i) acc += gv[i];
#pragma unroll
    for (int j = 0; j < NM; ++j) acc += mv[j];
    if (acc == 0x123456789ABCDEFull) P.counters[7] = acc;
    return;
  }
  vj_sink<J>(P, seg, row, active, gv, mv, lds, xoff, nfresh, V);
}
// Two surviving rows per active lane (hashed partitioning): a survivor's loads are two or three dependent round trips — its columns and
// where its ids lie, then the ids — and a wave that takes them one drain at a time sits out each of them with nothing else to do (few waves
// are resident: every one keeps extents open). Here both rows' loads of a step are in flight before either is looked at.
template <class J>
__device__ __forceinline__ void vj_drain2(const VhPlanDev& P, uint32_t seg, uint32_t row0, uint32_t row1, bool act0, bool act1, char* lds, uint64_t xoff,
                                          unsigned long long& nfresh, VjWave& V) {
  constexpr int NG = J::NG, NM = J::NM;
  if (!act0) row0 = 0;
  if (!act1) row1 = 0;
  uint64_t gv0[NG ? NG : 1], mv0[NM ? NM : 1], gv1[NG ? NG : 1], mv1[NM ? NM : 1];
  VjBits B0, B1;
  vj_bits_offsets<J>(P, seg, row0, act0, B0);
  vj_bits_offsets<J>(P, seg, row1, act1, B1);
  J::gather(P, seg, row0, gv0, mv0);
  J::gather(P, seg, row1, gv1, mv1);
  vj_bits_ids<J>(act0, B0);
  vj_bits_ids<J>(act1, B1);
  vj_sink<J>(P, seg, row0, act0, gv0, mv0, lds, xoff, nfresh, V, &B0);
  vj_sink<J>(P, seg, row1, act1, gv1, mv1, lds, xoff, nfresh, V, &B1);
}
// ... from the row's group and metric values on (the no-compaction form brings them in by itself: vj_lanes_rows)
// the two dependent steps of VjBits: offsets[row], offsets[row + 1] ...
template <class J>
__device__ __forceinline__ void vj_bits_offsets(const VhPlanDev& P, uint32_t seg, uint32_t row, bool active, VjBits& B) {
  if constexpr (J::MODE == VH_MODE_HASH && J::HPART && J::BITSET_J >= 0 && !(VJ_ABL & 4)) {
    const int b = (int)P.m[J::BITSET_J].slot();
    if (active) {
      // offsets[row], offsets[row + 1] in ONE 16-byte load (8-byte aligned), the first two ids in ONE 8-byte load (4-byte aligned; what lies
      // behind a segment's last id is readable: VH_BS_PAD): two address-processor passes per survivor instead of four
      B.bids = reinterpret_cast<const uint32_t*>(P.bs_vals[b][seg]);
      if (J::BS_OFF32) {          // 32-bit copies of the offsets (segments with < 2^32 ids): offsets[row], offsets[row + 1] in ONE 8-byte load
        const vj_u32x2_a4 o = *VJ_GLOBAL(vj_u32x2_a4, reinterpret_cast<const uint32_t*>(P.bs_offs[b][seg]) + row);
        B.bk = o.x; B.bk1 = o.y;
      } else if (VJ_BS_MERGE) {
        const vj_u64x2_a8 o = *VJ_GLOBAL(vj_u64x2_a8, P.bs_offs[b][seg] + row);
        B.bk = o.x; B.bk1 = o.y;
      } else {
        const uint64_t* offs = P.bs_offs[b][seg];
        B.bk = offs[row]; B.bk1 = offs[row + 1];
      }
    }
  }
}
// ... then the first two ids (the tuple says how many of the two count)
template <class J>
__device__ __forceinline__ void vj_bits_ids(bool active, VjBits& B) {
  if constexpr (J::MODE == VH_MODE_HASH && J::HPART && J::BITSET_J >= 0 && !(VJ_ABL & 4)) {
    if (active && B.bk < B.bk1) {
      if (J::BS_OFF32 || VJ_BS_MERGE) { const vj_u32x2_a4 i2 = *VJ_GLOBAL(vj_u32x2_a4, B.bids + B.bk); B.bid0 = i2.x; B.bid1 = i2.y; }
      else { B.bid0 = B.bids[B.bk]; B.bid1 = B.bk + 1 < B.bk1 ? B.bids[B.bk + 1] : 0u; }
    }
  }
}
template <class J>
__device__ __forceinline__ void vj_sink(const VhPlanDev& P, uint32_t seg, uint32_t row, bool active, uint64_t (&gv)[J::NG ? J::NG : 1], uint64_t (&mv)[J::NM ? J::NM : 1],
                                        char* lds, uint64_t xoff, unsigned long long& nfresh, VjWave& V, const VjBits* pre) {
  VhPartWave& W = V.W; VhPartTile& T = V.T; VhLdsHashWave& H = V.H;
  constexpr int MODE = J::MODE;
  constexpr int NG = J::NG, NM = J::NM;
  // hashed partitioning with a bitset metric: where the row's ids lie is asked for NOW, next to the record's loads, and the first two ids
  // as soon as that is known — they travel while the key is rolled up and mixed
  VjBits B;
  if (pre) B = *pre;           // (vj_drain2 brought them in for two rows at once)
  else { vj_bits_offsets<J>(P, seg, row, active, B); vj_bits_ids<J>(active, B); }
  uint64_t bk = B.bk, bk1 = B.bk1;
  const uint32_t* bids = B.bids;
  uint32_t bid0 = B.bid0, bid1 = B.bid1;
  vj_rollup_all<J>(P, gv);
  uint64_t gid = 0;
  uint64_t key[VH_KEY_WORDS];
  bool bad = false;
  if constexpr (MODE == VH_MODE_HASH) {
#pragma unroll
    for (int i = 0; i < VH_KEY_WORDS; ++i) key[i] = 0;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      uint64_t v = gv[i];
      if (J::g_type[i] == VH_F32 && (uint32_t)v == 0x80000000u) v = 0;            // -0.0f == 0.0f
      if (J::g_type[i] == VH_F64 && v == 0x8000000000000000ull) v = 0;
      key[J::g_key_word[i]] |= v << J::g_key_shift[i];
    }
  } else if constexpr (J::GID32) {
    // every digit in 32-bit wrap-around arithmetic: with v, lo in one 32-bit domain and lo + extent - 1 inside it, a value below
    // lo wraps to >= 2^32 - (lo - min) >= extent, so out-of-range digits are still caught (signed and unsigned alike)
    uint32_t g32 = 0;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const uint32_t d = (uint32_t)gv[i] - (uint32_t)P.g[i].lo;
      bad |= d >= (uint32_t)P.g[i].extent;
      g32 += d * (uint32_t)P.g[i].stride;
    }
    gid = g32;
  } else {
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const uint64_t d = gv[i] - P.g[i].lo;
      bad |= d >= P.g[i].extent;
      gid += d * P.g[i].stride;
    }
  }
  if constexpr (MODE == VH_MODE_HASH && J::HPART) {
    // hashed partitioning (vh_hpart.h): the row becomes a (mixed key, payload) tuple — with a bitset metric a 32-byte one that also carries
    // the row's first two ids, and further "ids only" tuples for a row with more than two; hp_scatter_kernel and hp_aggregate_kernel do the rest
    const int lane = (int)(threadIdx.x & 63);
    const uint64_t mkey = vh_splitmix64(key[0]);
    const uint32_t p = 0u;          // the tuples leave unpartitioned, 1-2 KiB per wave store (one "partition"): hp_scatter_kernel sorts them out
    uint64_t payload = 0ull;
#pragma unroll
    for (int j = 0; j < NM; ++j)
      if (J::m_sop[j] != SOP_BITSET) payload |= (vh_sop_bytes(J::m_sop[j]) == 4 ? (mv[j] & 0xFFFFFFFFull) : mv[j]) << J::m_tshift[j];
    if constexpr (J::BITSET_J >= 0 && J::HP_PACK && !(VJ_ABL & 4)) {
      // PACKED: the row in ONE 16-byte tuple — word 1 = payload | id 0 << PB | id 1 << (PB + IB) | ids that count << 61 (| ids only << 63 on
      // the tuples behind a row's first). The planner sized PB and IB from the columns' recorded min / max and the segments' largest id;
      // a value that needs more bits voids the attempt (VH_ERR_HP_WIDE: stats and data disagree — the plain hash table answers instead).
      constexpr int PB = J::HP_PBITS, IB = J::HP_IDBITS;
      constexpr uint64_t IMASK = (1ull << IB) - 1ull;
      uint64_t left = bk1 - bk;
      uint64_t pk = 0ull;
      bool wide = false;
#pragma unroll
      for (int j = 0; j < NM; ++j)
        if (J::m_sop[j] != SOP_BITSET) {
          const uint64_t v = vh_sop_bytes(J::m_sop[j]) == 4 ? (mv[j] & 0xFFFFFFFFull) : mv[j];
          if (J::m_tbits[j] < 64) wide |= (v >> J::m_tbits[j]) != 0ull;
          pk |= v << J::m_tshift[j];
        }
      if (left == 0) bid0 = 0u;
      if (left < 2) bid1 = 0u;            // (the merged load read whatever lies behind the row's only id)
      wide |= (((uint64_t)bid0 | (uint64_t)bid1) & ~IMASK) != 0ull;
      if (__ballot(active && wide)) { if (active && wide) atomicOr(P.counters + 2, VH_ERR_HP_WIDE); }
      const uint64_t words[2] = {mkey, pk | ((uint64_t)bid0 << PB) | ((uint64_t)bid1 << (PB + IB)) | ((left < 2 ? left : 2ull) << 61)};
      vj_fan_add<1>(P, V.F, active, words, lane);
      bool more = active && left > 2;
      while (__ballot(more)) {
        bk += 2; left -= 2;
        if (more) {
          if (VJ_BS_MERGE) { const vj_u32x2_a4 i2 = *VJ_GLOBAL(vj_u32x2_a4, bids + bk); bid0 = i2.x; bid1 = i2.y; }
          else { bid0 = bids[bk]; bid1 = left > 1 ? bids[bk + 1] : 0u; }
          if (left < 2) bid1 = 0u;
          if ((((uint64_t)bid0 | (uint64_t)bid1) & ~IMASK) != 0ull) atomicOr(P.counters + 2, VH_ERR_HP_WIDE);
        }
        const uint64_t w2[2] = {mkey, ((uint64_t)bid0 << PB) | ((uint64_t)bid1 << (PB + IB)) | ((left < 2 ? left : 2ull) << 61) | (1ull << 63)};
        vj_fan_add<1>(P, V.F, more, w2, lane);
        more = more && left > 2;
      }
    } else if constexpr (J::BITSET_J >= 0 && !(VJ_ABL & 4)) {
      // word 2: two ids, word 3: how many of them count | HP_IDS_ONLY (4) on the tuples behind a row's first
      uint64_t left = bk1 - bk;
      uint64_t words[4] = {mkey, payload, (uint64_t)bid0 | ((uint64_t)bid1 << 32), left < 2 ? left : 2ull};
      vj_fan_add<2>(P, V.F, active, words, lane);
      bool more = active && left > 2;
      while (__ballot(more)) {
        bk += 2; left -= 2;
        if (more) {
          if (VJ_BS_MERGE) { const vj_u32x2_a4 i2 = *VJ_GLOBAL(vj_u32x2_a4, bids + bk); bid0 = i2.x; bid1 = i2.y; }
          else { bid0 = bids[bk]; bid1 = left > 1 ? bids[bk + 1] : 0u; }
        }
        const uint64_t w2[4] = {mkey, 0ull, (uint64_t)bid0 | ((uint64_t)bid1 << 32), (left < 2 ? left : 2ull) | 4ull};
        vj_fan_add<2>(P, V.F, more, w2, lane);
        more = more && left > 2;
      }
    } else {
      const uint64_t words[2] = {mkey, payload};
      vj_fan_add<1>(P, V.F, active, words, lane);
    }
    return;
  }
  if constexpr (MODE == VH_MODE_HASH) {
    if constexpr (J::LDS_HASH) {
      if (!H.bypass) {
        uint32_t ls = 0;
        const bool in_lds = active && key[0] != VH_HASH_EMPTY && vh_lds_hash_find(P, lds, key[0], ls);
        const uint32_t nh = __popcll(__ballot(in_lds)), nm = __popcll(__ballot(active && !in_lds));
        H.hits += nh; H.misses += nm;
        if (H.hits + H.misses >= 4096u && H.misses > H.hits) H.bypass = true;
        if (in_lds) {
#pragma unroll
          for (int j = 0; j < NM; ++j) vh_state_update<__HIP_MEMORY_SCOPE_WORKGROUP>(lds + P.m[j].lds_off, ls, J::m_sop[j], mv[j]);
        }
        active = active && !in_lds;
      }
    }
    bool ok = true, fresh = false;
    if (active) {
      if constexpr (J::KEY_WORDS == 1) gid = vh_hash_insert64(P, key[0], ok, fresh);
      else gid = vh_hash_insert_wide(P, key, J::KEY_WORDS, ok, fresh);
    }
    bad = !ok;
    nfresh += __popcll(__ballot(active && fresh));
    if (__ballot(active && bad)) {
      if (active && bad) atomicOr(P.counters + 2, VH_ERR_HASH_FULL);
      H.dead = true;
    }
  } else if (__ballot(active && bad)) {
    if (active && bad) atomicOr(P.counters + 2, VH_ERR_RANGE);
  }
  active = active && !bad;
  if constexpr (MODE == VH_MODE_DENSE_PART && J::GID_BITS != 0) {
    // ONE-word tuples: gid in the low GID_BITS, every metric value behind it at the width the planner read off the column's recorded
    // min / max (refresh_stats) — half the bytes to write in phase 1 and to read back in phase 2, sixteen tuples per 128-byte line. A value
    // that needs more bits than recorded voids the attempt (VH_ERR_HP_WIDE: the query is answered with direct atomics instead).
    static_assert(J::TW == 1, "one-word tuples");
    // (four-byte tuples of a two-level plan: the gid's bits below part_shift — GID_BITS = part_shift there, the level-1 partition is where the tuple lies)
    uint64_t words[1] = {J::TUPLE4 ? gid & ((1ull << J::GID_BITS) - 1ull) : gid};
    bool wide = false;
#pragma unroll
    for (int j = 0; j < NM; ++j) {
      const uint64_t v = vh_sop_bytes(J::m_sop[j]) == 4 ? (mv[j] & 0xFFFFFFFFull) : mv[j];
      wide |= (v >> J::m_tbits[j]) != 0ull;
      words[0] |= v << J::m_tshift[j];
    }
    if (__ballot(active && wide)) { if (active && wide) atomicOr(P.counters + 2, VH_ERR_HP_WIDE); }
    if (VJ_ABL & 8) { if (words[0] == 0x123456789ABCDEFull) P.counters[7] = 1; return; }
    static_assert(J::PART_RING != 0, "one-word tuples leave through the block's ring writer");
    vj_part_ring_add<J, 1>(P, V.F, active, words, active ? (uint32_t)(gid >> P.part_shift) : 0u, (int)(threadIdx.x & 63));
    return;
  } else if constexpr (MODE == VH_MODE_DENSE_PART) {
    constexpr int TW = J::TW;
    uint64_t words[TW];
    words[0] = gid & 0xFFFFFFFFull;
#pragma unroll
    for (int w = 1; w < TW; ++w) words[w] = 0;
#pragma unroll
    for (int j = 0; j < NM; ++j) {
      const uint64_t v = (vh_sop_bytes(J::m_sop[j]) == 4 ? (mv[j] & 0xFFFFFFFFull) : mv[j]) << J::m_tshift[j];
      words[J::m_tword[j]] |= v;
    }
    if (VJ_ABL & 8) { uint64_t acc = 0; for (int w = 0; w < TW; ++w) acc += words[w]; if (acc == 0x123456789ABCDEFull) P.counters[7] = acc; return; }
    if (VJ_ABL & 0xF0) {      // measurement: (VJ_ABL >> 4) partitions fed round-robin by lane, i.e. contiguous runs of 64 / n tuples per store (wrong results)
      vh_part_direct_add<TW, 1, TW>(P, T, W, active, words, (uint32_t)((threadIdx.x & 63) % (VJ_ABL >> 4)), (int)(threadIdx.x & 63));
      return;
    }
    if constexpr (J::PART_RING != 0 && TW == 2) vj_part_ring_add<J, 2>(P, V.F, active, words, active ? (uint32_t)(gid >> P.part_shift) : 0u, (int)(threadIdx.x & 63));
    else vh_part_direct_add<TW, 1, TW>(P, T, W, active, words, (uint32_t)(gid >> P.part_shift), (int)(threadIdx.x & 63));
    return;
  }
  if constexpr (MODE == VH_MODE_DENSE_LDS) {
    if (active) reinterpret_cast<uint8_t*>(lds + P.lds_present_off)[gid] = 1;
  } else if constexpr (MODE == VH_MODE_DENSE_GLOBAL) {
    if constexpr (J::CARRIER < 0) { if (active) P.present[xoff + gid] = 1; }
  }
  if constexpr (MODE == VH_MODE_HASH) {
    // a wave most of whose survivors fall into ONE group lets one lane speak for them (vh_hot_lanes, vh_kernels.h: atomics on one address are
    // served one after the other — a hot key would otherwise cost ~6.5 ns per ROW)
    const uint64_t hot = vh_hot_lanes(active, gid);
    if (hot) {       // (wave-uniform)
      const int lane_ = (int)(threadIdx.x & 63);
      const bool in_hot = ((hot >> lane_) & 1ull) != 0, speaks = lane_ == __builtin_ctzll(hot);
#pragma unroll
      for (int j = 0; j < NM; ++j) {
        const uint64_t tot = vh_wave_combine(J::m_sop[j], mv[j], in_hot);
        if (in_hot ? speaks : active) vh_state_update<__HIP_MEMORY_SCOPE_AGENT>(vh_hash_state(P, P.m[j], gid), 0, J::m_sop[j], in_hot ? tot : mv[j]);
      }
      return;
    }
  }
  if (active) {
#pragma unroll
    for (int j = 0; j < NM; ++j) {
      if constexpr (MODE == VH_MODE_DENSE_LDS) vh_state_update<__HIP_MEMORY_SCOPE_WORKGROUP>(lds + P.m[j].lds_off, gid, J::m_sop[j], mv[j]);
      else if constexpr (MODE == VH_MODE_DENSE_GLOBAL) vh_state_update<J::SCOPE>(P.m[j].state, xoff + gid, J::m_sop[j], mv[j]);
      else vh_state_update<__HIP_MEMORY_SCOPE_AGENT>(vh_hash_state(P, P.m[j], gid), 0, J::m_sop[j], mv[j]);
    }
  }
}

// ---- bit-sliced predicate projections (J::SLICED; VhPredPack::sliced): a predicate column of NB bits is NB words per 32 rows, word b =
// bit b of those rows (x[0] the least significant). A comparison with a literal is a walk over the planes from the top bit down — a handful of
// bitwise operations per PLANE and 32 rows, whatever the relation — and its result is the lane's pass mask for its 32 rows: no compare, no
// ballot and no rank per row (BitWeaving/V's column-scalar comparison). `c` is wave-uniform (a literal): the branches on its bits are scalar.
template <int NB>
__device__ __forceinline__ void vj_bits_lt_eq(const uint32_t* x, uint64_t c, bool neg, uint32_t& lt, uint32_t& eq) {
  if (neg) { lt = 0u; eq = 0u; return; }                        // a negative literal: no (non-negative) value is below it or equals it
  if (NB < 64 && (c >> NB) != 0ull) { lt = ~0u; eq = 0u; return; }     // beyond the field's range: every value is below it
  const uint32_t cs = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)c);
  lt = 0u; eq = ~0u;
#pragma unroll
  for (int b = NB - 1; b >= 0; --b) {
    const uint32_t xb = x[b];
    if ((cs >> b) & 1u) { lt |= eq & ~xb; eq &= xb; } else { eq &= ~xb; }
  }
}
template <int NB>
__device__ __forceinline__ uint32_t vj_bits_eq(const uint32_t* x, uint64_t c, bool neg) {
  if (neg || (NB < 64 && (c >> NB) != 0ull)) return 0u;
  const uint32_t cs = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)c);
  uint32_t eq = ~0u;
#pragma unroll
  for (int b = NB - 1; b >= 0; --b) eq &= ((cs >> b) & 1u) ? x[b] : ~x[b];
  return eq;
}
// OP: enum vh_relop. The mask of the lane's 32 rows for which (column OP c) holds.
template <int NB, int OP>
__device__ __forceinline__ uint32_t vj_bits_rel(const uint32_t* x, uint64_t c, bool neg) {
  if constexpr (OP == VH_OP_EQ) return vj_bits_eq<NB>(x, c, neg);
  else if constexpr (OP == VH_OP_NE) return ~vj_bits_eq<NB>(x, c, neg);
  else {
    uint32_t lt, eq;
    vj_bits_lt_eq<NB>(x, c, neg, lt, eq);
    return OP == VH_OP_LT ? lt : OP == VH_OP_LE ? (lt | eq) : OP == VH_OP_GT ? ~(lt | eq) : ~lt;
  }
}
// The lane's passing rows (bits of `m`, row `base` + bit) appended to the wave's queue: ranks from a prefix sum of the lanes' counts, then
// every lane writes its own rows — as many rounds as the busiest lane has survivors. At most 1 024 rows a call (the queue's room).
__device__ __forceinline__ void vj_push_mask(uint32_t* q, uint32_t& cnt, uint32_t m, uint32_t base, int lane) {
  const uint32_t c = (uint32_t)__popc(m);
  uint32_t incl = c;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(incl, off); if (lane >= off) incl += o; }
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  if (total == 0u) return;
  uint32_t pos = cnt + incl - c;
  while (m) { q[pos++] = base + (uint32_t)__builtin_ctz(m); m &= m - 1u; }
  cnt += total;
}

// One row slot of the wave step: the generated predicate for slot I, its ballot, and the passing lanes' rows appended to the
// wave's queue. FULL = false: the step reaches the end of the segment's snapshot, rows at or beyond size() never pass.
template <class J, int I, bool FULL>
__device__ __forceinline__ void vj_slot(const typename J::Lits& L, const uint32_t (&v)[J::NV ? J::NV : 1], uint32_t row_l, uint32_t seg_rows,
                                        uint32_t* q, uint32_t& cnt) {
  const uint32_t row = row_l + (I >> 2) * 256u + (I & 3);
  // The slot's pass mask as the generated filter built it from the comparisons' own lane masks (v_cmp -> SGPR pair, s_and / s_or),
  // and the same predicate as this lane's bool: the two share every comparison, and the bool's lane mask IS the mask, so the branch
  // below is one s_and_saveexec on it. (__ballot(p) would rebuild the mask with v_cndmask + v_cmp per slot.)
  bool p;
  uint64_t bal = J::template pass<I>(L, v, p);
  if (!FULL) { const bool in = row < seg_rows; bal &= __builtin_amdgcn_ballot_w64(in); p = p & in; }
  // (streamed payload, J::QPAY: the passing row leaves its RECORD in the queue — it came in with the step's loads — not its number)
  uint32_t entry = row;
  if constexpr (J::QPAY != 0) entry = J::template payload<I>(v);
  if (p) q[__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, cnt))] = entry;
  cnt += (uint32_t)__popcll(bal);
}
template <class J, bool FULL, int I = 0>
__device__ __forceinline__ void vj_slots(const typename J::Lits& L, const uint32_t (&v)[J::NV ? J::NV : 1], uint32_t row_l, uint32_t seg_rows,
                                         uint32_t* q, uint32_t& cnt) {
  if constexpr (I < VH_LANE_ROWS) {
    vj_slot<J, I, FULL>(L, v, row_l, seg_rows, q, cnt);
    vj_slots<J, FULL, I + 1>(L, v, row_l, seg_rows, q, cnt);
  }
}

// The no-compaction form (J::LANES; DENSE_LDS with most rows passing — C2): the lane's pass bits of one step ...
template <class J, bool FULL, int I = 0>
__device__ __forceinline__ void vj_lanes_mask(const typename J::Lits& L, const uint32_t (&v)[J::NV ? J::NV : 1], uint32_t row_l, uint32_t seg_rows, uint32_t& mask) {
  if constexpr (I < VH_LANE_ROWS) {
    bool p;
    (void)J::template pass<I>(L, v, p);
    if (!FULL) p = p & (row_l + (I >> 2) * 256u + (I & 3) < seg_rows);
    mask |= (uint32_t)p << I;
    vj_lanes_mask<J, FULL, I + 1>(L, v, row_l, seg_rows, mask);
  }
}
// ... and its passing rows into the table, values out of the step's payload registers
template <class J, int I = 0>
__device__ __forceinline__ void vj_lanes_rows(const VhPlanDev& P, uint32_t seg, uint32_t row_l, uint32_t mask, const typename J::Payload& Y, char* lds, uint64_t xoff,
                                              unsigned long long& nfresh, VjWave& V) {
  if constexpr (I < VH_LANE_ROWS) {
    uint64_t gv[J::NG ? J::NG : 1], mv[J::NM ? J::NM : 1];
    J::template lanes_values<I>(Y, gv, mv);
    vj_sink<J>(P, seg, row_l + (I >> 2) * 256u + (I & 3), ((mask >> I) & 1u) != 0, gv, mv, lds, xoff, nfresh, V);
    vj_lanes_rows<J, I + 1>(P, seg, row_l, mask, Y, lds, xoff, nfresh, V);
  }
}

// The kernel body. Grid-stride over work units exactly like vh_scan_fast_body (same VhPlanDev, same unit decomposition, same
// counters), so the host plans and finalises a query the same way whichever kernel ran it.
template <class J>
__device__ __forceinline__ void vj_scan(const VhPlanDev& P) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int MODE = J::MODE, BLOCK = J::BLOCK;
  constexpr int kLaneRows = J::SLICED ? 32 : VH_LANE_ROWS;      // rows a lane owns per wave step: 32 consecutive ones (bit-sliced predicates), or 4 x 4
  constexpr int kStepRows = BLOCK * kLaneRows;
  constexpr uint32_t kWaveRows = 64u * kLaneRows, kLaneStride = J::SLICED ? 32u : 4u;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // wave-uniform, and the compiler is told so: rows, counts and
                                                                                 // ballots of the step then live in scalar registers
  uint32_t* q = reinterpret_cast<uint32_t*>(lds + ((MODE == VH_MODE_DENSE_LDS || MODE == VH_MODE_HASH) ? P.lds_bytes : 0)) + wave * VJ_QUEUE_CAP;
  VjWave V;
  V.H = VhLdsHashWave{0u, 0u, false, 0ull};
  if constexpr (MODE == VH_MODE_HASH && J::LDS_HASH) vh_lds_hash_init(P, lds, BLOCK);
  if constexpr (MODE == VH_MODE_DENSE_PART) vh_part_tile_init(P, lds, V.T, V.W);
  if constexpr (MODE == VH_MODE_HASH && J::HPART)     // the block's level-A writer (vj_fan_add), behind the block's queues
    vh_ring_init<BLOCK>(lds + (size_t)(BLOCK / 64) * VJ_QUEUE_CAP * sizeof(uint32_t), V.F, wave);
  if constexpr (MODE == VH_MODE_DENSE_PART && J::PART_RING != 0)   // the block's phase-1 writer (vj_part_ring_add), behind the block's queues
    vh_ring_init<BLOCK, J::PART_RING, 2>(lds + (size_t)(BLOCK / 64) * VJ_QUEUE_CAP * sizeof(uint32_t), V.F, wave);
  if constexpr (MODE == VH_MODE_DENSE_LDS) {
    // identities: 0 for SUM/AVG/COUNT, type max for MIN, cpp_min_value for MAX (src/codegen/db/store.cc:107-117)
#pragma unroll
    for (int j = 0; j < J::NM; ++j) {
      const uint64_t ident = P.m[j].ident;
      if (vh_sop_bytes(J::m_sop[j]) == 4) { for (uint64_t g = threadIdx.x; g < P.G; g += BLOCK) reinterpret_cast<uint32_t*>(lds + P.m[j].lds_off)[g] = (uint32_t)ident; }
      else { for (uint64_t g = threadIdx.x; g < P.G; g += BLOCK) reinterpret_cast<uint64_t*>(lds + P.m[j].lds_off)[g] = ident; }
    }
    for (uint64_t g = threadIdx.x; g < P.G; g += BLOCK) reinterpret_cast<uint8_t*>(lds + P.lds_present_off)[g] = 0;
    __syncthreads();
  }
  const uint64_t xoff = (MODE == VH_MODE_DENSE_GLOBAL && J::XCD) ? (uint64_t)(vh_xcc_id() % P.nxcd) * P.xcd_stride : 0;
  const typename J::Lits L(P);               // the filter's literals, decoded to their columns' types once
  unsigned long long npassed = 0, nfresh = 0;   // npassed is wave-uniform (sums of ballot population counts)
  const uint32_t spu = P.unit_rows / kStepRows;

  const uint32_t gdiv = gridDim.x / P.units_per_seg, gmod = gridDim.x % P.units_per_seg;
  uint32_t unit = blockIdx.x, seg = 0, useg = 0, ustep = 0, wave_base = 0, seg_rows = 0;
  bool have = unit < P.total_units;
  if (have) {
    seg = unit / P.units_per_seg;
    useg = unit - seg * P.units_per_seg;
    seg_rows = P.seg_rows[seg];
    wave_base = useg * P.unit_rows + wave * kWaveRows;
  }
  uint32_t v[J::NV ? J::NV : 1];
  if (have) {
    if (wave_base + kWaveRows <= seg_rows) J::template preload<true>(P, seg, wave_base + lane * kLaneStride, seg_rows, v);
    else J::template preload<false>(P, seg, wave_base + lane * kLaneStride, seg_rows, v);
  }
  uint32_t cnt = 0;
  if constexpr (J::LANES) {
    uint32_t lane_passed = 0;
    while (have) {
      const uint32_t row_l = wave_base + lane * kLaneStride;
      const bool full = wave_base + kWaveRows <= seg_rows;
      typename J::Payload Y;                      // the step's group and metric values: every load of them in flight before the filter is looked at
      if (full) J::template lanes_load<true>(P, seg, row_l, seg_rows, Y);
      else J::template lanes_load<false>(P, seg, row_l, seg_rows, Y);
      uint32_t mask = 0;
      if (full) vj_lanes_mask<J, true>(L, v, row_l, seg_rows, mask);
      else if (wave_base < seg_rows) vj_lanes_mask<J, false>(L, v, row_l, seg_rows, mask);
      lane_passed += __popc(mask);
      uint32_t nseg = seg, nwave_base = wave_base + kStepRows, nseg_rows = seg_rows;
      bool nhave = true;
      if (++ustep == spu) {
        ustep = 0;
        unit += gridDim.x;
        nhave = unit < P.total_units;
        useg += gmod; nseg = seg + gdiv;
        if (useg >= P.units_per_seg) { useg -= P.units_per_seg; ++nseg; }
        if (nhave) {
          nseg_rows = P.seg_rows[nseg];
          nwave_base = useg * P.unit_rows + wave * kWaveRows;
        }
      }
      if (nhave) {        // the next step's predicate columns travel while this step's rows go into the table
        if (nwave_base + kWaveRows <= nseg_rows) J::template preload<true>(P, nseg, nwave_base + lane * kLaneStride, nseg_rows, v);
        else J::template preload<false>(P, nseg, nwave_base + lane * kLaneStride, nseg_rows, v);
      }
      vj_lanes_rows<J>(P, seg, row_l, mask, Y, lds, xoff, nfresh, V);
      have = nhave; seg = nseg; wave_base = nwave_base; seg_rows = nseg_rows;
    }
    unsigned long long np = lane_passed;
    for (int off = 32; off > 0; off >>= 1) np += __shfl_down(np, off);
    npassed = np;                                 // (lane 0's is the wave's: it is the one that adds it to the counter below)
  }
  while (!J::LANES && have) {
    const uint32_t row_l = wave_base + lane * kLaneStride;
    const uint32_t cnt0 = cnt;
    uint32_t smask = 0u;                       // (bit-sliced: the lane's pass mask of its 32 rows; pushed below, half by half)
    if constexpr (J::SLICED) {
      if (wave_base < seg_rows) {
        smask = J::mask(L, v);
        if (wave_base + kWaveRows > seg_rows) smask &= row_l + 32u <= seg_rows ? ~0u : row_l < seg_rows ? (1u << (seg_rows - row_l)) - 1u : 0u;
      }
    } else {
      if (wave_base + kWaveRows <= seg_rows) vj_slots<J, true>(L, v, row_l, seg_rows, q, cnt);
      else if (wave_base < seg_rows) vj_slots<J, false>(L, v, row_l, seg_rows, q, cnt);
      npassed += cnt - cnt0;
    }
    // locate the next step and put its predicate columns in flight: they travel while this step's survivors are drained
    uint32_t nseg = seg, nwave_base = wave_base + kStepRows, nseg_rows = seg_rows;
    bool nhave = true;
    if (++ustep == spu) {
      ustep = 0;
      unit += gridDim.x;
      nhave = unit < P.total_units;
      useg += gmod; nseg = seg + gdiv;
      if (useg >= P.units_per_seg) { useg -= P.units_per_seg; ++nseg; }
      if (nhave) {
        nseg_rows = P.seg_rows[nseg];
        nwave_base = useg * P.unit_rows + wave * kWaveRows;
      }
    }
    if (nhave) {
      if (nwave_base + kWaveRows <= nseg_rows) J::template preload<true>(P, nseg, nwave_base + lane * kLaneStride, nseg_rows, v);
      else J::template preload<false>(P, nseg, nwave_base + lane * kLaneStride, nseg_rows, v);
    }
    __builtin_amdgcn_wave_barrier();
    const bool flush = !nhave || nseg != seg;       // queue entries are rows of the current segment
    auto drain_queue = [&](bool all) {
      if constexpr (MODE == VH_MODE_HASH && J::HPART && J::QPAY == 0 && J::ABLATE == 0 && VJ_DRAIN2) {
        while (cnt >= 128) {                        // two rows per lane while there are that many (vj_drain2)
          cnt -= 128;
          const uint32_t r0 = q[cnt + lane], r1 = q[cnt + 64 + lane];
          vj_drain2<J>(P, seg, r0, r1, true, true, lds, xoff, nfresh, V);
          __builtin_amdgcn_wave_barrier();
        }
      }
      while (cnt >= 64 || (all && cnt)) {
        const uint32_t take = cnt >= 64 ? 64u : cnt;
        cnt -= take;
        const bool act = (uint32_t)lane < take;
        const uint32_t r = act ? q[cnt + lane] : 0u;
        vj_drain<J>(P, seg, r, act, lds, xoff, nfresh, V);
        __builtin_amdgcn_wave_barrier();
      }
    };
    if constexpr (J::SLICED) {
      // two halves of the wave, 32 lanes x 32 rows each: at most 1 024 rows join the queue between two drains. By LANES, not by rows: a lane's
      // 32 rows are one 128-byte line of a 4-byte column (a record array), and a line whose survivors are all queued together is gathered by
      // one or two consecutive drains. Split by rows (16 + 16) the two halves of every line were asked for a whole drain sequence apart — with
      // half the rows passing, C5's scan fetched every payload line twice (FETCH_SIZE 2.9 GB for 1.5 GB of columns; profiles/r05/NOTES.md)
      vj_push_mask(q, cnt, lane < 32 ? smask : 0u, row_l, lane);
      npassed += cnt - cnt0;
      __builtin_amdgcn_wave_barrier();
      drain_queue(false);
      const uint32_t cnt1 = cnt;
      vj_push_mask(q, cnt, lane < 32 ? 0u : smask, row_l, lane);
      npassed += cnt - cnt1;
      __builtin_amdgcn_wave_barrier();
    }
    drain_queue(flush);
    have = nhave; seg = nseg; wave_base = nwave_base; seg_rows = nseg_rows;
    if (MODE == VH_MODE_HASH && V.H.dead) have = false;     // this wave saw the table overflow: the attempt is void (see scan_agg_kernel)
  }

  if constexpr (MODE == VH_MODE_DENSE_PART && J::PART_RING != 0)
    vh_ring_finish_tb<J::TUPLE4 ? 4 : J::TW * 8, BLOCK, VjPartDest, J::PART_RING, 2, true>(V.F, reinterpret_cast<char*>(P.tuples), (uint32_t)P.ext_stride, 31u - (uint32_t)__builtin_clz((uint32_t)P.ext_tuples),
                                                                           P.extent_missing, P.extent_part, VjPartDest(P), P.counters + 2, P.part_count);
  else if constexpr (MODE == VH_MODE_DENSE_PART) vh_part_tile_finish(P, V.T, lane);
  if constexpr (MODE == VH_MODE_HASH && J::HPART) vh_ring_finish<(J::BITSET_J >= 0 && !J::HP_PACK) ? 2 : 1, BLOCK>(V.F, reinterpret_cast<vh_u64x2*>(P.tuples2), (uint32_t)P.ext_tuples2, P.extent_missing2, P.extent_part2, VjFanDest(P), P.counters + 2);
  // the waves' counters, and how far the extents handed out by position reach, as one set of atomics per block (vh_scan_block_end)
  // (extents by position: every extent of the pool may hold something — phase 2 goes by the tags)
  vh_scan_block_end(P, npassed, nfresh, 0ull, (MODE == VH_MODE_DENSE_PART && J::PART_RING != 0) ? P.max_extents : MODE == VH_MODE_DENSE_PART ? vh_part_wave_end(P, V.W) : 0u);
  if constexpr (MODE == VH_MODE_HASH && J::LDS_HASH) vh_lds_hash_flush(P, lds, BLOCK);
  if constexpr (MODE == VH_MODE_DENSE_LDS) {
    __syncthreads();
    const uint64_t xo = J::XCD ? (uint64_t)(vh_xcc_id() % P.nxcd) * P.xcd_stride : 0;
    for (uint64_t g = threadIdx.x; g < P.G; g += BLOCK) {
      if (!reinterpret_cast<uint8_t*>(lds + P.lds_present_off)[g]) continue;
      P.present[xo + g] = 1;
#pragma unroll
      for (int j = 0; j < J::NM; ++j) {
        const uint64_t bits = vh_sop_bytes(J::m_sop[j]) == 4 ? reinterpret_cast<uint32_t*>(lds + P.m[j].lds_off)[g]
                                                             : reinterpret_cast<uint64_t*>(lds + P.m[j].lds_off)[g];
        vh_state_update<J::SCOPE>(P.m[j].state, xo + g, J::m_sop[j], bits);
      }
    }
  }
}


// ------------------------------------------------- DENSE_PART phase 2, compiled for the plan shape (`<kernel>_pagg`)
// part_agg_kernel (vh_kernels.h) with the tuple layout as constants: how many words a tuple has, where the gid ends, which word and shift
// every metric's value has, how wide it is, which state operation takes it and whether a state carries the presence flag — the pre-built
// kernel finds all of that out per launch (and, outside its two hand-written fast cases, per tuple). Same grid, same LDS layout
// (VhPlanDev::m[j].lds_off), same flush: the host launches one or the other (src/codegen/db/store.cc:131-161 is what both compute).
template <class J, int BLOCK>
__device__ __forceinline__ void vj_part_agg(const VhPlanDev& P, int blocks_per_part) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int NM = J::NM, TW = J::TW, GB = J::GID_BITS;
  constexpr bool carried = J::CARRIER >= 0;
  constexpr uint64_t gid_mask = GB ? (1ull << GB) - 1ull : 0xFFFFFFFFull;
  int part, b;
  const bool balanced = P.part_count != nullptr && P.nlevel == 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = BLOCK / 64;
  const uint64_t gpp = 1ull << P.agg_shift;
  // (the partitions' counts are asked for before the tables are filled — every range's table the same gpp entries, whichever partition this block
  // turns out to work for — and looked at behind the fill: their latency is not the first thing all blocks of the launch wait for)
  const uint32_t my_count = balanced && threadIdx.x < 64 ? vh_part_count_of(P.part_count, P.npart, (int)threadIdx.x) : 0u;
  char* mstate[NM ? NM : 1];
#pragma unroll
  for (int j = 0; j < NM; ++j) {
    mstate[j] = lds + P.m[j].lds_off;
    const uint64_t ident = P.m[j].ident;
    if (vh_sop_bytes(J::m_sop[j]) == 4) { for (uint64_t g = threadIdx.x; g < gpp; g += BLOCK) reinterpret_cast<uint32_t*>(mstate[j])[g] = (uint32_t)ident; }
    else { for (uint64_t g = threadIdx.x; g < gpp; g += BLOCK) reinterpret_cast<uint64_t*>(mstate[j])[g] = ident; }
  }
  if (!carried) for (uint64_t g = threadIdx.x; g < gpp; g += BLOCK) reinterpret_cast<uint8_t*>(lds + P.lds_present_off)[g] = 0;
  if (!vh_part_my_share(P, blocks_per_part, my_count, part, b, blocks_per_part)) return;      // (two barriers inside; blocks_per_part: from here on THIS partition's blocks — by the partitions' tuple counts where phase 1 counted them)
  const uint64_t g0 = (uint64_t)part << P.agg_shift;
  const uint64_t ng = g0 >= P.G ? 0 : (P.G - g0 < gpp ? P.G - g0 : gpp);
  __syncthreads();
  const bool two = P.nlevel == 2;
  const uint64_t g0_rel = J::TUPLE4 && two ? (uint64_t)(part & 63) << P.agg_shift : g0;      // (four-byte tuples of two levels: gids relative to the level-1 partition)
  uint32_t first = 0, total;
  if (two) {
    const uint32_t lo = P.l2[part >> 6], hi = P.l2[(part >> 6) + 1], used = P.l2[VH_L2_NEXT + (part >> 6)];
    first = lo;
    total = lo + (used < hi - lo ? used : hi - lo);
  } else {
    const unsigned long long allocated = P.counters[5];
    total = allocated < P.max_extents ? (uint32_t)allocated : P.max_extents;
  }
  const uint8_t want = (uint8_t)(two ? (part & 63) : part);
  const uint8_t* tags = two ? P.extent_part2 : P.extent_part;
  const uint16_t* missing = two ? P.extent_missing2 : P.extent_missing;
  const uint64_t* pool = two ? P.tuples2 : P.tuples;
  const uint32_t ext_tuples = (uint32_t)(two ? P.ext_tuples2 : P.ext_tuples), ext_stride = (uint32_t)(two ? P.ext_tuples2 : P.ext_stride);
  const uint32_t gsz = vh_tag_group(total - first, (uint32_t)blocks_per_part * nwaves);
  for (uint32_t c0 = first + ((uint32_t)b * nwaves + wave) * gsz; c0 < total; c0 += (uint32_t)blocks_per_part * nwaves * gsz) {
    const bool in = (uint32_t)lane < gsz && c0 + lane < total;
    const uint8_t tag = in ? tags[c0 + lane] : (uint8_t)0xFF;
    const uint32_t fill = in ? ext_tuples - missing[c0 + lane] : 0u;
    uint64_t mine = __ballot(in && tag == want && fill != 0);
    uint32_t ext = 0, valid = 0, at = 0;
    while (mine || at < valid) {
      if constexpr (J::TUPLE4) {
        // FOUR-byte tuples: a slot is 256 tuples — one 16-byte load per lane, four tuples each (a line's worth of bytes per
        // load instruction, as with the two-word tuples; one 4-byte load per lane read the pool at 1.1 TB/s). Extents hold a power of two of at
        // least 256 tuples and start on 128-byte lines: the loads are aligned and never leave the extent; places beyond `valid` are masked.
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const uint32_t* tb[VH_P2_SLOTS];
        uint32_t tn[VH_P2_SLOTS];
#pragma unroll
        for (int u = 0; u < VH_P2_SLOTS; ++u) {
          if (at >= valid && mine) {
            const int q = __builtin_ctzll(mine);
            mine &= mine - 1;
            ext = c0 + (uint32_t)q;
            valid = (uint32_t)__builtin_amdgcn_readlane((int)fill, q);
            at = 0;
          }
          if (at < valid) {
            tb[u] = reinterpret_cast<const uint32_t*>(pool) + (uint64_t)ext * ext_stride + at;
            tn[u] = valid - at < 256u ? valid - at : 256u;
            at += 256u;
          } else { tb[u] = reinterpret_cast<const uint32_t*>(pool); tn[u] = 0; }
        }
        u32x4 t4[VH_P2_SLOTS];
#pragma unroll
        for (int u = 0; u < VH_P2_SLOTS; ++u) t4[u] = (uint32_t)lane * 4u < tn[u] ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(tb[u]) + lane) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int u = 0; u < VH_P2_SLOTS; ++u) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if ((uint32_t)lane * 4u + (uint32_t)k >= tn[u]) continue;
            const uint64_t w0 = k == 0 ? t4[u].x : k == 1 ? t4[u].y : k == 2 ? t4[u].z : t4[u].w, local = (w0 & gid_mask) - g0_rel;
            if (local >= ng) continue;          // (a corrupt tuple cannot write outside the table)
            if (!carried) reinterpret_cast<uint8_t*>(lds + P.lds_present_off)[local] = 1;
#pragma unroll
            for (int j = 0; j < NM; ++j) {
              uint64_t v = w0 >> J::m_tshift[j];
              if (J::m_tbits[j]) v &= (1ull << (J::m_tbits[j] < 63 ? J::m_tbits[j] : 63)) - 1ull;
              vh_state_update<__HIP_MEMORY_SCOPE_WORKGROUP>(mstate[j], local, J::m_sop[j], v);
            }
          }
        }
        continue;
      }
      const uint64_t* sbase[VH_P2_SLOTS];
      uint32_t sn[VH_P2_SLOTS];
#pragma unroll
      for (int u = 0; u < VH_P2_SLOTS; ++u) {
        if (at >= valid && mine) {
          const int q = __builtin_ctzll(mine);
          mine &= mine - 1;
          ext = c0 + (uint32_t)q;
          valid = (uint32_t)__builtin_amdgcn_readlane((int)fill, q);
          at = 0;
        }
        if (at < valid) {
          sbase[u] = pool + ((uint64_t)ext * ext_stride + at) * TW;
          sn[u] = valid - at < 64u ? valid - at : 64u;
          at += 64u;
        } else { sbase[u] = pool; sn[u] = 0; }
      }
      uint64_t w[VH_P2_SLOTS][TW];
#pragma unroll
      for (int u = 0; u < VH_P2_SLOTS; ++u) {
        if constexpr (TW == 2) {        // both words of a tuple in one 16-byte load
          typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
          const u64x2 t2 = (uint32_t)lane < sn[u] ? __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(sbase[u]) + lane) : u64x2{~0ull, 0ull};
          w[u][0] = t2.x; w[u][1] = t2.y;
        } else {
#pragma unroll
          for (int x = 0; x < TW; ++x) w[u][x] = (uint32_t)lane < sn[u] ? __builtin_nontemporal_load(sbase[u] + (uint64_t)lane * TW + x) : (x == 0 ? ~0ull : 0ull);
        }
      }
#pragma unroll
      for (int u = 0; u < VH_P2_SLOTS; ++u) {
        const uint64_t w0 = w[u][0], local = (w0 & gid_mask) - g0;
        if (w0 == ~0ull || local >= ng) continue;          // an empty slot; (a corrupt tuple cannot write outside the table)
        if (!carried) reinterpret_cast<uint8_t*>(lds + P.lds_present_off)[local] = 1;
#pragma unroll
        for (int j = 0; j < NM; ++j) {
          uint64_t v = w[u][J::m_tword[j] < TW ? J::m_tword[j] : 0] >> J::m_tshift[j];
          if (GB && J::m_tbits[j]) v &= (1ull << (J::m_tbits[j] < 63 ? J::m_tbits[j] : 63)) - 1ull;      // (one-word tuples: never negative, the planner checked the column's minimum)
          else if (vh_sop_bytes(J::m_sop[j]) == 4) {
            v &= 0xFFFFFFFFull;
            if (vh_sop_sext(J::m_sop[j])) v = (uint64_t)(int64_t)(int32_t)v;
          }
          vh_state_update<__HIP_MEMORY_SCOPE_WORKGROUP>(mstate[j], local, J::m_sop[j], v);
        }
      }
    }
  }
  __syncthreads();
  // Sole block of the range, or one private copy of the range per block (P.nxcd == blocks_per_part: dense_merge_kernel adds
  // them up): plain stores of EVERY group, present or not. Otherwise one atomic update per present group into the shared table.
  const bool own = balanced || blocks_per_part == 1 || P.nxcd == blocks_per_part;
  const uint64_t xo = balanced || P.nxcd == blocks_per_part ? (uint64_t)b * P.xcd_stride : 0;
  for (uint64_t g = threadIdx.x; g < ng; g += BLOCK) {
    const uint8_t here = carried ? (uint8_t)(reinterpret_cast<uint64_t*>(mstate[carried ? J::CARRIER : 0])[g] != 0)
                                 : reinterpret_cast<uint8_t*>(lds + P.lds_present_off)[g];
    if (own && (balanced || blocks_per_part > 1)) { if (!carried) P.present[xo + g0 + g] = here; }
    else { if (!here) continue; if (!carried) P.present[g0 + g] = 1; }
#pragma unroll
    for (int j = 0; j < NM; ++j) {
      if (vh_sop_bytes(J::m_sop[j]) == 4) {
        const uint32_t bits = reinterpret_cast<uint32_t*>(mstate[j])[g];
        if (own) reinterpret_cast<uint32_t*>(P.m[j].state)[xo + g0 + g] = bits;
        else vh_state_update<__HIP_MEMORY_SCOPE_AGENT>(P.m[j].state, g0 + g, J::m_sop[j], bits);
      } else {
        const uint64_t bits = reinterpret_cast<uint64_t*>(mstate[j])[g];
        if (own) reinterpret_cast<uint64_t*>(P.m[j].state)[xo + g0 + g] = bits;
        else vh_state_update<__HIP_MEMORY_SCOPE_AGENT>(P.m[j].state, g0 + g, J::m_sop[j], bits);
      }
    }
  }
}

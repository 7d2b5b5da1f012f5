// vh_hpart.h — hashed partitioning of the hash path: the kernels behind the scan.
//
// Where it stands in for: `agg_map[agg_tuple.d].Update(agg_tuple.m)` on a std::unordered_map (src/codegen/query/scan.cc:174-177,242)
// and `_j |= metrics._j` + cardinality() of a bitset metric (src/codegen/db/store.cc:153-155, src/util/bitset.h:26-67) when the query
// produces TENS OF MILLIONS of groups (BASELINE.json configs[4], C5). A hash table of that size lives in no cache; every survivor costs
// it 2-5 read-modify-writes at random addresses, which the device executes at ~20 G/s whatever else is done (round 2: 312 M of them =
// 15.9 ms per 125 M rows). The classic answer is to move the DATA to the table instead: radix-partition the survivors by a hash of the
// group key until one partition's groups fit LDS, then aggregate there with LDS atomics only. Sequential traffic of 16 bytes per
// tuple and level instead of a 128-byte line read and written per update.
//
// The passes, all through the block-shared ring writer of vh_kernels.h (vh_ring_add: per digit a tuple counter and two waiting 128-byte lines in
// LDS, a tuple's number says where it goes, whole lines only; a stream's first extents by POSITION, what it holds beyond them — a hot key —
// out of its pool's shared overflow region). Tuples are written twice and read twice. (Rounds 3-5 also kept an older form — an unpartitioned
// "stream" pool behind the scan and two tiled scatter levels whose extents were handed out as they filled: written and read three times — as
// the re-run target after a hot key overflowed the positions; with the overflow regions inside the kernels it has no caller left and is gone.)
//
//   scan kernel (compiled per plan, vh_jit_body.h): a survivor becomes a TUPLE. Without a bitset metric: 16 bytes, (mixed key, payload
//     word) — the mixed key is a bijection of the packed 64-bit group key (vh_splitmix64 / vh_unmix64), so equal keys always travel
//     together and no key is ever compared through a lossy hash. With one: 32 bytes, (mixed key, payload, two ids, how many of them
//     count | "ids only") — or 16 when the values fit (PACKED tuples) —; a row with more than two ids sends further "ids only" tuples of the
//     same key. ONE stream, one trip through the levels and one insert per row in the aggregation. The scan block writes level A ITSELF
//     (vj_fan_add): 256 partitions by bits 63..56 of the mixed key, one 1024-thread block per CU sharing the 256 digits' waiting lines.
//   hp_count_kernel / hp_plan_kernel: the slices of the last pool, sized on the device from what level A holds per partition.
//   hp_ring_scatter_kernel, level B: partition a -> 256 ranges by bits 55..48, into slice a. Block a reads digit a's extents of pool a where
//     they lie (by position; the overflow region's by tag), every wave on its own, four tuples in flight per lane: no tile, no histogram,
//     no block barrier between the first load and the last store.
//   hp_aggregate_body (compiled per plan next to the scan: `<kernel>_hpagg`): 65 536 ranges of (at C5's size) ~550 groups. A block per
//     range: tuples -> open-addressing table in LDS (keys = mixed keys, 64-bit LDS compare-and-swap; states as in every other LDS table),
//     their ids -> (group slot, id) set in LDS, first sight bumping the group's cardinality; the range's groups leave straight into the
//     result's output columns, or as records for a compact list that the ordinary emission kernel reads when HAVING / top-N must see
//     them first. Ranges whose groups would not fit the tables are worked through in `passes` sub-ranges (the next bits of the mixed key).
#pragma once
#include "vh_kernels.h"

#define HP_ET 4096            // 16-byte units per extent (64 KB): 4096 plain tuples or 2048 tuples that carry ids
#define HP_FAN 256            // partitions per level
#define VH_HP_CHUNKS 8         // chunk launches of the aggregation when a big result is delivered while it is produced (a divisor of HP_FAN)

struct VhHpPool {             // extents of HP_ET 16-byte units of tuples, written through the ring writer
  uint64_t* tuples;
  uint16_t* fill;             // tuples in the extent; 0: never used
  uint8_t* tag;               // the digit whose tuples the extent holds
  uint32_t max_extents;
  uint32_t stride;            // tuples from one extent's first place to the next one's: HP_ET, or a line more (a block keeps an extent open per
                              // digit and fills them at the same pace — 64 KB apart, its stores of the moment would agree in the address bits
                              // that pick the HBM channel; VhPlanDev::ext_stride is the same remedy for DENSE_PART)
  uint32_t ovf_base;          // where the pool's shared overflow region starts (pool a; pool b's slices carry their own) ...
  unsigned long long* ovf_cursor;   // ... and how many extents of it were taken (nullptr: no such region). Found by their tags.
};
struct VhHpKind {             // the two pools of the tuples
  VhHpPool a, b;
  uint32_t* slice;            // [HP_FAN + 1] first extent of partition a's slice of pool b; [HP_FAN + 1 + a]: extents of it in use (the positional ones + the overflow extents taken)
  uint32_t* count;            // [HP_FAN] tuples per level-A digit (hp_count_kernel)
};
struct VhHpArgs {
  VhHpKind k[1];
  int32_t units;              // 16-byte units per tuple: 1, or 2 when the tuples carry the ids of a bitset metric in words of their own
  // PACKED tuples (round 4): a bitset metric's tuple in 16 bytes — word 0 the mixed key, word 1 = payload | id 0 << pk_pbits | id 1 <<
  // (pk_pbits + pk_idbits) | ids that count << 61 | "ids only" << 63 — whenever the values fit: the planner knows the metric columns' min / max
  // and the largest id of the scanned segments (refresh_stats, VhColumn::bs_maxid). Half the bytes through every stage (C5: 2 GB -> 1 GB of tuples).
  int32_t pk, pk_pbits, pk_idbits;
  int32_t passes;             // hp_aggregate_kernel: sub-ranges per range (power of two)
  int32_t gslots, sslots;     // its LDS tables: group slots (+ 1), (group slot, id) set slots (0: no bitset metric)
  uint32_t keys_off, set_off; // LDS byte offsets (metric states at VhPlanDev::m[j].lds_off)
  int32_t bitset_j;
  uint32_t chunk;             // group records a block of hp_aggregate_kernel takes from the list at a time
  uint64_t list_cap;          // records the group list holds (+ one reserved record behind them)
  uint32_t slice_levels_cap;  // (tests: see VhPlanDev::slice_levels_cap; ~0u otherwise)
  uint32_t* heavy_mark;       // = VhPlanDev::heavy_mark (nullptr: no second pass to be had): hp_plan_kernel marks all 256 ranges of a level-A partition
                              // that holds many times its share of the tuples — level B would push them through ONE block — and gives it no slice
  int32_t ablate;             // measurement only (VH_HP_ABLATE; results are wrong): 1 no id inserts, 2 no records written, 4 no tuples either, 8 no table clears
  // direct emission: the aggregation kernel writes a range's groups straight into the result's output columns (key columns in the
  // dimensions' own element types, states in the metrics') at places taken off the result's row counter — no list of group records, no
  // emission kernel behind it. Taken when no HAVING and no top-N have to look at the groups first.
  int32_t direct, ngroup;
  // STREAMED delivery (round 4): the aggregation runs as `nchunks` launches over consecutive level-A partitions, chunk c writing its groups
  // to rows [c * chunk_rows, ...) of the output columns and counting them in out_count[c]; the host copies a finished chunk's rows to
  // pinned memory while the next chunks aggregate (C5: 35 M groups leave over PCIe for longer than all kernels run). 0: one launch, one region.
  int32_t nchunks, pad_chunks;
  uint64_t chunk_rows;
  unsigned long long* out_count;
  void* out_key[VH_MAX_GROUP]; void* out_state[VH_MAX_METRIC];
  uint32_t gkey_shift[VH_MAX_GROUP], gesize[VH_MAX_GROUP], mesize[VH_MAX_METRIC];
};
// blocks hp_aggregate_kernel runs per level-A partition: enough to fill every CU's LDS twice over (a divisor of HP_FAN)
static inline int vh_hpart_bpp(int num_cu, size_t agg_lds) {
  const size_t per_cu = agg_lds + 4096 >= (size_t)(150 * 1024) ? 1 : (size_t)(150 * 1024) / (agg_lds + 4096);
  int bpp = (int)((size_t)num_cu * (per_cu < 4 ? per_cu : 4) * 2 / HP_FAN);
  if (bpp < 16) bpp = 16;      // (measured, C5: 4 blocks per partition 4.03 ms for the query's kernels, 8: 3.91, 16: 3.80, 32: 3.81 — shorter blocks even out the launch's tail)
  while (HP_FAN % bpp) --bpp;
  return bpp;
}

// The second pass over HEAVY level-A partitions reads their tuples where the scan left them in pool a (hp_heavy_tuples_kernel) instead of scanning
// the table again: what the tuples' words mean, taken from the first pass's plan.
struct VhHeavyTuples {
  const VhHpArgs* HA;           // the first pass's pools (its scratch lives until its result is freed)
  uint32_t src_blocks;          // blocks of the scan kernel that wrote level A
  int32_t units, pk, pbits, idbits;      // as VhHpArgs
  int32_t nmetric, bitset_j;    // states of the plan; which of them is the bitset's cardinality (-1: none)
  uint8_t tshift[VH_MAX_METRIC], tbytes[VH_MAX_METRIC], tsext[VH_MAX_METRIC];      // a value in the payload word: bit offset, bytes of the first pass's state (4 / 8), sign-extended
  uint32_t tbits[VH_MAX_METRIC];         // ... packed tuples: its bits (0: the state's width)
};
#ifdef VH_HPART_KERNELS        // (the kernels: vh_hpart.hip only; the host code of viya_hip.hip takes the descriptors above)
typedef uint64_t hp_u64x2 __attribute__((ext_vector_type(2)));
typedef uint32_t hp_u32x4 __attribute__((ext_vector_type(4)));

// U = 16-byte units per tuple (1 or 2). Everything below counts TUPLES: an extent holds HP_ET / U of them, a 128-byte line 8 / U.
template <int U> struct alignas(16) HpTuple { hp_u64x2 v[U]; };
template <int U> __device__ __forceinline__ HpTuple<U> hp_load_nt(const HpTuple<U>* p) {
  HpTuple<U> t;
#pragma unroll
  for (int u = 0; u < U; ++u) t.v[u] = __builtin_nontemporal_load(&p->v[u]);
  return t;
}

// ------------------------------------------------------------------ level B: no barriers, no tiles (behind the scan, which wrote level A)
// Block (a, j) of HP_FAN x NB: the tuples of partition a that scan blocks j, j + NB, ... wrote — extent k of (scan block, digit a) of pool a
// lies at k * (scan blocks * 256) + block * 256 + a, so there is nothing to list and nothing to search — go through the ring writer
// (vh_ring_add, vh_kernels.h) by bits 55..48 of the mixed key into the block's extents of slice a, also by position (hp_plan_kernel). A wave
// takes whole source extents, UNR x 64 tuples in flight per step; no tile, no histogram, no block barrier between the first load and the
// last store. C5 (62.5 M tuples): hp_scatter_kernel's level B 0.66-0.72 ms, this one see profiles/r05/NOTES.md.
struct HpRingDest {
  uint32_t lo, kmax, nb, j;
  VhRingOvf ovf;
  __device__ __forceinline__ uint64_t extent(uint32_t d, uint32_t k) const { return k < kmax ? (uint64_t)lo + ((uint64_t)k * nb + j) * HP_FAN + d : ~0ull; }
};
template <int BLOCK, int U>
__global__ __launch_bounds__(BLOCK) void hp_ring_scatter_kernel(const VhHpArgs* __restrict__ HA, uint32_t src_blocks, uint32_t nb, unsigned long long* counters) {
  typedef HpTuple<U> T;
  constexpr int UNR = 4;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  VhRing F;
  vh_ring_init<BLOCK>(lds, F, wave);
  const VhHpKind& K = HA->k[0];
  const uint32_t a = blockIdx.x / nb, j = blockIdx.x % nb;
  if (K.count[a] == 0xFFFFFFFFu) return;        // a heavy partition (hp_plan_kernel): the host's second pass takes its rows
  const uint32_t lo = K.slice[a], cap = K.slice[a + 1] - lo;
  // slice a: positional extents for the counted tuples' even spread over the (block, digit) streams, then the slice's shared overflow region
  // (hp_plan_kernel laid it out from the same count; its cursor K.slice[HP_FAN + 1 + a] starts behind the positional extents)
  const uint32_t kpos = vh_slice_levels(K.count[a], (unsigned long long)HP_FAN * nb, (uint32_t)HP_ET / (uint32_t)U, HA->slice_levels_cap);
  const HpRingDest D{lo, kpos * (uint32_t)HP_FAN * nb <= cap ? kpos : cap / ((uint32_t)HP_FAN * nb), nb, j, VhRingOvf{lo, cap, K.slice + HP_FAN + 1 + a, nullptr, K.b.fill, K.b.tag}};
  vh_u64x2* const out = reinterpret_cast<vh_u64x2*>(K.b.tuples);
  const T* const in = reinterpret_cast<const T*>(K.a.tuples);
  // the source: digit a's extents of pool a — by position for every scan block (levels below the overflow region), by tag inside the overflow
  // region (a hot digit's further extents; block j = 0 of the partition takes those)
  const uint32_t per = src_blocks * (uint32_t)HP_FAN, klev = K.a.ovf_base / per;
  const uint32_t mine = (src_blocks - j + nb - 1u) / nb;                 // scan blocks j, j + nb, ...
  uint32_t novf = 0;
  if (K.a.ovf_cursor && j == 0) { const unsigned long long c = *K.a.ovf_cursor, room = K.a.max_extents - K.a.ovf_base; novf = (uint32_t)(c < room ? c : room); }
  for (uint32_t s = (uint32_t)wave; s < klev * mine + novf; s += BLOCK / 64) {
    uint32_t e;
    if (s < klev * mine) e = (s / mine) * per + (j + (s % mine) * nb) * (uint32_t)HP_FAN + a;
    else { e = K.a.ovf_base + (s - klev * mine); if (K.a.tag[e] != (uint8_t)a) continue; }
    const uint32_t n = __builtin_amdgcn_readfirstlane((int)K.a.fill[e]);
    const T* const src = in + (uint64_t)e * K.a.stride;
    for (uint32_t i0 = 0; i0 < n; i0 += 64u * UNR) {
      T t[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) if (i0 + u * 64u + lane < n) t[u] = hp_load_nt<U>(src + i0 + u * 64u + lane);
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if (i0 + u * 64u >= n) break;                                       // (wave-uniform)
        const bool valid = i0 + u * 64u + lane < n;
        uint64_t w[2 * U];
#pragma unroll
        for (int q = 0; q < U; ++q) { w[2 * q] = t[u].v[q].x; w[2 * q + 1] = t[u].v[q].y; }
        vh_ring_add<U>(F, out, K.b.stride, valid, w, valid ? (uint32_t)(w[0] >> 48) & (HP_FAN - 1u) : 0u, lane, D, counters + 2);
      }
    }
  }
  vh_ring_finish<U, BLOCK>(F, out, K.b.stride, K.b.fill, K.b.tag, D, counters + 2);
}

// ------------------------------------------------------------------ between the levels: the slices of the last pool
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void hp_count_kernel(const VhHpArgs* __restrict__ HA) {
  __shared__ unsigned int cnt[HP_FAN];
  if (threadIdx.x < HP_FAN) cnt[threadIdx.x] = 0;
  __syncthreads();
  const VhHpKind& K = HA->k[0];
  // (every extent below the overflow region may hold something; of the region, what its cursor says was taken)
  uint32_t used = K.a.ovf_base;
  if (K.a.ovf_cursor) { const unsigned long long c = *K.a.ovf_cursor, room = K.a.max_extents - K.a.ovf_base; used += (uint32_t)(c < room ? c : room); }
  for (uint32_t e = blockIdx.x * BLOCK + threadIdx.x; e < used; e += gridDim.x * BLOCK) {
    const uint32_t f = K.a.fill[e];
    if (f) atomicAdd(&cnt[K.a.tag[e]], f);
  }
  __syncthreads();
  if (threadIdx.x < HP_FAN && cnt[threadIdx.x]) atomicAdd(K.count + threadIdx.x, cnt[threadIdx.x]);
}
// Every (block j of ring_blocks, digit) stream of partition a gets vh_slice_levels
// extents by POSITION — extent k of it is slice[a] + (k * ring_blocks + j) * HP_FAN + digit: its share of the partition's counted tuples and one
// more — and behind them the slice's shared overflow region with room for all the partition's tuples once more (a hot key inside the partition);
// the aggregation looks at every extent the slice's cursor says is used.
__global__ __launch_bounds__(HP_FAN) void hp_plan_kernel(const VhHpArgs* __restrict__ HA, unsigned long long* counters, int ring_blocks) {      // one block of HP_FAN threads
  __shared__ unsigned long long wave_tot[HP_FAN / 64];
  const VhHpKind& K = HA->k[0];
  const uint32_t et = (uint32_t)HP_ET / (uint32_t)HA->units;
  const int a = threadIdx.x, lane = a & 63, wave = a >> 6;
  // what partition a holds, in extents, + one open extent per digit of its single writer + the flush of the tails
  uint32_t c = K.count[a];
  const unsigned long long per = (unsigned long long)HP_FAN * (unsigned)(ring_blocks > 0 ? ring_blocks : 1);
  // a HEAVY partition — more than eight times the partitions' mean and more than 16 K tuples: a hot key, whose tuples all carry the same mixed key —
  // is left to the host's second pass whole (one block of level B would have to move it alone: C5h's 12.5 M tuples took half a second that way)
  if (HA->heavy_mark) {
    __shared__ unsigned long long s_sum[HP_FAN / 64];
    unsigned long long tot = c;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
    if (lane == 0) s_sum[wave] = tot;
    __syncthreads();
    tot = 0;
    for (int w = 0; w < HP_FAN / 64; ++w) tot += s_sum[w];
    if (c > 16384u && (unsigned long long)c * HP_FAN > 8ull * tot) {
      for (int q = 0; q < 8; ++q) HA->heavy_mark[a * 8 + q] = ~0u;
      atomicAdd(counters + 11, 256ull);
      atomicAdd(counters + 12, (unsigned long long)c);
      atomicAdd(counters + 13, 1ull);        // (partitions left out whole: when they account for every marked range, the second pass reads their tuples)
      c = 0;                                   // no slice: level B and the ranges' kernel find nothing of it
      K.count[a] = 0xFFFFFFFFu;                // (level B's block a: nothing to move)
    }
  }
  const unsigned long long need = !c ? 0ull : vh_slice_extents(c, per, et, HA->slice_levels_cap);
  unsigned long long incl = need;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const unsigned long long o = __shfl_up(incl, off); if (lane >= off) incl += o; }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  unsigned long long before = 0;
  for (int w = 0; w < wave; ++w) before += wave_tot[w];
  const unsigned long long at = before + incl - need, end = before + incl;
  K.slice[a] = (uint32_t)(at < K.b.max_extents ? at : K.b.max_extents);
  K.slice[HP_FAN + 1 + a] = c ? vh_slice_levels(c, per, et, HA->slice_levels_cap) * (uint32_t)per : 0u;      // (extents used so far: the positional ones; overflow extents are counted on top as they are taken)
  if (a == HP_FAN - 1) {
    K.slice[HP_FAN] = (uint32_t)(end < K.b.max_extents ? end : K.b.max_extents);
    if (end > K.b.max_extents) atomicOr(counters + 2, VH_ERR_PART_FULL);
  }
}

#define HP_IDS_ONLY 4ull    // tuples that carry ids, word 3: bits 0-1 = ids that count (0..2), bit 2 = the payload was sent with another tuple of the row
// ------------------------------------------------------------------ heavy level-A partitions: their tuples into the plain hash organisation
// P: the SECOND pass's plan (plain hash organisation: one key word, the group table and the (group, id) set in HBM). Every wave takes whole extents of
// pool a whose digit hp_plan_kernel marked heavy (K.count[a] == ~0u) — by position below the overflow region, by tag inside it —, a lane a tuple:
// the group key is the mixed key un-mixed, the values sit in the payload word where the first pass's plan put them, the ids in the tuple. What
// vh_consume does for a surviving row from here on, hot groups included (vh_hot_lanes / VhHotAcc: such a partition is one hot key and its ids).
// C5 with one (t, u) on a tenth of the rows: 12.8 M tuples of 16 bytes instead of 125 M rows through the interpreting scan.
template <int U, bool PK>
__global__ __launch_bounds__(256) void hp_heavy_tuples_kernel(const VhPlanDev P, const VhHeavyTuples A) {
  typedef HpTuple<U> T;
  __shared__ uint32_t s_heavy[HP_FAN / 32];
  const VhHpKind& K = A.HA->k[0];
  const int lane = threadIdx.x & 63;
  if (threadIdx.x < HP_FAN / 32) s_heavy[threadIdx.x] = 0;
  __syncthreads();
  if (threadIdx.x < HP_FAN && K.count[threadIdx.x] == 0xFFFFFFFFu) atomicOr(&s_heavy[threadIdx.x >> 5], 1u << (threadIdx.x & 31));
  __syncthreads();
  uint32_t used = K.a.ovf_base;
  if (K.a.ovf_cursor) { const unsigned long long c = *K.a.ovf_cursor, room = K.a.max_extents - K.a.ovf_base; used += (uint32_t)(c < room ? c : room); }
  const T* const in = reinterpret_cast<const T*>(K.a.tuples);
  const int PB = PK ? A.pbits : 0, IB = PK ? A.idbits : 32;
  const uint64_t PMASK = PK && PB < 64 ? (1ull << PB) - 1ull : ~0ull, IMASK = IB < 64 ? (1ull << IB) - 1ull : ~0ull;
  const int bj = A.bitset_j;
  int nvalue = 0;
  for (int j = 0; j < A.nmetric; ++j) nvalue += j != bj;
  VhHotAcc hot_acc{0ull, false, 0ull, {0ull, 0ull, 0ull, 0ull}};
  unsigned long long nfresh = 0, npairs = 0, ntuples = 0;
  const uint32_t wave_g = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  // the work list: slot s of a heavy digit's positional extents (extent s * HP_FAN + digit: every scan block's, every level's), digit after digit,
  // then the overflow region's extents (a heavy digit's by their tags) — dealt out to the waves round robin
  uint32_t nheavy = 0;
  for (int q = 0; q < HP_FAN / 32; ++q) nheavy += __popc(s_heavy[q]);
  const uint32_t npos = K.a.ovf_base / (uint32_t)HP_FAN;
  const uint64_t nwork = (uint64_t)nheavy * npos + (used - K.a.ovf_base);
  for (uint64_t wk = wave_g; wk < nwork; wk += nwaves) {
    uint32_t e;
    if (wk < (uint64_t)nheavy * npos) {
      uint32_t h = (uint32_t)(wk / npos), a = 0;           // the h-th heavy digit
      for (int q = 0; q < HP_FAN / 32; ++q) {
        const uint32_t c = __popc(s_heavy[q]);
        if (h < c) { uint32_t m = s_heavy[q]; for (uint32_t i = 0; i < h; ++i) m &= m - 1u; a = (uint32_t)q * 32u + (uint32_t)__builtin_ctz(m); break; }
        h -= c;
      }
      e = (uint32_t)(wk % npos) * (uint32_t)HP_FAN + a;
    } else {
      e = K.a.ovf_base + (uint32_t)(wk - (uint64_t)nheavy * npos);
      const uint32_t a = K.a.tag[e];
      if (!((s_heavy[a >> 5] >> (a & 31u)) & 1u)) continue;
    }
    const uint32_t n = __builtin_amdgcn_readfirstlane((int)K.a.fill[e]);
    if (!n) continue;
    const T* const src = in + (uint64_t)e * K.a.stride;
    for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
      bool act = i0 + lane < n;
      T tp{};
      if (act) tp = hp_load_nt<U>(src + i0 + lane);
      const uint64_t mkey = tp.v[0].x, w1 = tp.v[0].y, payload = PK ? (w1 & PMASK) : w1;
      const uint64_t meta = PK ? (w1 >> 61) : U == 2 ? tp.v[U - 1].y : 0ull;      // bits 0-1: ids that count, bit 2: ids only
      const bool vact = act && !(bj >= 0 && (meta & HP_IDS_ONLY));               // the tuple carries the row's values
      ntuples += vact ? 1 : 0;
      bool ok = true, fresh = false;
      uint64_t gid = 0;
      if (act) gid = vh_hash_insert64(P, vh_unmix64(mkey), ok, fresh);
      nfresh += act && fresh ? 1 : 0;
      if (__ballot(act && !ok)) { if (act && !ok) atomicOr(P.counters + 2, VH_ERR_HASH_FULL); }
      act = act && ok;
      // the wave's hot group, as in vh_consume
      uint64_t hot = vh_hot_lanes(act, gid);
      if (!hot && hot_acc.valid) hot = __ballot(act && gid == hot_acc.gid);
      const bool in_hot = ((hot >> lane) & 1ull) != 0;
      const bool keep = hot != 0 && nvalue <= VH_HOT_METRICS;
      const bool speaks = hot != 0 && !keep && lane == __builtin_ctzll(hot);
      if (keep) {
        const uint64_t hg = __shfl(gid, __builtin_ctzll(hot));
        if (!hot_acc.valid || hot_acc.gid != hg) {
          vh_hot_flush(P, hot_acc);
          hot_acc.gid = hg; hot_acc.valid = true; hot_acc.card = 0;
          int k = 0;
          for (int j = 0; j < A.nmetric; ++j) if (j != bj) { const int sop = P.m[j].sop(); hot_acc.v[k++] = sop == SOP_ADD32 || sop == SOP_ADD64 || sop == SOP_ADDF32 || sop == SOP_ADDF64 || sop == SOP_ADD32P ? 0ull : P.m[j].ident; }
        }
      }
      int kv = 0;
      for (int j = 0; j < A.nmetric; ++j) {
        const VhMetricDev& m = P.m[j];
        if (j == bj) {
          unsigned long long* const card = reinterpret_cast<unsigned long long*>(vh_hash_state(P, m, gid));
          unsigned long long mine = 0;
          if (act) {
            const uint64_t ids = PK ? 0ull : tp.v[U - 1].x;
            const uint32_t idv[2] = {PK ? (uint32_t)((w1 >> PB) & IMASK) : (uint32_t)ids, PK ? (uint32_t)((w1 >> (PB + IB)) & IMASK) : (uint32_t)(ids >> 32)};
            const int nid = (int)(meta & 3ull);
            const int b = m.slot();
            for (int q = 0; q < 2; ++q) {
              if (q >= nid || (q == 1 && idv[1] == idv[0])) break;
              bool sok = true, sfresh = false;
              vh_set_insert64(P.dset_keys[b], P.dset_mask[b], 4096u, (gid << 32) | idv[q], sok, sfresh);
              if (!sok) atomicOr(P.counters + 2, VH_ERR_HASH_FULL);
              mine += sfresh ? 1 : 0;
            }
            npairs += mine;
            if (mine && !in_hot) atomicAdd(card, mine);
          }
          if (hot) {
            const unsigned long long tot = vh_wave_combine(SOP_ADD64, mine, in_hot);
            if (keep) hot_acc.card += tot;
            else if (speaks && tot) atomicAdd(card, tot);
          }
          continue;
        }
        uint64_t v = payload >> A.tshift[j];
        if (PK && A.tbits[j] && A.tbits[j] < 64) v &= (1ull << A.tbits[j]) - 1ull;
        else if (A.tbytes[j] == 4) { v &= 0xFFFFFFFFull; if (A.tsext[j]) v = (uint64_t)(int64_t)(int32_t)v; }
        bool upd = vact && act;
        const bool vin = in_hot && upd;                      // (an "ids only" tuple of the hot group has no values to add)
        if (hot && __ballot(vin)) {
          const uint64_t tot = vh_wave_combine(m.sop(), v, vin);
          if (keep) { hot_acc.v[kv] = vh_combine(m.sop(), hot_acc.v[kv], tot); if (vin) upd = false; }
          else if (vin) { v = tot; upd = lane == __builtin_ctzll(__ballot(vin)); }
        }
        ++kv;
        if (upd) vh_state_update<__HIP_MEMORY_SCOPE_AGENT>(vh_hash_state(P, m, gid), 0, m.sop(), v);
      }
    }
  }
  vh_hot_flush(P, hot_acc);
  for (int off = 32; off > 0; off >>= 1) { ntuples += __shfl_down(ntuples, off); nfresh += __shfl_down(nfresh, off); npairs += __shfl_down(npairs, off); }
  vh_scan_block_end(P, ntuples, nfresh, npairs, 0u);
}

__device__ __forceinline__ void hp_store_sized(void* base, uint32_t esize, unsigned long long i, uint64_t v) {
  switch (esize) {
    case 1: reinterpret_cast<uint8_t*>(base)[i] = (uint8_t)v; break;
    case 2: reinterpret_cast<uint16_t*>(base)[i] = (uint16_t)v; break;
    case 4: reinterpret_cast<uint32_t*>(base)[i] = (uint32_t)v; break;
    default: reinterpret_cast<uint64_t*>(base)[i] = v; break;
  }
}

// ------------------------------------------------------------------ the ranges, one after the other, in LDS
// Probe sequences are DOUBLE-HASHED (the step an odd number out of other bits of the key: every slot of a power-of-two table is reached):
// a wave sits out the longest of its 64 lanes' sequences, and at 50-60 % load linear probing's clusters make that longest one ~15 probes
// where independent steps make it ~8 (`linear`: the old form, for measurement).
__device__ __forceinline__ uint32_t hp_slot(unsigned long long* keys, uint32_t gslots, uint64_t mkey, bool insert, bool& ok, bool linear = false) {
  const uint32_t mask = gslots - 1u;
  const uint32_t step = linear ? 1u : (((uint32_t)(mkey >> 20) & mask) | 1u);
  if (mkey == VH_HASH_EMPTY) {                  // the one mixed key that looks like an empty slot: the table's extra slot
    if (insert) keys[gslots] = 0ull;
    return gslots;
  }
  uint32_t slot = (uint32_t)mkey & mask;
  for (uint32_t probe = 0; probe <= mask; ++probe) {
    if (insert) {       // (no look before the compare-and-swap: most first probes of a range's tuples meet an empty slot)
      const unsigned long long seen = atomicCAS(&keys[slot], (unsigned long long)VH_HASH_EMPTY, (unsigned long long)mkey);
      if (seen == VH_HASH_EMPTY || seen == mkey) return slot;
    } else {
      const unsigned long long seen = keys[slot];
      if (seen == mkey) return slot;
      if (seen == VH_HASH_EMPTY) break;
    }
    slot = (slot + step) & mask;
  }
  ok = false;
  return 0;
}

// ... and in LOCKSTEP (HP_PROBE_UNIFORM, the compiled aggregation's form): every lane of the wave goes round the loop until the wave's last
// one is through, a lane that has its slot simply repeating its compare-and-swap there (it meets its own key: no change). The loop then
// has no divergence to keep books on — the form above spends ~25 instructions per round, most of them s_and / s_or / s_andn2 on exec
// masks, this one about half — and the rounds a wave runs are the same: those of its unluckiest lane.
#ifndef HP_PROBE_UNIFORM
#define HP_PROBE_UNIFORM 1      // (0: the divergent loops, for measurement — VH_JIT_FLAGS=-DHP_PROBE_UNIFORM=0)
#endif
__device__ __forceinline__ uint32_t hp_slot_uniform(unsigned long long* keys, uint32_t gslots, uint64_t mkey, bool active, bool& ok) {
  const uint32_t mask = gslots - 1u;
  const uint32_t step = ((uint32_t)(mkey >> 20) & mask) | 1u;
  const bool special = mkey == VH_HASH_EMPTY;        // the one mixed key that looks like an empty slot: the table's extra slot
  if (active && special) keys[gslots] = 0ull;
  const bool live = active && !special;
  uint32_t slot = (uint32_t)mkey & mask;
  ok = true;
  for (uint32_t round = 0; round <= mask; ++round) {
    unsigned long long seen = (unsigned long long)mkey;
    if (live) seen = atomicCAS(&keys[slot], (unsigned long long)VH_HASH_EMPTY, (unsigned long long)mkey);
    ok = seen == VH_HASH_EMPTY || seen == mkey;
    if (!__ballot(!ok)) break;
    slot = ok ? slot : (slot + step) & mask;
  }
  return special ? gslots : slot;
}

#define HP_OVF 128          // extents beyond the first of a range that a block remembers (skewed keys only: a range's share of
                            // uniform keys is a quarter of one extent)
#define HP_HEAVY_EXT 24     // a range of more extents than this (~100 K tuples: a hundred times a range's share) is HEAVY: left to the host's
                            // second pass over the plain hash organisation when the plan has one (VhPlanDev::heavy_mark), as is a range whose
                            // groups or ids overflow the block's LDS tables
struct HpAggLds {
  unsigned long long base, chunk_pos, chunk_end;
  uint32_t count, bad, novf, wave_tot[16];
  uint32_t rbad;                     // the range at hand overflowed its LDS tables (block-uniform after the barrier behind the tuples)
  uint16_t next[HP_FAN];             // extents of the slice per digit (heavy ranges: more than HP_HEAVY_EXT)
  uint32_t ext1[HP_FAN];             // per digit b: the range's first extent (~0u: none) ...
  uint16_t fill1[HP_FAN];            // ... and the tuples in it
  uint32_t ovf_ext[HP_OVF];          // the others: their digit in ovf_key
  uint16_t ovf_fill[HP_OVF], ovf_key[HP_OVF];
};

// grid: HP_FAN x blocks_per_partition; block (a, j) works through ranges (a, b), b = j, j + blocks_per_partition, ...
// Compiled PER PLAN SHAPE, next to the query's scan kernel (vh_jit.hip emits `viya_jit_hpagg_<hash>`, which is this body over the traits struct
// `J` of the shape): which states there are, how they are updated, where a tuple's payload fields and ids sit and what the output columns'
// element types are, are compile-time constants. Round 3's pre-built form walked the plan's metric descriptors per tuple and per group —
// counters: 651 M SALU + 308 M VALU wave instructions per C5 launch, about half the kernel's cycles in instruction issue.
//   U = 16-byte units per tuple: 1 (mixed key, payload — or the PACKED form with two ids in the payload word) or 2 (mixed key, payload, two
//   ids, how many of them count | ids only).
template <class J, int BLOCK>
__device__ __forceinline__ void hp_aggregate_body(const VhPlanDev& P, const VhHpArgs* __restrict__ HA, int blocks_per_partition, int a_first) {
  constexpr bool PK = J::HP_PACK;
  constexpr bool IDS = J::BITSET_J >= 0;         // the tuples carry ids
  constexpr int U = (IDS && !PK) ? 2 : 1;
  constexpr int NM = J::NM, NG = J::NG;
  constexpr int PB = PK ? J::HP_PBITS : 0, IB = PK ? J::HP_IDBITS : 32;
  typedef HpTuple<U> T;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  __shared__ HpAggLds S;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int a = a_first + blockIdx.x / blocks_per_partition, j0 = blockIdx.x % blocks_per_partition;
  // where this block's groups go when it writes the output columns itself: its chunk's rows and row counter (one region without chunks)
  const uint32_t ochunk = HA->nchunks ? (uint32_t)a / (uint32_t)(HP_FAN / HA->nchunks) : 0u;
  unsigned long long* const out_count = HA->out_count + ochunk;
  const unsigned long long out_cap = HA->nchunks ? HA->chunk_rows : HA->list_cap, out_base = HA->nchunks ? (unsigned long long)ochunk * HA->chunk_rows : 0ull;
  unsigned long long* const gkeys = reinterpret_cast<unsigned long long*>(lds + HA->keys_off);
  unsigned long long* const skeys = reinterpret_cast<unsigned long long*>(lds + HA->set_off);
  const uint32_t GS = (uint32_t)HA->gslots, SS = (uint32_t)HA->sslots;
  const int passes = HA->passes;
  const int sub_bits = 31 - __builtin_clz((uint32_t)passes | 1u);
  const uint32_t stride_w = P.hrec_bytes / 8u;
  const VhHpKind& K = HA->k[0];
  const uint32_t es = K.b.stride;
  const T* const pool = reinterpret_cast<const T*>(K.b.tuples);
  char* mstate[NM ? NM : 1];
#pragma unroll
  for (int j = 0; j < NM; ++j) mstate[j] = lds + P.m[j].lds_off;
  // ---- which extent of slice a holds which range: the slice's tags for all the block's ranges — first how many extents every digit has (a digit
  // with more than HP_HEAVY_EXT is a heavy range: marked for the host's second pass, not listed), then the first extent and the others of the rest
  for (int i = tid; i < HP_FAN; i += BLOCK) { S.ext1[i] = ~0u; S.next[i] = 0; }
  if (tid == 0) { S.chunk_pos = 0; S.chunk_end = 0; S.novf = 0; S.bad = 0; S.rbad = 0; }
  __syncthreads();
  const bool heavy_ok = P.heavy_mark != nullptr && passes == 1;
  auto mark_heavy = [&](uint32_t d, uint32_t extents) {      // (one thread)
    const uint32_t idx = ((uint32_t)a << 8) | d;
    atomicOr(P.heavy_mark + (idx >> 5), 1u << (idx & 31u));
    atomicAdd(P.counters + 11, 1ull);
    atomicAdd(P.counters + 12, (unsigned long long)extents * (unsigned long long)(HP_ET / U));
  };
  {
    const uint32_t lo = K.slice[a], cap = K.slice[a + 1] - lo, used = K.slice[HP_FAN + 1 + a];
    const uint32_t hi = lo + (used < cap ? used : cap);
    if (heavy_ok) {
      for (uint32_t e = lo + tid; e < hi; e += BLOCK) {
        if (!K.b.fill[e]) continue;
        const uint32_t d = K.b.tag[e];
        if ((int)(d % (uint32_t)blocks_per_partition) == j0) { const uint32_t was = atomicAdd(reinterpret_cast<uint32_t*>(S.next) + (d >> 1), (d & 1u) ? 0x10000u : 1u); (void)was; }
      }
      __syncthreads();
      for (int d = tid; d < HP_FAN; d += BLOCK) if (S.next[d] > HP_HEAVY_EXT) mark_heavy((uint32_t)d, S.next[d]);
    }
    for (uint32_t e = lo + tid; e < hi; e += BLOCK) {
      const uint32_t f = K.b.fill[e];
      if (!f) continue;
      const uint32_t d = K.b.tag[e];
      if ((int)(d % (uint32_t)blocks_per_partition) != j0) continue;          // (another block's range)
      if (heavy_ok && S.next[d] > HP_HEAVY_EXT) continue;                     // (a heavy range: the host's second pass)
      if (atomicCAS(&S.ext1[d], ~0u, e) == ~0u) S.fill1[d] = (uint16_t)f;
      else { const uint32_t at = atomicAdd(&S.novf, 1u); if (at < HP_OVF) { S.ovf_ext[at] = e; S.ovf_fill[at] = (uint16_t)f; S.ovf_key[at] = (uint16_t)d; } }
    }
  }
  __syncthreads();
  if (S.novf > HP_OVF) { if (tid == 0) atomicOr(P.counters + 2, VH_ERR_HPART_FULL); return; }     // (skew beyond what a block remembers: the plain hash table)
  const uint32_t novf = S.novf;
  // the first 1024 tuples of a range's first extent travel while the previous range is worked on
  constexpr int N = 1024 / BLOCK;
  T ng[N];
  auto prefetch = [&](int b, T (&g)[N]) {
    if (b >= HP_FAN) return;
    const uint32_t e0 = S.ext1[b], f0 = e0 == ~0u ? 0u : S.fill1[b];
#pragma unroll
    for (int u = 0; u < N; ++u) if ((uint32_t)(u * BLOCK + tid) < f0) g[u] = pool[(uint64_t)e0 * es + u * BLOCK + tid];
  };
  prefetch(j0, ng);
  const int abl = HA->ablate;
  uint32_t* const card = reinterpret_cast<uint32_t*>(mstate[IDS ? J::BITSET_J : 0]);
  const int set_shift = IDS ? 32 - (31 - __builtin_clz(SS | 1u)) : 0;
  constexpr uint64_t PMASK = PK && PB < 64 ? (1ull << PB) - 1ull : ~0ull, IMASK = (1ull << IB) - 1ull;
  for (int b = j0; b < HP_FAN; b += blocks_per_partition) {
    T cg[N];
#pragma unroll
    for (int u = 0; u < N; ++u) cg[u] = ng[u];
    const uint32_t e0 = S.ext1[b], f0 = e0 == ~0u ? 0u : S.fill1[b];
    prefetch(b + blocks_per_partition, ng);
    if (f0 == 0) continue;                       // (uniform: an empty range)
    for (int pass = 0; pass < passes; ++pass) {
      if (!(abl & 8)) {
        for (uint32_t g = tid; g <= GS; g += BLOCK) {
          gkeys[g] = VH_HASH_EMPTY;
#pragma unroll
          for (int j = 0; j < NM; ++j) {
            if (J::m_sop[j] == SOP_BITSET) reinterpret_cast<uint32_t*>(mstate[j])[g] = 0u;       // (a cardinality never exceeds the set's slots)
            else if (vh_sop_bytes(J::m_sop[j]) == 4) reinterpret_cast<uint32_t*>(mstate[j])[g] = (uint32_t)P.m[j].ident;
            else reinterpret_cast<uint64_t*>(mstate[j])[g] = P.m[j].ident;
          }
        }
        if (IDS) for (uint32_t g = tid; g < SS; g += BLOCK) skeys[g] = VH_HASH_EMPTY;
      }
      __syncthreads();
      bool bad = false;
      if (abl & 4) {      // (the tuples are still looked at)
        uint64_t acc = 0;
        for (int u = 0; u < N; ++u) acc += cg[u].v[0].x + cg[u].v[U - 1].y;
        if (acc == 0x123456789ABCDEFull) P.counters[7] = acc;
        continue;
      }
      // ---- a tuple: the group's slot (claimed if new), the metric values of its payload word, then its ids into the (group slot, id) set
      // (act: the lane has a tuple. With HP_PROBE_UNIFORM the whole wave calls, lanes without one ride along.)
      auto tuple = [&](const T& tp, bool act) {
        const uint64_t mkey = tp.v[0].x, w1 = tp.v[0].y, payload = PK ? (w1 & PMASK) : w1;
        if (sub_bits && (int)((uint32_t)(mkey >> (48 - sub_bits)) & (uint32_t)(passes - 1)) != pass) act = false;
        if (!HP_PROBE_UNIFORM && !act) return;
        bool ok = true;
        const uint32_t slot = HP_PROBE_UNIFORM ? hp_slot_uniform(gkeys, GS, mkey, act, ok) : hp_slot(gkeys, GS, mkey, true, ok, (abl & 32) != 0);
        if (act && !ok) bad = true;
        act = act && ok;
        if (!HP_PROBE_UNIFORM && !act) return;
        const uint64_t meta = PK ? (w1 >> 61) : U == 2 ? tp.v[U - 1].y : 0ull;      // bits 0-1: ids that count, bit 2: ids only
        if (act && (!IDS || !(meta & HP_IDS_ONLY))) {
#pragma unroll
          for (int j = 0; j < NM; ++j) {
            if (J::m_sop[j] == SOP_BITSET) continue;
            uint64_t v = payload >> J::m_tshift[j];
            if (PK && J::m_tbits[j] && J::m_tbits[j] < 64) v &= (1ull << J::m_tbits[j]) - 1ull;      // (packed values are never negative: the planner checked the column's minimum)
            else if (vh_sop_bytes(J::m_sop[j]) == 4) { v &= 0xFFFFFFFFull; if (vh_sop_sext(J::m_sop[j])) v = (uint64_t)(int64_t)(int32_t)v; }
            vh_state_update<__HIP_MEMORY_SCOPE_WORKGROUP>(mstate[j], slot, J::m_sop[j], v);
          }
        }
        if (IDS && !(abl & 1)) {
          const uint64_t ids = PK ? 0ull : tp.v[U - 1].x;
          const uint32_t idv[2] = {PK ? (uint32_t)((w1 >> PB) & IMASK) : (uint32_t)ids, PK ? (uint32_t)((w1 >> (PB + IB)) & IMASK) : (uint32_t)(ids >> 32)};
          const int n = act ? (int)(meta & 3ull) : 0;
          if constexpr (HP_PROBE_UNIFORM) {
            // both ids' probe sequences in ONE lockstep loop: two independent compare-and-swaps in flight per round
            unsigned long long key[2];
            uint32_t at[2], sstep[2];
            bool live[2], okq[2], first[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              key[q] = ((unsigned long long)slot << 32) | idv[q];
              const uint32_t hq = (idv[q] ^ (slot * 0x9E3779B1u)) * 0x85EBCA6Bu;
              at[q] = hq >> set_shift; sstep[q] = (hq & (SS - 1u)) | 1u;
              live[q] = q < n; okq[q] = true; first[q] = false;
            }
            for (uint32_t round = 0; round < SS; ++round) {
#pragma unroll
              for (int q = 0; q < 2; ++q) {
                unsigned long long seen = key[q];
                if (live[q]) seen = atomicCAS(&skeys[at[q]], (unsigned long long)VH_HASH_EMPTY, key[q]);
                first[q] = first[q] || seen == VH_HASH_EMPTY;      // (only a live lane ever sees an empty slot; its repeats meet its own key)
                okq[q] = seen == VH_HASH_EMPTY || seen == key[q];
              }
              if (!__ballot(!(okq[0] && okq[1]))) break;
#pragma unroll
              for (int q = 0; q < 2; ++q) at[q] = okq[q] ? at[q] : (at[q] + sstep[q]) & (SS - 1u);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              if (first[q]) atomicAdd(&card[slot], 1u);
              if (!okq[q]) bad = true;
            }
          } else
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            if (q >= n) break;
            const unsigned long long key = ((unsigned long long)slot << 32) | idv[q];
            const uint32_t hq = (idv[q] ^ (slot * 0x9E3779B1u)) * 0x85EBCA6Bu;
            uint32_t at = hq >> set_shift;                                                    // (multiply-shift: the top bits)
            const uint32_t sstep = (abl & 32) ? 1u : ((hq & (SS - 1u)) | 1u);                 // (double hashing: see hp_slot)
            bool placed = false;
            for (uint32_t probe = 0; probe < SS; ++probe) {
              const unsigned long long seen = atomicCAS(&skeys[at], (unsigned long long)VH_HASH_EMPTY, key);
              if (seen == VH_HASH_EMPTY) { atomicAdd(&card[slot], 1u); placed = true; break; }
              if (seen == key) { placed = true; break; }
              at = (at + sstep) & (SS - 1u);
            }
            if (!placed) bad = true;
          }
        }
      };
      // (every loop below runs the same number of times for every thread of a wave: lanes without a tuple call with act = false)
#pragma unroll
      for (int u = 0; u < N; ++u) {
        if ((uint32_t)(u * BLOCK + (tid & ~63)) >= f0) break;
        const bool act = (uint32_t)(u * BLOCK + tid) < f0;
        tuple(pass == 0 || !act ? cg[u] : pool[(uint64_t)e0 * es + u * BLOCK + tid], act);
      }
      for (uint32_t i0 = N * BLOCK + (tid & ~63); i0 < f0; i0 += BLOCK) {       // (a first extent of more than 1024 tuples)
        const bool act = i0 + lane < f0;
        tuple(pool[(uint64_t)e0 * es + (act ? i0 + lane : 0u)], act);
      }
      for (uint32_t x = 0; x < novf; ++x)
        if (S.ovf_key[x] == (uint16_t)b)
          for (uint32_t i0 = (tid & ~63); i0 < S.ovf_fill[x]; i0 += BLOCK) {
            const bool act = i0 + lane < S.ovf_fill[x];
            tuple(pool[(uint64_t)S.ovf_ext[x] * es + (act ? i0 + lane : 0u)], act);
          }
      if (__ballot(bad)) { if (lane == 0) S.rbad = 1; }
      __syncthreads();
      if (S.rbad) {       // (block-uniform) the range's groups or ids did not fit the LDS tables
        __syncthreads();
        if (tid == 0) {
          S.rbad = 0;
          if (heavy_ok) mark_heavy((uint32_t)b, (uint32_t)S.next[b] ? S.next[b] : 1u);      // ... the host's second pass takes the range; nothing of it is emitted here
          else S.bad = 1;
        }
        __syncthreads();
        if (heavy_ok) continue;
      }
      // ---- this pass's groups: count, take places (off the result's row counter, or a piece of the block's chunk of the list), write
      uint32_t mine = 0;
      for (uint32_t g = tid; g <= GS; g += BLOCK) mine += gkeys[g] != VH_HASH_EMPTY;
      uint32_t incl = mine;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(incl, off); if (lane >= off) incl += o; }
      if (lane == 63) S.wave_tot[wave] = incl;
      __syncthreads();
      uint32_t tot = 0, before = 0;
#pragma unroll
      for (int w = 0; w < BLOCK / 64; ++w) { const uint32_t c = S.wave_tot[w]; tot += c; if (w < wave) before += c; }
      if (HA->direct && !(abl & 2)) {      // (uniform) the groups go straight into the result's output columns
        if (tid == 0 && tot) S.base = atomicAdd(out_count, (unsigned long long)tot);
        __syncthreads();
        unsigned long long at = out_base + S.base + before + (incl - mine);
        if (tot && S.base + tot <= out_cap) {
          for (uint32_t g = tid; g <= GS; g += BLOCK) {
            const unsigned long long mk = gkeys[g];
            if (mk == VH_HASH_EMPTY) continue;
            const unsigned long long key = vh_unmix64(g == GS ? VH_HASH_EMPTY : mk);
#pragma unroll
            for (int c = 0; c < NG; ++c) hp_store_sized(HA->out_key[c], (uint32_t)J::g_esize[c], at, key >> J::g_key_shift[c]);
#pragma unroll
            for (int j = 0; j < NM; ++j) {
              const uint64_t bits = (J::m_sop[j] == SOP_BITSET || vh_sop_bytes(J::m_sop[j]) == 4) ? (uint64_t)reinterpret_cast<const uint32_t*>(mstate[j])[g]
                                                                                                   : reinterpret_cast<const uint64_t*>(mstate[j])[g];
              hp_store_sized(HA->out_state[j], (uint32_t)J::m_esize[j], at, bits);
            }
            ++at;
          }
        } else if (tot && tid == 0) atomicOr(P.counters + 2, VH_ERR_PART_FULL);
        __syncthreads();
        continue;
      }
      if (tot && S.chunk_pos + tot > S.chunk_end) {       // (uniform) a new chunk of the list: what is left of the old one is marked empty
        for (unsigned long long i = S.chunk_pos + tid; i < S.chunk_end && i < HA->list_cap; i += BLOCK) P.hkeys[i * stride_w] = VH_HASH_EMPTY;      // (a chunk taken beyond the list's end — the attempt is void — is not written to)
        __syncthreads();
        if (tid == 0) {
          const unsigned long long want = tot > HA->chunk ? tot : HA->chunk;
          const unsigned long long got = atomicAdd(P.counters + 1, want);
          S.chunk_pos = got; S.chunk_end = got + want;
        }
        __syncthreads();
      }
      const unsigned long long base = S.chunk_pos;
      unsigned long long at = base + before + (incl - mine);
      if (abl & 2) {
      } else if (base + tot <= HA->list_cap) {
        for (uint32_t g = tid; g <= GS; g += BLOCK) {
          const unsigned long long mk = gkeys[g];
          if (mk == VH_HASH_EMPTY) continue;
          const unsigned long long key = vh_unmix64(g == GS ? VH_HASH_EMPTY : mk);
          P.hkeys[at * stride_w] = key;     // (a group whose KEY is the empty marker is told apart by the list's last, reserved record)
#pragma unroll
          for (int j = 0; j < NM; ++j) {
            char* dstp = vh_hash_state(P, P.m[j], key == VH_HASH_EMPTY ? HA->list_cap : at);
            if (J::m_sop[j] == SOP_BITSET) *reinterpret_cast<uint64_t*>(dstp) = reinterpret_cast<const uint32_t*>(mstate[j])[g];
            else if (vh_sop_bytes(J::m_sop[j]) == 4) *reinterpret_cast<uint32_t*>(dstp) = reinterpret_cast<const uint32_t*>(mstate[j])[g];
            else *reinterpret_cast<uint64_t*>(dstp) = reinterpret_cast<const uint64_t*>(mstate[j])[g];
          }
          if (key == VH_HASH_EMPTY) atomicOr(P.counters + 3, 1ull);
          ++at;
        }
      } else if (tid == 0) atomicOr(P.counters + 2, VH_ERR_PART_FULL);
      __syncthreads();
      if (tid == 0) S.chunk_pos = base + tot;
    }
  }
  __syncthreads();
  if (tid == 0 && S.bad) atomicOr(P.counters + 2, VH_ERR_HPART_FULL);
  for (unsigned long long i = S.chunk_pos + tid; i < S.chunk_end && i < HA->list_cap; i += BLOCK) P.hkeys[i * stride_w] = VH_HASH_EMPTY;
}
#endif  // VH_HPART_KERNELS

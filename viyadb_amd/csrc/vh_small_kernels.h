// vh_small_kernels.h — the non-hot kernels: table finalisation, group emission, synthetic data,
// segment stats, bandwidth probe. Included by viya_hip.hip only (non-template __global__ symbols).
#pragma once
#include "vh_kernels.h"

// ----------------------------------------------------------- table finalisation
// Combine the per-XCD private copies of a dense table into copy 0 (they were only ever
// touched through their own XCD's L2; the kernel boundary made them visible). (vh_combine: vh_kernels.h)
struct VhMergeArgs {
  int32_t nmetric; int32_t nxcd;
  int32_t present_carrier; int32_t pad;
  uint64_t G; uint64_t xcd_stride;
  uint8_t* present;
  void* state[VH_MAX_METRIC];
  uint8_t sop[VH_MAX_METRIC];
  // phase 2 of DENSE_PART with its blocks shared out by the partitions' tuple counts: group g belongs to partition g >> agg_shift, whose first
  // vh_part_shares(...) copies were written by this query (the others hold whatever an earlier query left there)
  const uint32_t* part_count; int32_t npart, agg_shift; uint32_t blocks;
};

__device__ __forceinline__ void vh_merge_group(const VhMergeArgs& A, uint64_t g, int ncopies = -1) {
  const int nx = ncopies >= 0 ? ncopies : A.nxcd;
  uint8_t p;
  if (A.present_carrier >= 0) {
    const uint64_t* s = reinterpret_cast<const uint64_t*>(A.state[A.present_carrier]);
    uint64_t any = 0;
    for (int x = 0; x < nx; ++x) any |= s[x * A.xcd_stride + g];
    p = any != 0;
  } else {
    p = A.present[g];
    for (int x = 1; x < nx; ++x) p |= A.present[x * A.xcd_stride + g];
    A.present[g] = p;
  }
  if (!p) return;
  for (int j = 0; j < A.nmetric; ++j) {
    const int sop = A.sop[j];
    if (vh_sop_bytes(sop) == 4) {
      uint32_t* s = reinterpret_cast<uint32_t*>(A.state[j]);
      uint64_t a = s[g];
      for (int x = 1; x < nx; ++x) a = vh_combine(sop, a, s[x * A.xcd_stride + g]);
      s[g] = (uint32_t)a;
    } else {
      uint64_t* s = reinterpret_cast<uint64_t*>(A.state[j]);
      uint64_t a = s[g];
      for (int x = 1; x < nx; ++x) a = vh_combine(sop, a, s[x * A.xcd_stride + g]);
      s[g] = a;
    }
  }
}
__global__ __launch_bounds__(256) void dense_merge_kernel(const VhMergeArgs A) {
  const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (A.part_count) {
    __shared__ uint32_t s_share[64];
    if (threadIdx.x < 64) { uint32_t start; s_share[threadIdx.x] = vh_part_shares(vh_part_count_of(A.part_count, A.npart, (int)threadIdx.x), A.npart, A.blocks, (uint32_t)A.nxcd, (int)threadIdx.x, &start); }
    __syncthreads();
    if (g < A.G) vh_merge_group(A, g, (int)s_share[g >> A.agg_shift]);
    return;
  }
  if (g < A.G) vh_merge_group(A, g);
}

// Emit one (key columns, metric states) row per existing group into dense output arrays,
// already in each column's own element type.
struct VhEmitArgs {
  int32_t mode;  // VH_MODE_*
  int32_t ngroup; int32_t nmetric; int32_t key_words;
  uint64_t n;    // dense: G; hash: capacity + 1
  uint32_t hstride;               // hash: u64 words between the keys of consecutive slots (key_words, or the record size)
  uint32_t state_stride[VH_MAX_METRIC];   // bytes between consecutive entries' states (the state's size, or the record size)
  int32_t present_carrier; int32_t pad;
  const uint8_t* present;
  const uint64_t* hkeys; const uint32_t* htags;
  const unsigned long long* counters;
  unsigned long long* out_count;
  // Key decode parameters as dword-aligned arrays of their own, NOT VhGroupDev: with the byte-sized members of
  // that struct next to the 64-bit ones, hipcc (ROCm 7.2) formed `s_load_dwordx2 sN, s[&g.type()], 0x5e` —
  // a scalar load off a base that is 2 mod 4. SMEM ignores the low address bits of the base, so lo / extent /
  // stride were read 2 bytes early (tests/test_isa_hazards.py greps the disassembly for that pattern).
  uint64_t glo[VH_MAX_GROUP], gextent[VH_MAX_GROUP], gstride[VH_MAX_GROUP];
  uint32_t gtype[VH_MAX_GROUP], gkey_word[VH_MAX_GROUP], gkey_shift[VH_MAX_GROUP];
  void* out_key[VH_MAX_GROUP];
  const void* state[VH_MAX_METRIC];
  void* out_state[VH_MAX_METRIC];
  uint32_t sop[VH_MAX_METRIC];
  uint32_t mtype[VH_MAX_METRIC];  // output element type of metric j
  // HAVING pushed down: the reference evaluates it on the aggregated tuples, one group at a
  // time, with ComparisonBuilder semantics (src/codegen/query/post_agg.cc:77-83); `slot` of a node is the
  // result column: < ngroup = key column, else metric (AVG compares its raw sum, bitset its cardinality).
  int32_t nhaving; int32_t pad2;
  unsigned long long* total_groups;   // groups before HAVING = agg_map.size()
  VhProgOp hprog[VH_MAX_HAVING];
  const unsigned long long* n_dev;    // != nullptr: the hash table is a compact list, entries [0, *n_dev) are all groups (hashed partitioning)
  uint32_t htype[VH_MAX_HAVING];      // element type the comparison happens in
  uint64_t hlits[VH_MAX_HAVING_LITS];
};

__device__ __forceinline__ bool vh_cmp_bits(int type, uint64_t a, uint64_t b, int op) {
  int r;  // -1 / 0 / +1 / 2 (unordered)
#define VH_C(T) { const T x = vh_lit<T>(a), y = vh_lit<T>(b); r = x < y ? -1 : (x > y ? 1 : (x == y ? 0 : 2)); }
  switch (type) {
    case VH_U8: VH_C(uint8_t) break;   case VH_U16: VH_C(uint16_t) break;
    case VH_U32: VH_C(uint32_t) break; case VH_U64: VH_C(uint64_t) break;
    case VH_I8: VH_C(int8_t) break;    case VH_I16: VH_C(int16_t) break;
    case VH_I32: VH_C(int32_t) break;  case VH_I64: VH_C(int64_t) break;
    case VH_F32: VH_C(float) break;    default: VH_C(double) break;
  }
#undef VH_C
  switch (op) {
    case VH_OP_EQ: return r == 0;
    case VH_OP_NE: return r != 0;
    case VH_OP_LT: return r == -1;
    case VH_OP_LE: return r == -1 || r == 0;
    case VH_OP_GT: return r == 1;
    default: return r == 1 || r == 0;
  }
}

__device__ __forceinline__ void vh_store_elem(void* base, int type, uint64_t idx, uint64_t bits) {
  switch (type) {
    case VH_U8: case VH_I8: reinterpret_cast<uint8_t*>(base)[idx] = (uint8_t)bits; break;
    case VH_U16: case VH_I16: reinterpret_cast<uint16_t*>(base)[idx] = (uint16_t)bits; break;
    case VH_U32: case VH_I32: case VH_F32: reinterpret_cast<uint32_t*>(base)[idx] = (uint32_t)bits; break;
    default: reinterpret_cast<uint64_t*>(base)[idx] = bits; break;
  }
}

// value of group column c / state of metric j for table entry i
__device__ __forceinline__ uint64_t vh_emit_key(const VhEmitArgs& A, uint64_t i, int c) {
  if (A.mode == VH_MODE_HASH) {
    uint64_t w = A.hkeys[i * A.hstride + A.gkey_word[c]];
    if (A.key_words == 1 && i + 1 == A.n) w = VH_HASH_EMPTY;
    return w >> A.gkey_shift[c];
  }
  return A.glo[c] + (i / A.gstride[c]) % A.gextent[c];
}
__device__ __forceinline__ uint64_t vh_emit_state(const VhEmitArgs& A, uint64_t i, int j) {
  const char* p = static_cast<const char*>(A.state[j]) + i * A.state_stride[j];
  return vh_sop_bytes(A.sop[j]) == 4 ? *reinterpret_cast<const uint32_t*>(p) : *reinterpret_cast<const uint64_t*>(p);
}

// does table entry i hold a group that passes the (optional) HAVING?
__device__ __forceinline__ bool vh_emit_have(const VhEmitArgs& A, uint64_t i, bool& present) {
  bool have = false;
  if (i < A.n) {
    if (A.mode == VH_MODE_HASH && A.n_dev) {       // a list of group records, handed out in chunks whose unused tails hold the empty marker; the
                                                   // group whose key IS the marker lives in the last, reserved record, like in the table proper
      if (i + 1 == A.n) have = A.counters[3] != 0;
      else have = i < *A.n_dev && A.hkeys[i * A.hstride] != VH_HASH_EMPTY;
    }
    else if (A.mode == VH_MODE_HASH) {
      if (i + 1 == A.n) have = A.key_words == 1 && A.counters[3] != 0;  // reserved slot
      else have = A.key_words == 1 ? A.hkeys[i * A.hstride] != VH_HASH_EMPTY : A.htags[i] == 2u;
    } else if (A.present_carrier >= 0) {
      have = reinterpret_cast<const uint64_t*>(A.state[A.present_carrier])[i] != 0;
    } else {
      have = A.present[i] != 0;
    }
  }
  present = have;
  if (have && A.nhaving) {   // postfix, bitwise & / | like the filter
    bool st[VH_MAX_STACK];
    int sp = 0;
    for (int pc = 0; pc < A.nhaving; ++pc) {
      const VhProgOp o = A.hprog[pc];
      if (o.kind() == VH_F_TRUE) st[sp++] = true;
      else if (o.kind() == VH_F_AND || o.kind() == VH_F_OR) {
        bool a = st[--sp];
        for (int k = 1; k < o.count(); ++k) { const bool b = st[--sp]; a = o.kind() == VH_F_AND ? (a & b) : (a | b); }
        st[sp++] = a;
      } else {
        const uint64_t v = o.slot() < A.ngroup ? vh_emit_key(A, i, o.slot()) : vh_emit_state(A, i, o.slot() - A.ngroup);
        bool r;
        if (o.kind() == VH_F_REL) r = vh_cmp_bits(A.htype[pc], v, A.hlits[o.lit()], o.op());
        else {
          r = !o.op();
          for (int k = 0; k < o.count(); ++k) {
            const bool e = vh_cmp_bits(A.htype[pc], v, A.hlits[o.lit() + k], o.op() ? VH_OP_EQ : VH_OP_NE);
            r = o.op() ? (r | e) : (r & e);
          }
        }
        st[sp++] = r;
      }
    }
    have = st[0];
  }
  return have;
}

// Compacted emission of the groups. A wave covers VH_EMIT_SPAN x 64 consecutive table entries and takes ONE output
// range for all of them: a position atomic per 64 entries (the first version) is a returning atomic on a single address
// and serialises at ~20 ns each — 26 ms for a 64 M-slot hash table, five times the scan that filled it.
// VH_EMIT_SPAN x 64 table entries per wave and ONE position atomic for all of them: the atomic is a returning one on a single
// address (~20 ns each; 64 M-slot hash tables used to spend 26 ms there with one per 64 entries). Small tables take SPAN = 2
// instead: 100 K entries are then 780 waves instead of 98, which is what keeps the PCIe writes of a direct emission in flight.
template <int VH_EMIT_SPAN>
__global__ __launch_bounds__(256) void emit_groups_kernel(const VhEmitArgs A) {
  const int lane = threadIdx.x & 63;
  const uint64_t wave_first = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (VH_EMIT_SPAN * 64ull);
  if (wave_first >= A.n) return;
  uint64_t keep[VH_EMIT_SPAN];
  uint32_t total = 0, seen = 0;
#pragma unroll
  for (int k = 0; k < VH_EMIT_SPAN; ++k) {
    bool present;
    const bool have = vh_emit_have(A, wave_first + (uint64_t)k * 64 + lane, present);
    keep[k] = __ballot(have);
    total += __popcll(keep[k]);
    seen += __popcll(__ballot(present));
  }
  if (A.nhaving && lane == 0 && seen) atomicAdd(A.total_groups, (unsigned long long)seen);
  if (total == 0) return;
  unsigned long long base = 0;
  if (lane == 0) base = atomicAdd(A.out_count, (unsigned long long)total);
  base = __shfl(base, 0);
#pragma unroll
  for (int k = 0; k < VH_EMIT_SPAN; ++k) {
    const uint64_t bal = keep[k];
    if ((bal >> lane) & 1ull) {
      const uint64_t i = wave_first + (uint64_t)k * 64 + lane;
      const uint64_t pos = base + __popcll(bal & ((1ull << lane) - 1ull));
      for (int c = 0; c < A.ngroup; ++c) vh_store_elem(A.out_key[c], A.gtype[c], pos, vh_emit_key(A, i, c));
      for (int j = 0; j < A.nmetric; ++j) vh_store_elem(A.out_state[j], A.mtype[j], pos, vh_emit_state(A, i, j));
    }
    base += __popcll(bal);
  }
}

// The whole tail of a SMALL dense query in one single-block launch: the private copies added up (M.nxcd > 1), the groups emitted (HAVING
// included) straight into the pinned host buffer, the 512-byte header behind them. As three launches (dense_merge_kernel, emit_groups_kernel,
// publish_header_kernel) it was 5 + 5 + 4 us and a gap behind C1's 20 us scan. The host reads nothing before the event behind this kernel.
#define VH_SMALL_TAIL_MAX 8192      // table entries
#define VH_SMALL_TAIL_STATES 65536  // ... and states read by the one block (entries x private copies x metrics)
__global__ __launch_bounds__(1024) void small_tail_kernel(const VhMergeArgs M, const VhEmitArgs A, unsigned long long* head, const unsigned long long* dev_head) {
  __shared__ uint32_t s_keep[16], s_seen[16];
  __shared__ unsigned long long s_total, s_seen_total;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (M.nxcd > 1) {
    for (uint64_t g = tid; g < M.G; g += 1024) vh_merge_group(M, g);
    __syncthreads();
  }
  unsigned long long run = 0, seen_run = 0;
  for (uint64_t base = 0; base < A.n; base += 1024) {
    const uint64_t i = base + tid;
    bool present;
    const bool have = vh_emit_have(A, i, present);
    const uint64_t bal = __ballot(have), sbal = __ballot(present);
    if (lane == 0) { s_keep[wave] = (uint32_t)__popcll(bal); s_seen[wave] = (uint32_t)__popcll(sbal); }
    __syncthreads();
    uint32_t before = 0, total = 0, seen = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const uint32_t c = s_keep[w]; total += c; seen += s_seen[w]; if (w < wave) before += c; }
    if (have) {
      const uint64_t pos = run + before + __popcll(bal & ((1ull << lane) - 1ull));
      for (int c = 0; c < A.ngroup; ++c) vh_store_elem(A.out_key[c], A.gtype[c], pos, vh_emit_key(A, i, c));
      for (int j = 0; j < A.nmetric; ++j) vh_store_elem(A.out_state[j], A.mtype[j], pos, vh_emit_state(A, i, j));
    }
    run += total; seen_run += seen;
    __syncthreads();
  }
  if (tid == 0) {
    s_total = run; s_seen_total = seen_run;
    *A.out_count = run;                                   // (the device's own copies: whoever looks at the result's scratch later finds what the three launches left there)
    if (A.nhaving) *A.total_groups = seen_run;
  }
  __syncthreads();
  if (tid < 64) {
    unsigned long long v = dev_head[tid];
    if (dev_head + tid == A.out_count) v = s_total;
    if (A.nhaving && dev_head + tid == A.total_groups) v = s_seen_total;
    head[tid] = v;
  }
}

__device__ __forceinline__ uint64_t vh_load_sized(const void* base, uint32_t esize, uint64_t i) {
  switch (esize) {
    case 1: return reinterpret_cast<const uint8_t*>(base)[i];
    case 2: return reinterpret_cast<const uint16_t*>(base)[i];
    case 4: return reinterpret_cast<const uint32_t*>(base)[i];
    default: return reinterpret_cast<const uint64_t*>(base)[i];
  }
}
__device__ __forceinline__ void vh_store_sized(void* base, uint32_t esize, uint64_t i, uint64_t v) {
  switch (esize) {
    case 1: reinterpret_cast<uint8_t*>(base)[i] = (uint8_t)v; break;
    case 2: reinterpret_cast<uint16_t*>(base)[i] = (uint16_t)v; break;
    case 4: reinterpret_cast<uint32_t*>(base)[i] = (uint32_t)v; break;
    default: reinterpret_cast<uint64_t*>(base)[i] = v; break;
  }
}

// ------------------------------------------------- device top-N over emitted groups
// `sort` + `limit` on a numeric column: instead of shipping every group to the host to be formatted and string-
// sorted (src/codegen/query/post_agg.cc:50-147, sort.cc:24-75 — what dominates at ~10 M groups), the device keeps
// a SUPERSET of the rows the reference would return: every group whose primary sort key is at least the K-th
// best (K = skip + limit), ties and a rounding slack included. The host then runs the reference's exact string
// comparators on those few rows. Keys only have to be MONOTONE in the reference's order, not exact:
//   INTEGER columns compare as strings by (length, lexicographic) (src/util/string.h:28-49): ascending order is
//     0..9, -1..-9, 10..99, -10..-99, ...  ->  key = class(digits, sign) : |v|
//   FLOAT columns compare stod("%.15g" / "%g" text): numeric order up to the rounding of the formatter, which the
//     slack covers.
enum { VH_TOPK_INT = 0, VH_TOPK_FLOAT = 1 };
struct VhTopkState { unsigned long long k_remaining, prefix, out_count, pad; unsigned long long hist[256]; };

__device__ __forceinline__ uint64_t vh_topk_key(int cls, int elem, uint64_t bits) {
  if (cls == VH_TOPK_FLOAT) {
    if (elem == VH_F32) {
      uint32_t b = (uint32_t)bits;
      if (b == 0x80000000u) b = 0;                                  // "-0" and "0" compare equal through stod
      b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
      return (uint64_t)b << 32;
    }
    uint64_t b = bits;
    if (b == 0x8000000000000000ull) b = 0;
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
  }
  bool neg = false;
  uint64_t mag = bits;
  switch (elem) {
    case VH_I8: { const int64_t v = (int8_t)bits; neg = v < 0; mag = neg ? (uint64_t)(-v) : (uint64_t)v; } break;
    case VH_I16: { const int64_t v = (int16_t)bits; neg = v < 0; mag = neg ? (uint64_t)(-v) : (uint64_t)v; } break;
    case VH_I32: { const int64_t v = (int32_t)bits; neg = v < 0; mag = neg ? (uint64_t)(-v) : (uint64_t)v; } break;
    case VH_I64: { const int64_t v = (int64_t)bits; neg = v < 0; mag = neg ? 0ull - (uint64_t)v : (uint64_t)v; } break;
    case VH_U8: mag = bits & 0xFFull; break;
    case VH_U16: mag = bits & 0xFFFFull; break;
    case VH_U32: mag = bits & 0xFFFFFFFFull; break;
    default: break;
  }
  int nd = 1;                                                        // decimal digits of |v|
  uint64_t base = 1;                                                 // 10^(nd-1)
  while (nd < 20 && mag / 10 >= base) { base *= 10; ++nd; }
  const uint64_t cls_rank = neg ? 2ull * nd + 1 : 2ull * nd;       // string length, '-' sorts before digits
  // 58 bits for the magnitude: exact up to 17 digits; the 18..20-digit classes keep (|v| - 10^(nd-1)) >> 6,
  // still monotone inside the class (the low bits only merge near-ties, which a superset tolerates)
  return (cls_rank << 58) | (nd <= 17 ? mag : (mag - base) >> 6);
}

__global__ __launch_bounds__(256) void topk_keys_kernel(const void* src, int elem, uint32_t esize, int cls, int desc,
                                                        const unsigned long long* n_ptr, uint64_t* keys) {
  const uint64_t n = *n_ptr;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    const uint64_t k = vh_topk_key(cls, elem, vh_load_sized(src, esize, i));
    keys[i] = desc ? k : ~k;                                         // always select the LARGEST keys
  }
}

// one radix-select pass: histogram of the byte at `shift` among keys that match the prefix chosen so far
__global__ __launch_bounds__(256) void topk_hist_kernel(const uint64_t* keys, const unsigned long long* n_ptr, int shift, VhTopkState* S) {
  __shared__ unsigned int h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t n = *n_ptr;
  const uint64_t prefix = S->prefix;
  const uint64_t hi_mask = shift >= 56 ? 0ull : ~0ull << (shift + 8);
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    const uint64_t k = keys[i];
    if ((k & hi_mask) == (prefix & hi_mask)) atomicAdd(&h[(k >> shift) & 0xFF], 1u);
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&S->hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

__global__ void topk_pick_kernel(int shift, VhTopkState* S) {
  if (threadIdx.x != 0) return;
  unsigned long long need = S->k_remaining, cum = 0;
  int d = 255;
  for (; d > 0; --d) {
    if (cum + S->hist[d] >= need) break;
    cum += S->hist[d];
  }
  S->k_remaining = need - cum;                 // rank inside the chosen digit's bucket
  S->prefix |= (unsigned long long)d << shift;
  for (int i = 0; i < 256; ++i) S->hist[i] = 0;
}

struct VhTopkCompact {
  int32_t ncols; int32_t pad;
  uint64_t slack;
  const void* src[VH_MAX_GROUP + VH_MAX_METRIC];
  void* dst[VH_MAX_GROUP + VH_MAX_METRIC];
  uint32_t esize[VH_MAX_GROUP + VH_MAX_METRIC];
};

__global__ __launch_bounds__(256) void topk_compact_kernel(const VhTopkCompact A, const uint64_t* keys, const unsigned long long* n_ptr,
                                                           unsigned long long k, VhTopkState* S) {
  const uint64_t n = *n_ptr;
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const uint64_t T = n > k ? S->prefix : 0ull;
  const uint64_t thr = T > A.slack ? T - A.slack : 0ull;
  const bool have = i < n && keys[i] >= thr;
  const uint64_t bal = __ballot(have);
  if (bal == 0) return;
  unsigned long long base = 0;
  if (lane == 0) base = atomicAdd(&S->out_count, (unsigned long long)__popcll(bal));
  base = __shfl(base, 0);
  if (!have) return;
  const uint64_t pos = base + __popcll(bal & ((1ull << lane) - 1ull));
  for (int c = 0; c < A.ncols; ++c) vh_store_sized(A.dst[c], A.esize[c], pos, vh_load_sized(A.src[c], A.esize[c], i));
}

// ------------------------------------------------- key-partitioned exchange (hash path)
// Emitted groups of one GPU are regrouped by owner = mix(key columns) % nparts so that every column can be
// shipped with one all-to-all; the owner then merges what it received by re-aggregation. Two passes of the
// same kernel: pass 0 counts rows per owner, pass 1 scatters (wave-aggregated cursor bumps).
#define VH_MAX_XCHG_COLS (VH_MAX_GROUP + VH_MAX_METRIC)
struct VhPartitionArgs {
  uint64_t n;
  uint32_t nparts; int32_t nkeys; int32_t ncols; int32_t pass;
  const void* src[VH_MAX_XCHG_COLS];
  void* dst[VH_MAX_XCHG_COLS];
  uint32_t esize[VH_MAX_XCHG_COLS];
  unsigned long long* counts;            // [nparts], pass 0
  unsigned long long* cursors;           // [nparts], pass 1
  const unsigned long long* offsets;     // [nparts + 1], pass 1
};

// Owner bookkeeping shared by the two partition kernels. A block handles VH_XCHG_SPAN x 256 rows: owners are counted
// in an LDS histogram, each owner's block total takes ONE global atomic (a returning atomic per wave and owner on a
// handful of addresses serialises, like the emission kernel's used to), and rows get their slot from an LDS cursor.
#define VH_XCHG_SPAN 16
__device__ __forceinline__ void vh_xchg_positions(uint32_t nparts, int pass, unsigned long long* counts, unsigned long long* cursors,
                                                  const unsigned long long* offsets, const uint32_t (&owner)[VH_XCHG_SPAN],
                                                  uint64_t (&pos)[VH_XCHG_SPAN]) {
  __shared__ unsigned int hist[64];
  __shared__ unsigned long long basep[64];
  if (threadIdx.x < 64) hist[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < VH_XCHG_SPAN; ++k)
    if (owner[k] != 0xFFFFFFFFu) atomicAdd(&hist[owner[k]], 1u);
  __syncthreads();
  if (threadIdx.x < nparts) {
    const unsigned int c = hist[threadIdx.x];
    unsigned long long b = 0;
    if (c) b = atomicAdd((pass == 0 ? counts : cursors) + threadIdx.x, (unsigned long long)c);
    basep[threadIdx.x] = (offsets ? offsets[threadIdx.x] : 0ull) + b;
    hist[threadIdx.x] = 0;   // becomes the cursor inside the block
  }
  __syncthreads();
  if (pass == 0) return;
#pragma unroll
  for (int k = 0; k < VH_XCHG_SPAN; ++k)
    if (owner[k] != 0xFFFFFFFFu) pos[k] = basep[owner[k]] + atomicAdd(&hist[owner[k]], 1u);
}

__global__ __launch_bounds__(256) void partition_groups_kernel(const VhPartitionArgs A) {
  const uint64_t first = (uint64_t)blockIdx.x * (256 * VH_XCHG_SPAN) + threadIdx.x;
  uint32_t owner[VH_XCHG_SPAN];
  uint64_t pos[VH_XCHG_SPAN];
#pragma unroll
  for (int k = 0; k < VH_XCHG_SPAN; ++k) {
    const uint64_t i = first + (uint64_t)k * 256;
    owner[k] = 0xFFFFFFFFu;
    pos[k] = 0;
    if (i < A.n) {
      uint64_t h = 0x9E3779B97F4A7C15ull;
      for (int c = 0; c < A.nkeys; ++c) h = vh_splitmix64(h ^ vh_load_sized(A.src[c], A.esize[c], i));
      owner[k] = (uint32_t)(h % A.nparts);
    }
  }
  vh_xchg_positions(A.nparts, A.pass, A.counts, A.cursors, A.offsets, owner, pos);
  if (A.pass == 0) return;
#pragma unroll
  for (int k = 0; k < VH_XCHG_SPAN; ++k) {
    if (owner[k] == 0xFFFFFFFFu) continue;
    const uint64_t i = first + (uint64_t)k * 256;
    for (int c = 0; c < A.ncols; ++c) vh_store_sized(A.dst[c], A.esize[c], pos[k], vh_load_sized(A.src[c], A.esize[c], i));
  }
}

// ------------------------------------------------- select: ordered row emission
// The reference's SelectQuery (src/codegen/query/scan.cc:75-166) sends the passing rows in storage order through
// a skip/limit window. Same scan geometry and predicate code as the aggregate kernels, different sink:
//   1. select_count_kernel : passing rows per chunk (one wave-step = 1024 rows)
//   2. select_scan_kernel  : per segment, exclusive prefix over its chunks + segment total
//      (host turns the totals into one ordinal window per segment — the reference's skip/limit/break rules)
//   3. select_emit_kernel  : predicate again; ordinal of a passing row = chunk prefix + ballot prefix; rows
//      inside the segment's window gather the selected columns into dense output arrays, in order.
#define VH_MAX_SELECT 32
struct VhSelectDev {              // lives in device scratch, read through a pointer
  int32_t ncols; int32_t pad;
  const char* base[VH_MAX_SELECT];          // column arena (segment 0), or NULL for a bitset column
  uint64_t stride[VH_MAX_SELECT];
  const uint64_t* const* bs_offs[VH_MAX_SELECT];   // bitset column: [nseg] -> offsets[rows + 1] (value = cardinality)
  uint32_t esize[VH_MAX_SELECT];
  void* out[VH_MAX_SELECT];
};
struct VhSelectWindow { uint64_t lo, hi, out_base; };   // ordinals [lo, hi) of the segment's passing rows -> out_base + (o - lo)

__device__ __forceinline__ uint32_t vh_select_mask(const VhPlanDev& P, uint32_t seg, uint32_t wave_base, uint32_t seg_rows, int lane) {
  const uint32_t row_l = wave_base + lane * 4;
  return wave_base + VH_WAVE_STEP_ROWS <= seg_rows ? vh_eval_filter<true>(P, seg, row_l, seg_rows)
                                                    : vh_eval_filter<false>(P, seg, row_l, seg_rows);
}

__global__ __launch_bounds__(256) void select_count_kernel(const VhPlanDev P, uint32_t chunks_per_seg, uint32_t* counts) {
  const int lane = threadIdx.x & 63;
  const uint64_t nchunks = (uint64_t)P.nseg * chunks_per_seg;
  unsigned long long npassed = 0;
  for (uint64_t c = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); c < nchunks; c += (uint64_t)gridDim.x * 4) {
    const uint32_t seg = (uint32_t)(c / chunks_per_seg);
    const uint32_t wave_base = (uint32_t)(c - (uint64_t)seg * chunks_per_seg) * VH_WAVE_STEP_ROWS;
    const uint32_t seg_rows = P.seg_rows[seg];
    uint32_t n = 0;
    if (wave_base < seg_rows) {
      n = __popc(vh_select_mask(P, seg, wave_base, seg_rows, lane));
      for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off);
    }
    if (lane == 0) { counts[c] = n; npassed += n; }
  }
  if (lane == 0 && npassed) atomicAdd(P.counters + 0, npassed);
}

__global__ __launch_bounds__(256) void select_scan_kernel(uint32_t* counts, uint32_t chunks_per_seg, unsigned long long* seg_totals) {
  __shared__ uint32_t wsum[4];
  uint32_t* c = counts + (uint64_t)blockIdx.x * chunks_per_seg;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long carry = 0;
  for (uint32_t t0 = 0; t0 < chunks_per_seg; t0 += 256) {
    const uint32_t i = t0 + threadIdx.x;
    const uint32_t v = i < chunks_per_seg ? c[i] : 0u;
    uint32_t incl = v;
    for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(incl, off); if (lane >= off) incl += o; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t before = 0, tile = 0;
    for (int w = 0; w < 4; ++w) { if (w < wave) before += wsum[w]; tile += wsum[w]; }
    if (i < chunks_per_seg) c[i] = (uint32_t)(carry + before + incl - v);   // < 2^32: a segment holds < 2^32 rows
    carry += tile;
    __syncthreads();
  }
  if (threadIdx.x == 0) seg_totals[blockIdx.x] = carry;
}

__global__ __launch_bounds__(256) void select_emit_kernel(const VhPlanDev P, uint32_t chunks_per_seg, const uint32_t* prefix,
                                                          const VhSelectWindow* win, const VhSelectDev* S) {
  const int lane = threadIdx.x & 63;
  const uint64_t lanemask_lt = (1ull << lane) - 1ull;
  const uint64_t nchunks = (uint64_t)P.nseg * chunks_per_seg;
  const int ncols = S->ncols;
  for (uint64_t c = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); c < nchunks; c += (uint64_t)gridDim.x * 4) {
    const uint32_t seg = (uint32_t)(c / chunks_per_seg);
    const uint32_t wave_base = (uint32_t)(c - (uint64_t)seg * chunks_per_seg) * VH_WAVE_STEP_ROWS;
    const uint32_t seg_rows = P.seg_rows[seg];
    if (wave_base >= seg_rows) continue;
    const VhSelectWindow w = win[seg];
    uint64_t ord = prefix[c];                       // ordinal (within the segment) of this chunk's first passing row
    if (w.lo >= w.hi || ord >= w.hi) continue;
    const uint32_t mask = vh_select_mask(P, seg, wave_base, seg_rows, lane);
#pragma unroll
    for (int k = 0; k < VH_SUBSTEPS; ++k) {
      const uint32_t mk = (mask >> (4 * k)) & 0xFu;
      uint32_t below = 0, total = 0;                // passing rows of this sub-step in lower lanes / in all lanes
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint64_t bal = __ballot((mk >> j) & 1u);
        below += __popcll(bal & lanemask_lt);
        total += __popcll(bal);
      }
      uint64_t o = ord + below;
      for (int j = 0; j < 4; ++j) {
        if (!((mk >> j) & 1u)) continue;
        if (o >= w.lo && o < w.hi) {
          const uint32_t row = wave_base + lane * 4 + k * 256u + j;
          const uint64_t dst = w.out_base + (o - w.lo);
          for (int cidx = 0; cidx < ncols; ++cidx) {
            uint64_t v;
            if (S->base[cidx]) v = vh_load_sized(S->base[cidx] + (uint64_t)seg * S->stride[cidx], S->esize[cidx], row);
            else { const uint64_t* offs = S->bs_offs[cidx][seg]; v = offs[row + 1] - offs[row]; }
            vh_store_sized(S->out[cidx], S->esize[cidx], dst, v);
          }
        }
        ++o;
      }
      ord += total;
    }
  }
}

// Count-distinct across GPUs ("(group, value) pairs take the same route; distinct-count is finalised
// on the owner"). A rank's partial for a bitset metric is not its cardinalities but the SET of distinct
// (group, id) pairs it saw — the device-wide set the scan kernel filled. This kernel walks that set, turns the
// group slot back into the group's key columns, and regroups the pairs by the SAME owner function as the groups
// (partition_groups_kernel), so the owner of a group receives its states and its pairs.
struct VhPairArgs {
  int32_t mode, ngroup, key_words, wide;
  uint64_t nslots;                 // capacity of the (group, id) set
  uint64_t hcap;                   // hash mode: capacity of the group table (slot hcap = the reserved sentinel group)
  const uint64_t* hkeys;
  uint64_t hstride;                // u64 words between the keys of consecutive group slots
  const uint64_t* dkeys; const uint32_t* dtags;
  uint64_t glo[VH_MAX_GROUP], gextent[VH_MAX_GROUP], gstride[VH_MAX_GROUP];
  uint32_t gkey_word[VH_MAX_GROUP], gkey_shift[VH_MAX_GROUP], gesize[VH_MAX_GROUP];
  uint32_t nparts; int32_t pass;
  void* dst[VH_MAX_GROUP + 1];     // key columns, then the id column (u32 / u64)
  unsigned long long* counts; unsigned long long* cursors; const unsigned long long* offsets;
};

__device__ __forceinline__ bool vh_pair_at(const VhPairArgs& A, uint64_t i, uint64_t& gid, uint64_t& id) {
  if (i >= A.nslots) return false;
  if (A.wide) { if (A.dtags[i] != 2u) return false; gid = A.dkeys[2 * i]; id = A.dkeys[2 * i + 1]; return true; }
  const uint64_t w = A.dkeys[i];
  if (w == VH_HASH_EMPTY) return false;
  gid = w >> 32; id = w & 0xFFFFFFFFull;
  return true;
}
__device__ __forceinline__ uint64_t vh_pair_key(const VhPairArgs& A, uint64_t gid, int c) {
  uint64_t v;
  if (A.mode == VH_MODE_HASH) {
    const uint64_t w = gid == A.hcap ? VH_HASH_EMPTY : A.hkeys[gid * A.hstride + A.gkey_word[c]];
    v = w >> A.gkey_shift[c];
  } else {
    v = A.glo[c] + (gid / A.gstride[c]) % A.gextent[c];
  }
  if (A.gesize[c] < 8) v &= (1ull << (8 * A.gesize[c])) - 1ull;     // what the emitted column holds
  return v;
}

__global__ __launch_bounds__(256) void partition_pairs_kernel(const VhPairArgs A) {
  const uint64_t first = (uint64_t)blockIdx.x * (256 * VH_XCHG_SPAN) + threadIdx.x;
  uint32_t owner[VH_XCHG_SPAN];
  uint64_t pos[VH_XCHG_SPAN];
#pragma unroll
  for (int k = 0; k < VH_XCHG_SPAN; ++k) {
    uint64_t gid = 0, id = 0;
    owner[k] = 0xFFFFFFFFu;
    pos[k] = 0;
    if (vh_pair_at(A, first + (uint64_t)k * 256, gid, id)) {
      uint64_t h = 0x9E3779B97F4A7C15ull;
      for (int c = 0; c < A.ngroup; ++c) h = vh_splitmix64(h ^ vh_pair_key(A, gid, c));
      owner[k] = (uint32_t)(h % A.nparts);
    }
  }
  vh_xchg_positions(A.nparts, A.pass, A.counts, A.cursors, A.offsets, owner, pos);
  if (A.pass == 0) return;
#pragma unroll
  for (int k = 0; k < VH_XCHG_SPAN; ++k) {
    if (owner[k] == 0xFFFFFFFFu) continue;
    uint64_t gid = 0, id = 0;
    (void)vh_pair_at(A, first + (uint64_t)k * 256, gid, id);
    for (int c = 0; c < A.ngroup; ++c) vh_store_sized(A.dst[c], A.gesize[c], pos[k], vh_pair_key(A, gid, c));
    vh_store_sized(A.dst[A.ngroup], A.wide ? 8u : 4u, pos[k], id);
  }
}

// The same for a result of the hashed partitioning (vh_hpart.h), which keeps no device-wide (group, id) set: the ids still lie in the
// last tuple pool, next to the mixed key of their row's group — every (group, id) a rank saw, duplicates included, which the owner's set
// union does not mind. An item is (extent, tuple, first or second id); extents are mostly part full, items beyond their fill are skipped.
struct VhHpPairArgs {
  const uint64_t* tuples;          // pool b: 32-byte tuples (mixed key, payload, two ids, how many of them count | ids only), or packed 16-byte ones (vh_hpart.h)
  int32_t pk, pk_pbits, pk_idbits, pad;
  const uint16_t* fill;            // tuples per extent
  uint32_t max_extents, stride, et;   // extents; tuples between extent starts; tuples an extent can hold
  int32_t ngroup;
  uint32_t gkey_shift[VH_MAX_GROUP], gesize[VH_MAX_GROUP];      // (one key word)
  uint32_t nparts; int32_t pass;
  void* dst[VH_MAX_GROUP + 1];     // key columns, then the id column (u32)
  unsigned long long* counts; unsigned long long* cursors; const unsigned long long* offsets;
};
__device__ __forceinline__ bool vh_hp_pair_at(const VhHpPairArgs& A, uint64_t i, uint64_t& key, uint32_t& id) {
  const uint64_t per = (uint64_t)A.et * 2;
  const uint64_t e = i / per;
  if (e >= A.max_extents) return false;
  const uint32_t rem = (uint32_t)(i - e * per), tup = rem >> 1, q = rem & 1u;
  if (tup >= A.fill[e]) return false;
  if (A.pk) {
    const uint64_t* w = A.tuples + (e * A.stride + tup) * 2;
    const uint64_t w1 = w[1];
    if (q >= (uint32_t)((w1 >> 61) & 3ull)) return false;
    key = vh_unmix64(w[0]);
    id = (uint32_t)((w1 >> (A.pk_pbits + (q ? A.pk_idbits : 0))) & ((1ull << A.pk_idbits) - 1ull));
    return true;
  }
  const uint64_t* w = A.tuples + (e * A.stride + tup) * 4;
  if (q >= (uint32_t)(w[3] & 3ull)) return false;
  key = vh_unmix64(w[0]);
  id = q ? (uint32_t)(w[2] >> 32) : (uint32_t)w[2];
  return true;
}
__device__ __forceinline__ uint64_t vh_hp_pair_key(const VhHpPairArgs& A, uint64_t key, int c) {
  uint64_t v = key >> A.gkey_shift[c];
  if (A.gesize[c] < 8) v &= (1ull << (8 * A.gesize[c])) - 1ull;     // what the emitted column holds
  return v;
}
__global__ __launch_bounds__(256) void hp_partition_pairs_kernel(const VhHpPairArgs A) {
  const uint64_t first = (uint64_t)blockIdx.x * (256 * VH_XCHG_SPAN) + threadIdx.x;
  uint32_t owner[VH_XCHG_SPAN];
  uint64_t pos[VH_XCHG_SPAN];
#pragma unroll
  for (int k = 0; k < VH_XCHG_SPAN; ++k) {
    uint64_t key = 0; uint32_t id = 0;
    owner[k] = 0xFFFFFFFFu;
    pos[k] = 0;
    if (vh_hp_pair_at(A, first + (uint64_t)k * 256, key, id)) {
      uint64_t h = 0x9E3779B97F4A7C15ull;
      for (int c = 0; c < A.ngroup; ++c) h = vh_splitmix64(h ^ vh_hp_pair_key(A, key, c));
      owner[k] = (uint32_t)(h % A.nparts);
    }
  }
  vh_xchg_positions(A.nparts, A.pass, A.counts, A.cursors, A.offsets, owner, pos);
  if (A.pass == 0) return;
#pragma unroll
  for (int k = 0; k < VH_XCHG_SPAN; ++k) {
    if (owner[k] == 0xFFFFFFFFu) continue;
    uint64_t key = 0; uint32_t id = 0;
    (void)vh_hp_pair_at(A, first + (uint64_t)k * 256, key, id);
    for (int c = 0; c < A.ngroup; ++c) vh_store_sized(A.dst[c], A.gesize[c], pos[k], vh_hp_pair_key(A, key, c));
    vh_store_sized(A.dst[A.ngroup], 4u, pos[k], id);
  }
}

// The first 512 bytes of a result region — scan counters and the emitted-row count — into the pinned host buffer
// the emission kernel wrote its rows to (direct emission, see result_finalize_locked).
// Big results leave through THIS kernel, not through the DMA engine: a few blocks that read the output columns and store them into the
// pinned (device-visible) staging buffer in whole aligned 16-byte pieces, 1 KB contiguous per wave instruction. Measured
// (tools/experiments/d2h_bw*.hip, tools/r04): hipMemcpyAsync device-to-host ran at 30 GB/s inside the library's process — 57 GB/s in a bare
// program; with HSA_ENABLE_SDMA=0 the runtime's own shader copies made C5 10 ms faster — while a 64-block store kernel reaches 55 GB/s
// everywhere and leaves the other CUs to the chunks that are still aggregating. Source and destination offsets of a column differ by
// whole rows, not by multiples of 16 bytes: the loads are dword-aligned (funnel-shifted when a 1- or 2-byte column starts mid-dword).
#define VH_DELIVER_COLS (VH_MAX_GROUP + VH_MAX_METRIC)
struct VhDeliverArgs {
  int32_t ncols, pad;
  const char* src[VH_DELIVER_COLS];
  char* dst[VH_DELIVER_COLS];
  uint64_t bytes[VH_DELIVER_COLS];
};
typedef uint32_t vh_u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
__global__ __launch_bounds__(256) void deliver_kernel(const VhDeliverArgs A) {
  const uint64_t gtid = (uint64_t)blockIdx.x * 256 + threadIdx.x, nthr = (uint64_t)gridDim.x * 256;
  for (int c = 0; c < A.ncols; ++c) {
    const char* __restrict__ src = A.src[c];
    char* __restrict__ dst = A.dst[c];
    const uint64_t n = A.bytes[c];
    uint64_t head = (16u - (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 15u)) & 15u;
    if (head > n) head = n;
    if (gtid < head) dst[gtid] = src[gtid];
    const uint64_t nvec = (n - head) / 16;
    const char* s0 = src + head;
    const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(s0) & 3u) * 8u;       // bits the source lies beyond a dword boundary
    const uint32_t* sw = reinterpret_cast<const uint32_t*>(s0 - (sh >> 3));
    vh_u32x4* dv = reinterpret_cast<vh_u32x4*>(dst + head);
    for (uint64_t v = gtid; v < nvec; v += nthr) {
      vh_u32x4 w = *reinterpret_cast<const vh_u32x4_a4*>(sw + 4 * v);
      if (sh) {
        const uint32_t e = sw[4 * v + 4];
        w.x = (w.x >> sh) | (w.y << (32u - sh)); w.y = (w.y >> sh) | (w.z << (32u - sh));
        w.z = (w.z >> sh) | (w.w << (32u - sh)); w.w = (w.w >> sh) | (e << (32u - sh));
      }
      __builtin_nontemporal_store(w, dv + v);
    }
    const uint64_t done = head + nvec * 16;
    if (gtid < n - done) dst[done + gtid] = src[done + gtid];
  }
}

// Streamed delivery of a hashed-partitioning result (VhHpArgs::nchunks): a chunk's row count into pinned host memory, right behind the chunk.
__global__ __launch_bounds__(64) void publish_count_kernel(unsigned long long* host, const unsigned long long* dev) {
  if (threadIdx.x == 0) *host = *dev;
}
__global__ __launch_bounds__(64) void publish_header_kernel(unsigned long long* host, const unsigned long long* dev) {
  host[threadIdx.x] = dev[threadIdx.x];
}

// Hash table of records (VhPlanDev::hrec_bytes): every slot gets the same `words`-word template (empty key, state identities).
struct VhRecordTemplate { uint64_t w[8]; };
__global__ __launch_bounds__(256) void fill_records_kernel(uint64_t* p, uint64_t nwords, uint32_t words, VhRecordTemplate T) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (uint64_t)gridDim.x * 256) p[i] = T.w[i % words];
}

__global__ __launch_bounds__(256) void narrow_offsets_kernel(const uint64_t* src, uint32_t* dst, uint64_t n) {      // CSR offsets of a segment with < 2^32 ids
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) dst[i] = (uint32_t)src[i];
}
__global__ __launch_bounds__(256) void iota_kernel(uint64_t* p, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) p[i] = i;
}

// ----------------------------------------------------------- payload projection ("pack")
// Row-major copy of a few columns of a table (vh_table_pack): record r of a segment holds the values of row r, every
// column at a fixed, naturally aligned offset; records are a power of two <= 64 B, so one never straddles a 128 B line.
// The compacting scan kernels gather a survivor's group / metric values from it with ONE line touched instead of one
// per column (C3: four lines of four column arenas -> one 32 B record). Built from the column arenas in HBM; a block
// stages 256 records in LDS so that both the column reads and the record writes are coalesced.
#define VH_PACK_MAX_COLS 8
// One unit of work of the derived-layout kernels: rows [first, first + count) of segment `seg` (first a multiple of 256, count a multiple of 4
// and at most VH_JOB_ROWS), of which the segment holds `seg_rows`. The host cuts what changed since a layout was last refreshed — whole
// segments when it is built, the row ranges an upsert batch touched afterwards (vh_table::journal) — into such jobs; one block each.
#define VH_JOB_ROWS 16384u
struct VhJob { uint32_t seg, first, count, seg_rows; };
struct VhPackArgs {
  int32_t ncols; uint32_t rec_bytes;
  const char* src[VH_PACK_MAX_COLS];   // column arenas
  uint64_t src_stride[VH_PACK_MAX_COLS];
  uint32_t esize[VH_PACK_MAX_COLS], off[VH_PACK_MAX_COLS];
  uint32_t wbytes[VH_PACK_MAX_COLS];   // bytes a value takes in the record (< esize: a compressed projection, the value's low bytes)
  uint32_t sgn_mask;                   // bit c: column c is a signed integer (its stored bytes sign-extend)
  unsigned int* overflow;              // set when a value does not survive its stored width (the projection is then void)
  char* dst; uint64_t dst_stride;      // pack arena, bytes between segments
  const VhJob* jobs;                   // [gridDim.x]
};
__global__ __launch_bounds__(256) void pack_kernel(const VhPackArgs A) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const VhJob J = A.jobs[blockIdx.x];
  const uint32_t seg = J.seg, nrows = J.seg_rows, rec = A.rec_bytes;
  for (uint32_t i = threadIdx.x; i < 256u * rec / 16u; i += 256u) reinterpret_cast<vh_u32x4*>(lds)[i] = vh_u32x4{0u, 0u, 0u, 0u};
  __syncthreads();
  char* dst = A.dst + (uint64_t)seg * A.dst_stride;
  for (uint32_t row0 = J.first; row0 < J.first + J.count && row0 < nrows; row0 += 256u) {
    const uint32_t row = row0 + threadIdx.x;
    if (row < nrows) {
      bool ovf = false;
      for (int c = 0; c < A.ncols; ++c) {
        const char* s = A.src[c] + (uint64_t)seg * A.src_stride[c] + (uint64_t)row * A.esize[c];
        char* d = lds + threadIdx.x * rec + A.off[c];
        uint64_t v;
        switch (A.esize[c]) {
          case 1: v = *reinterpret_cast<const uint8_t*>(s); break;
          case 2: v = *reinterpret_cast<const uint16_t*>(s); break;
          case 4: v = *reinterpret_cast<const uint32_t*>(s); break;
          default: v = *reinterpret_cast<const uint64_t*>(s); break;
        }
        const uint32_t w = A.wbytes[c];
        if (w < A.esize[c]) {           // compressed: the low w bytes must give the value back (zero- or sign-extended)
          const int sh = 64 - 8 * (int)w, sh_e = 64 - 8 * (int)A.esize[c];
          if ((A.sgn_mask >> c) & 1u) ovf |= ((int64_t)(v << sh_e) >> sh_e) != ((int64_t)(v << sh) >> sh);
          else ovf |= (v << sh >> sh) != v;
        }
        switch (w) {
          case 1: *reinterpret_cast<uint8_t*>(d) = (uint8_t)v; break;
          case 2: *reinterpret_cast<uint16_t*>(d) = (uint16_t)v; break;
          case 4: *reinterpret_cast<uint32_t*>(d) = (uint32_t)v; break;
          default: *reinterpret_cast<uint64_t*>(d) = v; break;
        }
      }
      if (ovf) atomicOr(A.overflow, 1u);
    }
    __syncthreads();
    vh_u32x4* out = reinterpret_cast<vh_u32x4*>(dst + (uint64_t)row0 * rec);
    for (uint32_t i = threadIdx.x; i < 256u * rec / 16u; i += 256u) out[i] = reinterpret_cast<const vh_u32x4*>(lds)[i];
    __syncthreads();
  }
}

// Bit-field records (VhPack::bits): one thread per row, every column's value OR-ed into the record word at its bit offset. A value that needs
// more bits than the record gives it (or a negative one: the fields are unsigned) voids the projection, like a byte width outgrown above.
struct VhPackBitsArgs {
  int32_t ncols; uint32_t rec_bytes;
  const char* src[VH_PACK_MAX_COLS];
  uint64_t src_stride[VH_PACK_MAX_COLS];
  uint32_t esize[VH_PACK_MAX_COLS], bitoff[VH_PACK_MAX_COLS], bitw[VH_PACK_MAX_COLS];
  unsigned int* overflow;
  char* dst; uint64_t dst_stride;
  const VhJob* jobs;
};
__global__ __launch_bounds__(256) void pack_bits_kernel(const VhPackBitsArgs A) {
  const VhJob J = A.jobs[blockIdx.x];
  const uint32_t seg = J.seg, nrows = J.seg_rows < J.first + J.count ? J.seg_rows : J.first + J.count;
  char* dst = A.dst + (uint64_t)seg * A.dst_stride;
  bool ovf = false;
  for (uint32_t row = J.first + threadIdx.x; row < nrows; row += 256u) {
    uint64_t rec = 0;
    for (int c = 0; c < A.ncols; ++c) {
      const char* s = A.src[c] + (uint64_t)seg * A.src_stride[c] + (uint64_t)row * A.esize[c];
      uint64_t v;
      switch (A.esize[c]) {
        case 1: v = *reinterpret_cast<const uint8_t*>(s); break;
        case 2: v = *reinterpret_cast<const uint16_t*>(s); break;
        case 4: v = *reinterpret_cast<const uint32_t*>(s); break;
        default: v = *reinterpret_cast<const uint64_t*>(s); break;
      }
      if (A.bitw[c] < 64u) ovf |= (v >> A.bitw[c]) != 0ull;      // (a negative value of a signed column has its top bits set: caught here too)
      rec |= v << A.bitoff[c];
    }
    if (A.rec_bytes == 4) reinterpret_cast<uint32_t*>(dst)[row] = (uint32_t)rec;
    else reinterpret_cast<uint64_t*>(dst)[row] = rec;
  }
  if (ovf) atomicOr(A.overflow, 1u);
}

// Bit-packed predicate projection (vh_table_predpack): every row's predicate columns as bit fields of one word (<= 32 bits), the word kept as
// byte planes of 2 or 1 bytes per row. One row per thread and step; values are known to fit their fields (the host sizes the fields from
// the columns' recorded min / max, which cover every mirrored value, and drops the projection when they no longer do).
struct VhPredPackArgs {
  int32_t ncols, nplanes;
  const char* src[VH_PACK_MAX_COLS];
  uint64_t src_stride[VH_PACK_MAX_COLS];
  uint32_t esize[VH_PACK_MAX_COLS], bitoff[VH_PACK_MAX_COLS];
  char* plane[4]; uint64_t plane_stride[4]; uint32_t plane_width[4], plane_pos[4];
  const VhJob* jobs;
};
__global__ __launch_bounds__(256) void predpack_kernel(const VhPredPackArgs A) {
  const VhJob J = A.jobs[blockIdx.x];
  const uint32_t seg = J.seg, end = J.first + J.count;
  for (uint32_t row = J.first + threadIdx.x; row < end; row += 256u) {
    uint32_t word = 0;
    if (row < J.seg_rows) {
      for (int c = 0; c < A.ncols; ++c) {
        const char* s = A.src[c] + (uint64_t)seg * A.src_stride[c] + (uint64_t)row * A.esize[c];
        uint32_t v;
        switch (A.esize[c]) {
          case 1: v = *reinterpret_cast<const uint8_t*>(s); break;
          case 2: v = *reinterpret_cast<const uint16_t*>(s); break;
          case 4: v = *reinterpret_cast<const uint32_t*>(s); break;
          default: v = (uint32_t)*reinterpret_cast<const uint64_t*>(s); break;
        }
        word |= v << A.bitoff[c];
      }
    }
    for (int q = 0; q < A.nplanes; ++q) {
      char* d = A.plane[q] + (uint64_t)seg * A.plane_stride[q];
      if (A.plane_width[q] == 2) reinterpret_cast<uint16_t*>(d)[row] = (uint16_t)(word >> A.plane_pos[q]);
      else reinterpret_cast<uint8_t*>(d)[row] = (uint8_t)(word >> A.plane_pos[q]);
    }
  }
}

// The BIT-SLICED form (VhPredPack::sliced): bit b of the row word is plane b, one bit per row — 64 rows of a plane are one 8-byte word, which
// is what a wave's ballot of "bit b of my row's word" IS. A wave takes 64 rows a step; lane b keeps plane b's ballot and stores it. plane[0] =
// the projection's arena, plane_stride[0] = bytes between segments, plane_pos[0] .. = unused; `bits` planes lie `pitch` bytes apart.
__global__ __launch_bounds__(256) void predslice_kernel(const VhPredPackArgs A, uint32_t bits, uint64_t pitch) {
  const VhJob J = A.jobs[blockIdx.x];
  const uint32_t seg = J.seg, end = J.first + J.count;
  const int lane = threadIdx.x & 63;
  char* dst = A.plane[0] + (uint64_t)seg * A.plane_stride[0];
  for (uint32_t row0 = J.first + (threadIdx.x >> 6) * 64u; row0 < end; row0 += 256u) {
    const uint32_t row = row0 + lane;
    uint32_t word = 0;
    if (row < J.seg_rows) {
      for (int c = 0; c < A.ncols; ++c) {
        const char* s = A.src[c] + (uint64_t)seg * A.src_stride[c] + (uint64_t)row * A.esize[c];
        uint32_t v;
        switch (A.esize[c]) {
          case 1: v = *reinterpret_cast<const uint8_t*>(s); break;
          case 2: v = *reinterpret_cast<const uint16_t*>(s); break;
          case 4: v = *reinterpret_cast<const uint32_t*>(s); break;
          default: v = (uint32_t)*reinterpret_cast<const uint64_t*>(s); break;
        }
        word |= v << A.bitoff[c];
      }
    }
    unsigned long long mine = 0ull;
    for (uint32_t b = 0; b < bits; ++b) {
      const unsigned long long m = __ballot((word >> b) & 1u);
      if ((uint32_t)lane == b) mine = m;
    }
    if ((uint32_t)lane < bits) *reinterpret_cast<unsigned long long*>(dst + (uint64_t)lane * pitch + (uint64_t)(row0 >> 6) * 8ull) = mine;
  }
}

// Narrow copy of an unsigned 32-bit column whose values fit T (vh_table_narrow): one job per block, 4 elements per thread and step.
template <typename T>
__global__ __launch_bounds__(256) void narrow_kernel(const uint32_t* src, uint64_t src_stride_elems, T* dst, uint64_t dst_stride_elems, const VhJob* jobs) {
  const VhJob J = jobs[blockIdx.x];
  const uint32_t* s = src + (uint64_t)J.seg * src_stride_elems;
  T* d = dst + (uint64_t)J.seg * dst_stride_elems;
  for (uint64_t i = (uint64_t)J.first + threadIdx.x * 4u; i < (uint64_t)J.first + J.count; i += 1024u) {
    const vh_u32x4 v = *reinterpret_cast<const vh_u32x4*>(s + i);
    d[i] = (T)v.x; d[i + 1] = (T)v.y; d[i + 2] = (T)v.z; d[i + 3] = (T)v.w;
  }
}

// Every buffer a query must clear before its scan — counters, zero-identity states, presence bytes, extent tags, empty hash
// keys — in ONE launch: the runtime's fill kernels cost ~8 us apiece back to back, five to seven of them per query were
// 45 us of a 2.6 ms C3 step (profiles/r03/NOTES.md). Regions are 16-byte multiples (scratch regions are padded to 256 B).
#define VH_INIT_MAX 12
// (+ one small copy out of pinned host memory: the plan's per-segment row counts, filter program and literals — as a hipMemcpyAsync in front
// of this launch it was one more dispatch, and one more gap, before every scan)
struct VhInitArgs { int32_t n; uint32_t cp_words; char* p[VH_INIT_MAX]; uint64_t end[VH_INIT_MAX]; uint32_t pat[VH_INIT_MAX]; const uint32_t* cp_src; uint32_t* cp_dst; };   // end: running total of 16-byte units
__global__ __launch_bounds__(256) void init_regions_kernel(VhInitArgs A) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < A.cp_words; i += gridDim.x * 256) A.cp_dst[i] = A.cp_src[i];
  const uint64_t total = A.end[A.n - 1];
  for (uint64_t u = (uint64_t)blockIdx.x * 256 + threadIdx.x; u < total; u += (uint64_t)gridDim.x * 256) {
    int r = 0;
    while (u >= A.end[r]) ++r;
    const uint64_t local = u - (r ? A.end[r - 1] : 0);
    vh_u32x4 v; v.x = v.y = v.z = v.w = A.pat[r];
    reinterpret_cast<vh_u32x4*>(A.p[r])[local] = v;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void fill_kernel(T* p, uint64_t n, T v) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) p[i] = v;
}

// ----------------------------------------------------------- synthetic data
// value(c, r) = splitmix64(seed ^ c*GAMMA ^ r) reduced to the column domain.
template <typename T>
__global__ __launch_bounds__(256) void gen_kernel(T* base, uint64_t seg_stride_elems, uint64_t rows_per_seg,
                                                  uint64_t row_base, vh_gen_spec spec, uint64_t colseed, uint64_t seed) {
  const uint32_t seg = blockIdx.y;
  T* col = base + (uint64_t)seg * seg_stride_elems;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < rows_per_seg; i += (uint64_t)gridDim.x * 256) {
    const uint64_t r = row_base + (uint64_t)seg * rows_per_seg + i;
    T out;
    if (spec.mode == VH_GEN_ROWID) {
      out = (T)r;
    } else if (spec.mode == VH_GEN_CONST) {
      out = (T)spec.add;
    } else if (spec.mode == VH_GEN_SORTED) {
      out = (T)(spec.add + (int64_t)(r / spec.mod));
    } else if (spec.mode == VH_GEN_ZIPF) {
      const uint64_t h = vh_splitmix64(colseed ^ r);
      const uint32_t octaves = 64u - (uint32_t)__builtin_clzll(spec.mod);      // floor(log2 mod) + 1
      const uint32_t b = (uint32_t)(h >> 32) % octaves;
      const uint64_t v = ((1ull << b) - 1ull + ((h & 0xFFFFFFFFull) % (1ull << b))) % spec.mod;
      out = (T)(spec.add + (int64_t)v);
    } else if (spec.mode == VH_GEN_HOT && vh_splitmix64(seed ^ r ^ 0x407ull) % 1000ull < (uint64_t)spec.param) {
      out = (T)(spec.add + (int64_t)(spec.mod / 2));
    } else {
      const uint64_t h = vh_splitmix64(colseed ^ r);
      const int64_t iv = spec.add + (int64_t)(h % spec.mod);
      if (std::is_floating_point<T>::value) out = (T)((double)iv * spec.scale);
      else out = (T)iv;
    }
    col[i] = out;
  }
}

// synthetic bitset (count-distinct) column as CSR: exactly `k` ids per row, id = splitmix64(colseed ^ (r * 8 + i)) % mod
template <typename T>
__global__ __launch_bounds__(256) void gen_csr_kernel(uint64_t* offsets, T* values, uint64_t rows, uint32_t k,
                                                      uint64_t row_base, uint64_t mod, uint64_t colseed) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i <= rows; i += (uint64_t)gridDim.x * 256) {
    offsets[i] = i * k;
    if (i == rows) break;
    const uint64_t r = row_base + i;
    for (uint32_t j = 0; j < k; ++j) values[i * k + j] = (T)(vh_splitmix64(colseed ^ (r * 8 + j)) % mod);
  }
}

// ------------------------------------------------------------ segment stats
// Order-preserving map of a column value to u64 so one pair of u64 atomics does min/max.
template <typename T> __device__ __forceinline__ uint64_t vh_order_key(T v);
template <> __device__ __forceinline__ uint64_t vh_order_key<uint8_t>(uint8_t v) { return v; }
template <> __device__ __forceinline__ uint64_t vh_order_key<uint16_t>(uint16_t v) { return v; }
template <> __device__ __forceinline__ uint64_t vh_order_key<uint32_t>(uint32_t v) { return v; }
template <> __device__ __forceinline__ uint64_t vh_order_key<uint64_t>(uint64_t v) { return v; }
template <> __device__ __forceinline__ uint64_t vh_order_key<int8_t>(int8_t v) { return (uint64_t)(int64_t)v ^ (1ull << 63); }
template <> __device__ __forceinline__ uint64_t vh_order_key<int16_t>(int16_t v) { return (uint64_t)(int64_t)v ^ (1ull << 63); }
template <> __device__ __forceinline__ uint64_t vh_order_key<int32_t>(int32_t v) { return (uint64_t)(int64_t)v ^ (1ull << 63); }
template <> __device__ __forceinline__ uint64_t vh_order_key<int64_t>(int64_t v) { return (uint64_t)v ^ (1ull << 63); }
template <> __device__ __forceinline__ uint64_t vh_order_key<float>(float v) {
  const uint32_t b = __float_as_uint(v);
  return (b & 0x80000000u) ? (uint32_t)~b : (b | 0x80000000u);
}
template <> __device__ __forceinline__ uint64_t vh_order_key<double>(double v) {
  const uint64_t b = (uint64_t)__double_as_longlong(v);
  return (b & (1ull << 63)) ? ~b : (b | (1ull << 63));
}

// stats[(seg*2)+0] = min key, +1 = max key; pre-filled with ~0 / 0.
template <typename T>
__global__ __launch_bounds__(256) void seg_minmax_kernel(const T* base, uint64_t seg_stride_elems,
                                                         const uint32_t* seg_rows, uint32_t seg_first,
                                                         unsigned long long* stats) {
  const uint32_t seg = seg_first + blockIdx.y;
  const T* col = base + (uint64_t)seg * seg_stride_elems;
  const uint64_t n = seg_rows[blockIdx.y];
  uint64_t lo = ~0ull, hi = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    const uint64_t k = vh_order_key<T>(col[i]);
    lo = k < lo ? k : lo;
    hi = k > hi ? k : hi;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const uint64_t l2 = __shfl_down(lo, off), h2 = __shfl_down(hi, off);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  if ((threadIdx.x & 63) == 0 && n) {
    atomicMin(stats + 2ull * blockIdx.y, (unsigned long long)lo);
    atomicMax(stats + 2ull * blockIdx.y + 1, (unsigned long long)hi);
  }
}

// ------------------------------------------------------- batched dirty-range sync (vh_table_sync_batch)
// One block per descriptor = one contiguous run of one column: the block PULLS the run — out of host memory the caller registered
// (zero copy over PCIe: 16-byte loads of the source's own aligned pieces), out of the pinned staging ring, or (copy = 0) out of the arena
// itself behind a DMA copy — stores it into the arena and leaves the run's min / max order keys in ITS slot of a pinned host array: no
// atomics, nothing to clear, nothing to copy back. The host merges the slots into the per-segment stats when a planner next needs them.
struct VhSyncDesc {
  const char* src; char* dst;
  uint32_t nelem; uint8_t elem, copy; uint16_t pad0;
  uint32_t pad1, pad2;
};
template <typename T>
__device__ __forceinline__ void vh_sync_run(const VhSyncDesc& d, uint64_t& lo, uint64_t& hi) {
  constexpr int PER = 16 / (int)sizeof(T);
  const uintptr_t s = reinterpret_cast<uintptr_t>(d.src), a0 = s & ~(uintptr_t)15;
  const uint32_t head = (uint32_t)((s - a0) / sizeof(T));
  T* dst = reinterpret_cast<T*>(d.dst);
  const bool dst16 = ((reinterpret_cast<uintptr_t>(d.dst) - (uintptr_t)head * sizeof(T)) & 15) == 0;
  const uint32_t npieces = (head + d.nelem + PER - 1) / PER;
  for (uint32_t c = threadIdx.x; c < npieces; c += 256u) {
    const vh_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const vh_u32x4*>(a0) + c);   // (an aligned piece never crosses a page: what lies in it beside the run is readable)
    T e[PER];
    __builtin_memcpy(e, &v, 16);
    const int64_t i0 = (int64_t)c * PER - head;
    const bool whole = i0 >= 0 && i0 + PER <= (int64_t)d.nelem;
    if (d.copy && whole && dst16) *reinterpret_cast<vh_u32x4*>(dst + i0) = v;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int64_t i = i0 + k;
      if (i < 0 || i >= (int64_t)d.nelem) continue;
      if (d.copy && !(whole && dst16)) dst[i] = e[k];
      const uint64_t key = vh_order_key<T>(e[k]);
      lo = key < lo ? key : lo;
      hi = key > hi ? key : hi;
    }
  }
}
__global__ __launch_bounds__(256) void sync_pull_kernel(const VhSyncDesc* __restrict__ descs, unsigned long long* __restrict__ slots) {
  __shared__ uint64_t part[8];
  const VhSyncDesc d = descs[blockIdx.x];
  uint64_t lo = ~0ull, hi = 0;
  switch (d.elem) {
    case VH_U8: vh_sync_run<uint8_t>(d, lo, hi); break;
    case VH_U16: vh_sync_run<uint16_t>(d, lo, hi); break;
    case VH_U32: vh_sync_run<uint32_t>(d, lo, hi); break;
    case VH_U64: vh_sync_run<uint64_t>(d, lo, hi); break;
    case VH_I8: vh_sync_run<int8_t>(d, lo, hi); break;
    case VH_I16: vh_sync_run<int16_t>(d, lo, hi); break;
    case VH_I32: vh_sync_run<int32_t>(d, lo, hi); break;
    case VH_I64: vh_sync_run<int64_t>(d, lo, hi); break;
    case VH_F32: vh_sync_run<float>(d, lo, hi); break;
    default: vh_sync_run<double>(d, lo, hi); break;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const uint64_t l2 = __shfl_down(lo, off), h2 = __shfl_down(hi, off);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  if ((threadIdx.x & 63) == 0) { part[(threadIdx.x >> 6) * 2] = lo; part[(threadIdx.x >> 6) * 2 + 1] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) { lo = part[2 * w] < lo ? part[2 * w] : lo; hi = part[2 * w + 1] > hi ? part[2 * w + 1] : hi; }
    slots[2ull * blockIdx.x] = lo;
    slots[2ull * blockIdx.x + 1] = hi;
  }
}


// ------------------------------------------------------- bandwidth ceiling
__global__ __launch_bounds__(256) void read_bw_kernel(const vh_u32x4* p, uint64_t n16, unsigned long long* sink) {
  uint32_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) {
    const vh_u32x4 v = __builtin_nontemporal_load(p + i);
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x9E3779B9u) atomicAdd(sink, 1ull);  // defeat dead-code elimination
}

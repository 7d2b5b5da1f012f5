// vh_kernels.h — hand-written gfx950 (CDNA4, wave64) kernels of the aggregate path.
//
// What they replace in the reference (all emitted as C++ text and JIT-compiled
// per query by g++, then run on ONE host thread):
//   scan_agg_kernel    <- the segment/row loops of ScanVisitor::IterationStart
//                         (src/codegen/query/scan.cc:40-66), the row predicate of
//                         ComparisonBuilder (src/codegen/query/filter.cc:206-261),
//                         the key/metric copy + `agg_map[key].Update(m)` of
//                         ScanVisitor::Visit(AggregateQuery*) (scan.cc:168-247) with
//                         the Update semantics of TupleStruct (src/codegen/db/store.cc:131-161)
//   vh_time_rollup     <- TimestampRollup + util::Time32/Time64/Truncator
//                         (src/codegen/db/rollup.cc:77-95, src/util/time.h:57-137)
//   seg_minmax_kernel  <- SegmentStats::Update (src/codegen/db/store.cc:171-201)
//
// No MFMA anywhere: the path is compares, integer adds and table probes; the
// bound is HBM bandwidth.
#pragma once
#include "vh_internal.h"
#include "vh_time.h"
#include <type_traits>

#ifndef VH_ABLATE
#define VH_ABLATE 0      // measurement builds only (tools/build_variant.py -DVH_ABLATE=n -> viyadb_amd/build/variants/, never the shipped library):
#endif                   // 1 = no payload gathers, 2 = no aggregate sink, 4 = no tuple store in phase 1 of DENSE_PART, 8 = that store non-temporal

// ------------------------------------------------------------------ utilities
__host__ __device__ __forceinline__ uint64_t vh_splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// vh_splitmix64 is a bijection of the 64-bit integers (an addition, three xor-shifts, two odd multiplications); its inverse
// lets the hashed partitioning carry a group key as its mixed image and get the key back at the end.
__host__ __device__ __forceinline__ uint64_t vh_unmix64(uint64_t x) {
  x ^= x >> 31; x ^= x >> 62;
  x *= 0x319642B2D24D8EC3ull;           // inverse of 0x94D049BB133111EB mod 2^64
  x ^= x >> 27; x ^= x >> 54;
  x *= 0x96DE1B173F119089ull;           // inverse of 0xBF58476D1CE4E5B9 mod 2^64
  x ^= x >> 30; x ^= x >> 60;
  return x - 0x9E3779B97F4A7C15ull;
}

template <typename T> struct VhVec4 { typedef T type __attribute__((ext_vector_type(4))); };
typedef uint32_t vh_u32x4 __attribute__((ext_vector_type(4)));

template <typename T> __device__ __forceinline__ T vh_lit(uint64_t bits) {
  T v; __builtin_memcpy(&v, &bits, sizeof(T)); return v;   // low bytes, like db::AnyNum
}

__device__ __forceinline__ int vh_xcc_id() {
  // s_getreg_b32 HW_REG_XCC_ID (id 20), bits [3:0]; used for table placement only:
  // any value is correct, the right value is fast (MI355X_MICROARCH.md §dispatch).
  return (int)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u);
}

// ------------------------------------------------------- time truncation (a8): vh_time.h (vh_trunc_secs, its 32- and 64-bit forms)
// set_ts -> first matching rollup rule truncates -> query granularity truncates -> get_ts
// (src/codegen/query/scan.cc:197-218; Time64::trunc also zeroes the microseconds).
__host__ __device__ __forceinline__ uint64_t vh_time_rollup(uint64_t ts, const VhGroupDev& g) {
  uint64_t secs = g.micro() ? ts / 1000000ull : ts;
  uint64_t micros = g.micro() ? ts % 1000000ull : 0;
  for (int i = 0; i < g.nroll(); ++i) {
    if (ts < g.roll_before[i]) {
      secs = vh_trunc_secs(secs, g.roll_unit(i));
      micros = 0;
      break;
    }
  }
  if (g.gran() != VH_T_NONE) {
    secs = vh_trunc_secs(secs, g.gran());
    micros = 0;
  }
  // a `time` column is uint32 seconds (util::Time32: src/util/time.h:91-113); a non-micro time column of 8 bytes — allowed by the C ABI — keeps
  // all its bits (the compiled kernels' vj_time_rollup makes the same distinction by the column's element type)
  const bool wide = g.type() == VH_U64 || g.type() == VH_I64;
  return g.micro() ? secs * 1000000ull + micros : wide ? secs : (uint64_t)(uint32_t)secs;
}

// -------------------------------------------------------------- column loads
// 4 consecutive rows (row % 4 == 0) as one naturally aligned vector load:
// 4 B types -> global_load_dwordx4, 8 B -> 2x dwordx4, 2 B -> dwordx2, 1 B -> dword.
template <typename T>
__device__ __forceinline__ void vh_load4(const T* __restrict__ p, T* out) {
  typedef typename VhVec4<T>::type V;
  V v = __builtin_nontemporal_load(reinterpret_cast<const V*>(p));
  out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}

template <typename T, bool FULL>
__device__ __forceinline__ void vh_load16(const T* __restrict__ col, uint32_t row_l,
                                          uint32_t seg_rows, T (&v)[VH_LANE_ROWS]) {
#pragma unroll
  for (int k = 0; k < VH_SUBSTEPS; ++k) {
    const uint32_t r = row_l + k * 256u;
    if (FULL || r < seg_rows) {
      vh_load4<T>(col + r, &v[k * 4]);
    } else {
      v[k * 4] = v[k * 4 + 1] = v[k * 4 + 2] = v[k * 4 + 3] = T(0);
    }
  }
}

// (col OP lit) for 16 rows, in the column's own C++ type, like the generated
// `(tuple_dims._i[tuple_idx] OP fargK)` (filter.cc:206-221).
template <typename T>
__device__ __forceinline__ uint32_t vh_cmp16(const T (&v)[VH_LANE_ROWS], T lit, int op) {
  uint32_t m = 0;
  switch (op) {
    case VH_OP_EQ:
#pragma unroll
      for (int i = 0; i < VH_LANE_ROWS; ++i) m |= (uint32_t)(v[i] == lit) << i;
      break;
    case VH_OP_NE:
#pragma unroll
      for (int i = 0; i < VH_LANE_ROWS; ++i) m |= (uint32_t)(v[i] != lit) << i;
      break;
    case VH_OP_LT:
#pragma unroll
      for (int i = 0; i < VH_LANE_ROWS; ++i) m |= (uint32_t)(v[i] < lit) << i;
      break;
    case VH_OP_LE:
#pragma unroll
      for (int i = 0; i < VH_LANE_ROWS; ++i) m |= (uint32_t)(v[i] <= lit) << i;
      break;
    case VH_OP_GT:
#pragma unroll
      for (int i = 0; i < VH_LANE_ROWS; ++i) m |= (uint32_t)(v[i] > lit) << i;
      break;
    default:
#pragma unroll
      for (int i = 0; i < VH_LANE_ROWS; ++i) m |= (uint32_t)(v[i] >= lit) << i;
      break;
  }
  return m;
}

template <typename T, bool FULL>
__device__ __forceinline__ uint32_t vh_leaf(const VhPlanDev& P, const VhProgOp o, const char* base,
                                            uint32_t row_l, uint32_t seg_rows) {
  T v[VH_LANE_ROWS];
  vh_load16<T, FULL>(reinterpret_cast<const T*>(base), row_l, seg_rows, v);
  if (o.kind() == VH_F_REL) return vh_cmp16<T>(v, vh_lit<T>(P.lits[o.lit()]), o.op());
  // IN: OR of ==, NOT IN: AND of != (filter.cc:223-241)
  uint32_t m = o.op() ? 0u : VH_ROWMASK;
  for (int i = 0; i < o.count(); ++i) {
    const T lit = vh_lit<T>(P.lits[o.lit() + i]);
    if (o.op()) m |= vh_cmp16<T>(v, lit, VH_OP_EQ);
    else m &= vh_cmp16<T>(v, lit, VH_OP_NE);
  }
  return m;
}

// (bitset-metric OP lit): the row's cardinality in the metric's own number type (`tuple_metrics._j[idx].cardinality()`,
// filter.cc:216,235; util::Bitset<N>::cardinality returns NumType). In the CSR mirror that is offsets[r + 1] - offsets[r].
template <typename T, bool FULL>
__device__ __forceinline__ uint32_t vh_leaf_card(const VhPlanDev& P, const VhProgOp o, const uint64_t* offs, uint32_t row_l, uint32_t seg_rows) {
  T v[VH_LANE_ROWS];
#pragma unroll
  for (int k = 0; k < VH_SUBSTEPS; ++k) {
    const uint32_t r = row_l + k * 256u;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[k * 4 + j] = (r + j < seg_rows) ? (T)(offs[r + j + 1] - offs[r + j]) : T(0);   // (rows at or beyond size() are masked off below)
  }
  if (o.kind() == VH_F_REL) return vh_cmp16<T>(v, vh_lit<T>(P.lits[o.lit()]), o.op());
  uint32_t m = o.op() ? 0u : VH_ROWMASK;
  for (int i = 0; i < o.count(); ++i) {
    const T lit = vh_lit<T>(P.lits[o.lit() + i]);
    if (o.op()) m |= vh_cmp16<T>(v, lit, VH_OP_EQ);
    else m &= vh_cmp16<T>(v, lit, VH_OP_NE);
  }
  return m;
}

// Evaluate the postfix filter program for the 16 rows this lane owns in the current
// wave step. Returns a 16-bit pass mask (bit k*4+j <-> row k*256 + lane*4 + j).
// Composites are bitwise AND / OR with no short circuit, exactly like the reference.
template <bool FULL>
__device__ __forceinline__ uint32_t vh_eval_filter(const VhPlanDev& P, uint32_t seg, uint32_t row_l,
                                                   uint32_t seg_rows) {
  uint32_t st[VH_MAX_STACK];
  int sp = 0;
  for (int pc = 0; pc < P.nprog; ++pc) {
    const VhProgOp o = P.prog[pc];
    switch (o.kind()) {
      case VH_F_TRUE: st[sp++] = VH_ROWMASK; break;
      case VH_F_AND: {
        uint32_t a = st[--sp];
        for (int i = 1; i < o.count(); ++i) a &= st[--sp];
        st[sp++] = a;
      } break;
      case VH_F_OR: {
        uint32_t a = st[--sp];
        for (int i = 1; i < o.count(); ++i) a |= st[--sp];
        st[sp++] = a;
      } break;
      default: {
        if (o.type() == VH_BITSET32 || o.type() == VH_BITSET64) {
          const uint64_t* offs = P.fbs_offs[o.slot()][seg];
          st[sp++] = o.type() == VH_BITSET32 ? vh_leaf_card<uint32_t, FULL>(P, o, offs, row_l, seg_rows)
                                             : vh_leaf_card<uint64_t, FULL>(P, o, offs, row_l, seg_rows);
          break;
        }
        const char* base = P.colbase[o.slot()] + (uint64_t)seg * P.colstride[o.slot()];
        uint32_t m;
        switch (o.type()) {
          case VH_U8: m = vh_leaf<uint8_t, FULL>(P, o, base, row_l, seg_rows); break;
          case VH_U16: m = vh_leaf<uint16_t, FULL>(P, o, base, row_l, seg_rows); break;
          case VH_U32: m = vh_leaf<uint32_t, FULL>(P, o, base, row_l, seg_rows); break;
          case VH_U64: m = vh_leaf<uint64_t, FULL>(P, o, base, row_l, seg_rows); break;
          case VH_I8: m = vh_leaf<int8_t, FULL>(P, o, base, row_l, seg_rows); break;
          case VH_I16: m = vh_leaf<int16_t, FULL>(P, o, base, row_l, seg_rows); break;
          case VH_I32: m = vh_leaf<int32_t, FULL>(P, o, base, row_l, seg_rows); break;
          case VH_I64: m = vh_leaf<int64_t, FULL>(P, o, base, row_l, seg_rows); break;
          case VH_F32: m = vh_leaf<float, FULL>(P, o, base, row_l, seg_rows); break;
          default: m = vh_leaf<double, FULL>(P, o, base, row_l, seg_rows); break;
        }
        st[sp++] = m;
      } break;
    }
  }
  uint32_t m = st[0];
  if (!FULL) {
    // rows at or beyond the segment's snapshotted size() never pass
#pragma unroll
    for (int k = 0; k < VH_SUBSTEPS; ++k) {
      const uint32_t r = row_l + k * 256u;
      const uint32_t n = r >= seg_rows ? 0u : (seg_rows - r >= 4u ? 4u : seg_rows - r);
      m &= ~(((0xFu << n) & 0xFu) << (k * 4));
    }
  }
  return m;
}

// ------------------------------------------------------------ scalar gathers
// One element of a column as raw bits, zero-extended (hash keys) or sign-extended
// (dense digits, min/max on small signed types).
// `at`: address of the element. A column arena is indexed with the element size, a payload projection (VhPlanDev::colpitch,
// vh_table_pack) with its record size: the caller forms the address, the switch only picks width and extension.
__device__ __forceinline__ uint64_t vh_load_at(const char* at, int type, bool sext) {
  switch (type) {
    case VH_U8: return *reinterpret_cast<const uint8_t*>(at);
    case VH_I8: { int8_t v = *reinterpret_cast<const int8_t*>(at); return sext ? (uint64_t)(int64_t)v : (uint64_t)(uint8_t)v; }
    case VH_U16: return *reinterpret_cast<const uint16_t*>(at);
    case VH_I16: { int16_t v = *reinterpret_cast<const int16_t*>(at); return sext ? (uint64_t)(int64_t)v : (uint64_t)(uint16_t)v; }
    case VH_U32: case VH_F32: return *reinterpret_cast<const uint32_t*>(at);
    case VH_I32: { int32_t v = *reinterpret_cast<const int32_t*>(at); return sext ? (uint64_t)(int64_t)v : (uint64_t)(uint32_t)v; }
    default: return *reinterpret_cast<const uint64_t*>(at);
  }
}
__device__ __forceinline__ uint64_t vh_load_bits(const char* base, int type, uint32_t row, bool sext) {
  switch (type) {
    case VH_U8: return *reinterpret_cast<const uint8_t*>(base + row);
    case VH_I8: { int8_t v = *reinterpret_cast<const int8_t*>(base + row);
                  return sext ? (uint64_t)(int64_t)v : (uint64_t)(uint8_t)v; }
    case VH_U16: return *reinterpret_cast<const uint16_t*>(base + 2ull * row);
    case VH_I16: { int16_t v = *reinterpret_cast<const int16_t*>(base + 2ull * row);
                   return sext ? (uint64_t)(int64_t)v : (uint64_t)(uint16_t)v; }
    case VH_U32: case VH_F32: return *reinterpret_cast<const uint32_t*>(base + 4ull * row);
    case VH_I32: { int32_t v = *reinterpret_cast<const int32_t*>(base + 4ull * row);
                   return sext ? (uint64_t)(int64_t)v : (uint64_t)(uint32_t)v; }
    default: return *reinterpret_cast<const uint64_t*>(base + 8ull * row);
  }
}

// Type-agnostic half of a gather: the naturally aligned 8-byte word that holds the element (arenas and projections are
// padded, so the word is always inside the allocation) and the bit offset of the element inside it.
__device__ __forceinline__ uint64_t vh_gather_raw(const VhPlanDev& P, uint32_t slot, uint32_t seg, uint32_t row, uint32_t& shift) {
  const uint64_t a = reinterpret_cast<uint64_t>(P.colbase[slot]) + (uint64_t)seg * P.colstride[slot] + (uint64_t)row * P.colpitch[slot];
  shift = ((uint32_t)a & 7u) * 8u;
  return *reinterpret_cast<const uint64_t*>(a & ~7ull);
}
// ... and the typed half: width and extension of the element sitting in the low bits of `v`.
__device__ __forceinline__ uint64_t vh_decode_bits(uint64_t v, int type, bool sext) {
  switch (type) {
    case VH_U8: return v & 0xFFull;
    case VH_I8: return sext ? (uint64_t)(int64_t)(int8_t)v : (v & 0xFFull);
    case VH_U16: return v & 0xFFFFull;
    case VH_I16: return sext ? (uint64_t)(int64_t)(int16_t)v : (v & 0xFFFFull);
    case VH_U32: case VH_F32: return v & 0xFFFFFFFFull;
    case VH_I32: return sext ? (uint64_t)(int64_t)(int32_t)v : (v & 0xFFFFFFFFull);
    default: return v;
  }
}
// One value of row `row` of the column in slot `slot`: out of its arena, or out of a payload projection's record
// (colpitch = record size, colbase = arena + the column's offset inside the record).
__device__ __forceinline__ uint64_t vh_gather(const VhPlanDev& P, uint32_t slot, uint32_t seg, uint32_t row, int type, bool sext) {
  return vh_load_at(P.colbase[slot] + (uint64_t)seg * P.colstride[slot] + (uint64_t)row * P.colpitch[slot], type, sext);
}

// ----------------------------------------------------------- state updates
// One surviving row's contribution to one metric state: TupleStruct::Metrics::Update
// (src/codegen/db/store.cc:131-161): += for SUM/AVG/COUNT (in the metric's own type:
// a narrower integer wraps, which a wider modular accumulator truncated at the end
// reproduces bit for bit), std::max / std::min for MAX / MIN.
template <int SCOPE>
__device__ __forceinline__ void vh_state_update(void* state, uint64_t idx, int sop, uint64_t bits) {
  switch (sop) {
    case SOP_ADD32:
      __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(state) + idx, (uint32_t)bits, __ATOMIC_RELAXED, SCOPE);
      break;
    case SOP_ADD64:
      __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(state) + idx, (unsigned long long)bits,
                             __ATOMIC_RELAXED, SCOPE);
      break;
    case SOP_ADD32P:
      __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(state) + idx,
                             (1ull << 32) | (unsigned long long)(uint32_t)bits, __ATOMIC_RELAXED, SCOPE);
      break;
    case SOP_ADDF32:
      __hip_atomic_fetch_add(reinterpret_cast<float*>(state) + idx, __uint_as_float((uint32_t)bits),
                             __ATOMIC_RELAXED, SCOPE);
      break;
    case SOP_ADDF64:
      __hip_atomic_fetch_add(reinterpret_cast<double*>(state) + idx, __longlong_as_double((long long)bits),
                             __ATOMIC_RELAXED, SCOPE);
      break;
    case SOP_MIN_I32:
      __hip_atomic_fetch_min(reinterpret_cast<int32_t*>(state) + idx, (int32_t)bits, __ATOMIC_RELAXED, SCOPE);
      break;
    case SOP_MAX_I32:
      __hip_atomic_fetch_max(reinterpret_cast<int32_t*>(state) + idx, (int32_t)bits, __ATOMIC_RELAXED, SCOPE);
      break;
    case SOP_MIN_U32:
      __hip_atomic_fetch_min(reinterpret_cast<uint32_t*>(state) + idx, (uint32_t)bits, __ATOMIC_RELAXED, SCOPE);
      break;
    case SOP_MAX_U32:
      __hip_atomic_fetch_max(reinterpret_cast<uint32_t*>(state) + idx, (uint32_t)bits, __ATOMIC_RELAXED, SCOPE);
      break;
    case SOP_MIN_I64:
      __hip_atomic_fetch_min(reinterpret_cast<long long*>(state) + idx, (long long)bits, __ATOMIC_RELAXED, SCOPE);
      break;
    case SOP_MAX_I64:
      __hip_atomic_fetch_max(reinterpret_cast<long long*>(state) + idx, (long long)bits, __ATOMIC_RELAXED, SCOPE);
      break;
    case SOP_MIN_U64:
      __hip_atomic_fetch_min(reinterpret_cast<unsigned long long*>(state) + idx, (unsigned long long)bits,
                             __ATOMIC_RELAXED, SCOPE);
      break;
    case SOP_MAX_U64:
      __hip_atomic_fetch_max(reinterpret_cast<unsigned long long*>(state) + idx, (unsigned long long)bits,
                             __ATOMIC_RELAXED, SCOPE);
      break;
    case SOP_MIN_F32:
      __hip_atomic_fetch_min(reinterpret_cast<float*>(state) + idx, __uint_as_float((uint32_t)bits),
                             __ATOMIC_RELAXED, SCOPE);
      break;
    case SOP_MAX_F32:
      __hip_atomic_fetch_max(reinterpret_cast<float*>(state) + idx, __uint_as_float((uint32_t)bits),
                             __ATOMIC_RELAXED, SCOPE);
      break;
    case SOP_MIN_F64:
      __hip_atomic_fetch_min(reinterpret_cast<double*>(state) + idx, __longlong_as_double((long long)bits),
                             __ATOMIC_RELAXED, SCOPE);
      break;
    case SOP_MAX_F64:
      __hip_atomic_fetch_max(reinterpret_cast<double*>(state) + idx, __longlong_as_double((long long)bits),
                             __ATOMIC_RELAXED, SCOPE);
      break;
    default: break;
  }
}

__host__ __device__ __forceinline__ int vh_sop_bytes(int sop) {
  switch (sop) {
    case SOP_ADD32: case SOP_ADDF32: case SOP_MIN_I32: case SOP_MAX_I32: case SOP_MIN_U32:
    case SOP_MAX_U32: case SOP_MIN_F32: case SOP_MAX_F32: return 4;
    default: return 8;
  }
}
__host__ __device__ __forceinline__ bool vh_sop_sext(int sop) {
  return sop == SOP_MIN_I32 || sop == SOP_MAX_I32 || sop == SOP_MIN_I64 || sop == SOP_MAX_I64;
}

// --------------------------------------------------------------- hash insert
#define VH_HASH_EMPTY 0xFFFFFFFFFFFFFFFFull

// Open-addressing insert of a 64-bit key (never the all-ones sentinel) into `keys`; returns the slot.
// Device-scope CAS on the key word is the only synchronisation.
// PEEK: look at the slot with a plain load first and only CAS an empty one. Worth it where keys repeat (the group table: a
// read-modify-write costs ~2.3x a load on this part and 42 % of C5's survivors find their group already there); not where
// nearly every key is new (the (group, id) set), where it would only add a round trip.
template <bool PEEK = false>
__device__ __forceinline__ uint64_t vh_set_insert64(uint64_t* keys, uint64_t mask, uint32_t max_probe, uint64_t key,
                                                   bool& ok, bool& fresh, uint32_t stride_words = 1) {
  fresh = false;
  uint64_t slot = vh_splitmix64(key) & mask;
  for (uint32_t probe = 0; probe <= max_probe; ++probe) {
    if (PEEK) {
      const unsigned long long seen = __hip_atomic_load(reinterpret_cast<unsigned long long*>(keys) + slot * stride_words, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (seen == key) return slot;
      if (seen != VH_HASH_EMPTY) { slot = (slot + 1) & mask; continue; }
    }
    unsigned long long expect = VH_HASH_EMPTY;
    const bool won = __hip_atomic_compare_exchange_strong(reinterpret_cast<unsigned long long*>(keys) + slot * stride_words, &expect,
                                                          (unsigned long long)key, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT);
    if (won) { fresh = true; return slot; }
    if (expect == key) return slot;
    slot = (slot + 1) & mask;
  }
  ok = false;
  return 0;
}

// Multi-word keys: slot tag word 0 = empty, 1 = being written, 2 = ready.
// A lane that wins the tag writes the key words, releases, and is done inside the same
// loop iteration, so lanes of its own wave that spin on the tag cannot starve it.
__device__ __forceinline__ uint64_t vh_set_insert_wide(uint64_t* keys, uint32_t* tags, uint64_t mask, uint32_t max_probe,
                                                      const uint64_t* key, int kw, bool& ok, bool& fresh) {
  fresh = false;
  uint64_t h = 0x243F6A8885A308D3ull;
  for (int i = 0; i < kw; ++i) h = vh_splitmix64(h ^ key[i]);
  uint64_t slot = h & mask;
  uint32_t probe = 0;
  while (probe <= max_probe) {
    uint32_t tag = __hip_atomic_load(tags + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tag == 0) {
      uint32_t expect = 0;
      if (__hip_atomic_compare_exchange_strong(tags + slot, &expect, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT)) {
        for (int i = 0; i < kw; ++i)
          __hip_atomic_store(keys + slot * kw + i, key[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(tags + slot, 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        fresh = true;
        return slot;
      }
      continue;  // lost the race: re-read the tag
    }
    if (tag == 1) continue;  // writer in flight
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    bool same = true;
    for (int i = 0; i < kw; ++i)
      same &= __hip_atomic_load(keys + slot * kw + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == key[i];
    if (same) return slot;
    slot = (slot + 1) & mask;
    ++probe;
  }
  ok = false;
  return 0;
}

// group table: the table is in HBM (the group count that sends a query here exceeds what LDS or the dense
// table can hold)
__device__ __forceinline__ uint64_t vh_hash_insert64(const VhPlanDev& P, uint64_t key, bool& ok, bool& fresh) {
  fresh = false;
  if (key == VH_HASH_EMPTY) {          // the one key that collides with the sentinel
    atomicOr(P.counters + 3, 1ull);    // marks the reserved extra slot as used
    return P.hmask + 1;
  }
  return vh_set_insert64<true>(P.hkeys, P.hmask, P.max_probe, key, ok, fresh, P.hrec_bytes ? P.hrec_bytes / 8u : 1u);
}
// HBM address of metric m's state of hash slot gid (records of hrec_bytes, or one array per metric)
__device__ __forceinline__ char* vh_hash_state(const VhPlanDev& P, const VhMetricDev& m, uint64_t gid) {
  return static_cast<char*>(m.state) + gid * (P.hrec_bytes ? P.hrec_bytes : (uint32_t)vh_sop_bytes(m.sop()));
}
__device__ __forceinline__ uint64_t vh_hash_insert_wide(const VhPlanDev& P, const uint64_t* key, int kw, bool& ok, bool& fresh) {
  return vh_set_insert_wide(P.hkeys, P.htags, P.hmask, P.max_probe, key, kw, ok, fresh);
}

// Two partial states of one metric into one (the merge of private table copies; the wave-level combining below).
__device__ __forceinline__ uint64_t vh_combine(int sop, uint64_t a, uint64_t b) {
  switch (sop) {
    case SOP_ADD32: return (uint32_t)((uint32_t)a + (uint32_t)b);
    case SOP_ADD64: case SOP_ADD32P: case SOP_BITSET: return a + b;
    case SOP_ADDF32: return __float_as_uint(__uint_as_float((uint32_t)a) + __uint_as_float((uint32_t)b));
    case SOP_ADDF64: return (uint64_t)__double_as_longlong(__longlong_as_double((long long)a) + __longlong_as_double((long long)b));
    case SOP_MIN_I32: return (uint32_t)((int32_t)b < (int32_t)a ? b : a);
    case SOP_MAX_I32: return (uint32_t)((int32_t)a < (int32_t)b ? b : a);
    case SOP_MIN_U32: return (uint32_t)b < (uint32_t)a ? (uint32_t)b : (uint32_t)a;
    case SOP_MAX_U32: return (uint32_t)a < (uint32_t)b ? (uint32_t)b : (uint32_t)a;
    case SOP_MIN_I64: return (int64_t)b < (int64_t)a ? b : a;
    case SOP_MAX_I64: return (int64_t)a < (int64_t)b ? b : a;
    case SOP_MIN_U64: return b < a ? b : a;
    case SOP_MAX_U64: return a < b ? b : a;
    case SOP_MIN_F32: return __uint_as_float((uint32_t)b) < __uint_as_float((uint32_t)a) ? (uint32_t)b : (uint32_t)a;
    case SOP_MAX_F32: return __uint_as_float((uint32_t)a) < __uint_as_float((uint32_t)b) ? (uint32_t)b : (uint32_t)a;
    case SOP_MIN_F64: return __longlong_as_double((long long)b) < __longlong_as_double((long long)a) ? b : a;
    default: return __longlong_as_double((long long)a) < __longlong_as_double((long long)b) ? b : a;
  }
}

// HOT KEYS in the hash organisation. Every surviving row updates its group's record with device-scope atomics; atomics on ONE address are served
// one after the other (~6.5 ns each), so a group that holds a tenth of the rows — 12.5 M of C5h's — costs its query a quarter of a second where
// the reference's unordered_map does not care (src/codegen/db/store.cc:131-161). A wave that finds two neighbouring survivors in one group lets the first
// lane of that group speak for all of the group's lanes: their values are combined across the wave (a butterfly over the members, vh_combine per
// step) and ONE atomic per metric goes out; the others go on as before. The test is a readlane, a compare and a ballot per drain.
__device__ __forceinline__ uint64_t vh_wave_combine(int sop, uint64_t v, bool in) {      // every lane: the members' total (garbage when there is no member)
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint64_t o = __shfl_xor(v, off);
    const bool oin = __shfl_xor((int)in, off) != 0;
    if (oin) { v = in ? vh_combine(sop, v, o) : o; in = true; }
  }
  return v;
}
__device__ __forceinline__ uint64_t vh_hot_lanes(bool active, uint64_t gid) {      // lanes of the wave's hot group (0: no two neighbouring survivors share a group)
  // which group? One that two NEIGHBOURING survivors share (distance 1 or 2): with tens of millions of groups and uniform keys that does not happen,
  // with a group that holds an eighth of the survivors or more it happens in nearly every drain
  const int lane = (int)(threadIdx.x & 63);
  const uint64_t g1 = __shfl_down(gid, 1), g2 = __shfl_down(gid, 2);
  const bool a1 = __shfl_down((int)active, 1) != 0 && lane < 63, a2 = __shfl_down((int)active, 2) != 0 && lane < 62;
  const uint64_t pairs = __ballot(active && ((a1 && g1 == gid) || (a2 && g2 == gid)));
  if (!pairs) return 0ull;
  const uint64_t lg = __shfl(gid, __builtin_ctzll(pairs));
  return __ballot(active && gid == lg);      // (two lanes at least: one atomic on a contended address costs more than the butterfly that saves it — measured ~18 ns per same-address
                                             //  atomic with 256 CUs at it: C5h's COUNT alone 224 ms through the uncombined kernel)
}
// ... and what a WAVE keeps of its hot group between drains (the pre-built scan of the hash organisation): the group's slot and, for up to four
// value metrics and one count-distinct, what the wave's rows have added to them so far — wave-uniform registers, flushed with one atomic each when
// the hot group changes and at the wave's end. A group that holds a tenth of a billion rows then costs a few thousand atomics, not a hundred million.
#define VH_HOT_METRICS 4
struct VhHotAcc {
  uint64_t gid; bool valid; unsigned long long card;
  uint64_t v[VH_HOT_METRICS];
};
__device__ __forceinline__ void vh_hot_flush(const VhPlanDev& P, VhHotAcc& A) {      // (wave-uniform)
  if (!A.valid) return;
  if ((threadIdx.x & 63) == 0) {
    int k = 0;
    for (int j = 0; j < P.nmetric; ++j) {
      const VhMetricDev& m = P.m[j];
      if (m.sop() == SOP_BITSET) { if (A.card) atomicAdd(reinterpret_cast<unsigned long long*>(vh_hash_state(P, m, A.gid)), A.card); continue; }
      if (k < VH_HOT_METRICS) vh_state_update<__HIP_MEMORY_SCOPE_AGENT>(vh_hash_state(P, m, A.gid), 0, m.sop(), A.v[k]);
      ++k;
    }
  }
  A.valid = false;
}

// COUNT DISTINCT: insert every id of the row's set into metric b's (group, id) set; first sight bumps card[gid]
__device__ __forceinline__ void vh_distinct_update(const VhPlanDev& P, int b, unsigned long long* card, uint64_t gid,
                                                   uint32_t seg, uint32_t row, unsigned long long& npairs, unsigned long long* fresh_out = nullptr) {      // fresh_out: the caller adds the row's first sights to card itself (vh_wave_combine)
  const uint64_t* offs = P.bs_offs[b][seg];
  const uint64_t o0 = offs[row], o1 = offs[row + 1];
  const void* vals = P.bs_vals[b][seg];
  uint64_t k = o0;
  if (!P.bs_wide[b]) {
    // Two ids at a time with both first probes in flight (the common shape: a couple of ids per stored row): the
    // CAS round trips overlap instead of queueing behind each other, and the row's fresh pairs bump the group's
    // cardinality with ONE atomic. Whatever does not settle on its first probe takes the general loop below.
    unsigned long long* tab = reinterpret_cast<unsigned long long*>(P.dset_keys[b]);
    const uint32_t* ids = reinterpret_cast<const uint32_t*>(vals);
    unsigned long long nfresh = 0;
    for (; k + 2 <= o1; k += 2) {
      const uint64_t ka = (gid << 32) | ids[k], kb = (gid << 32) | ids[k + 1];
      const uint64_t sa = vh_splitmix64(ka) & P.dset_mask[b], sb = vh_splitmix64(kb) & P.dset_mask[b];
      unsigned long long ea = VH_HASH_EMPTY, eb = VH_HASH_EMPTY;
      const bool wa = __hip_atomic_compare_exchange_strong(tab + sa, &ea, (unsigned long long)ka, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // identical ids in one row would race for the same slot: the second one must see the first one's write
      const bool wb = ka == kb ? false : __hip_atomic_compare_exchange_strong(tab + sb, &eb, (unsigned long long)kb, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      nfresh += (wa ? 1 : 0) + (wb ? 1 : 0);
      if (!wa && ea != ka) { bool ok = true, fresh = false; vh_set_insert64(P.dset_keys[b], P.dset_mask[b], 4096u, ka, ok, fresh); if (!ok) atomicOr(P.counters + 2, VH_ERR_HASH_FULL); nfresh += fresh ? 1 : 0; }
      if (ka != kb && !wb && eb != kb) { bool ok = true, fresh = false; vh_set_insert64(P.dset_keys[b], P.dset_mask[b], 4096u, kb, ok, fresh); if (!ok) atomicOr(P.counters + 2, VH_ERR_HASH_FULL); nfresh += fresh ? 1 : 0; }
    }
    if (nfresh) { if (fresh_out) *fresh_out += nfresh; else atomicAdd(card, nfresh); npairs += nfresh; }
  }
  for (; k < o1; ++k) {
    bool ok = true, fresh = false;
    if (P.bs_wide[b]) {
      const uint64_t key[2] = {gid, reinterpret_cast<const uint64_t*>(vals)[k]};
      vh_set_insert_wide(P.dset_keys[b], P.dset_tags[b], P.dset_mask[b], 4096u, key, 2, ok, fresh);
    } else {
      const uint64_t key = (gid << 32) | reinterpret_cast<const uint32_t*>(vals)[k];   // gid < 2^32 - 1 (host checks)
      vh_set_insert64(P.dset_keys[b], P.dset_mask[b], 4096u, key, ok, fresh);
    }
    if (!ok) atomicOr(P.counters + 2, VH_ERR_HASH_FULL);
    if (fresh) { if (fresh_out) *fresh_out += 1ull; else atomicAdd(card, 1ull); ++npairs; }   // pairs are totalled per lane: one hot-spot atomic per wave, not per pair
  }
}

// ------------------------------------------------------------ scan + aggregate
// (VH_MODE_*: vh_internal.h)

// One surviving row (one per active lane, lanes are dense after compaction):
// build the AggTuple key, then Update every selected metric.
// ---------------------------------------------------------- LDS front table of the hash path
// Low-cardinality sparse keys (GROUP BY a time bucket, a float, ...) would otherwise hammer a handful of HBM slots
// with device-scope atomics (measured: 100 M rows into 25 month buckets = 26 ms, 50x off the roofline). Each block
// therefore keeps a small open-addressing table in LDS; a row whose key finds (or claims) a slot there costs LDS
// atomics only, everything else falls through to the HBM table. A wave that mostly falls through (high-cardinality
// keys: the table fills up at once) stops probing LDS after a warm-up.
struct VhLdsHashWave { uint32_t hits, misses; bool bypass; unsigned long long npairs; /* per lane: fresh (group, id) pairs */ bool dead = false; /* wave-uniform: an insert of this wave found the table full */
                       VhHotAcc hot{0ull, false, 0ull, {0ull, 0ull, 0ull, 0ull}}; /* the wave's hot group between drains (vh_hot_flush at the wave's end) */ };

__device__ __forceinline__ bool vh_lds_hash_find(const VhPlanDev& P, char* lds, uint64_t key, uint32_t& slot_out) {
  unsigned long long* lk = reinterpret_cast<unsigned long long*>(lds + P.lds_hkeys_off);
  const uint32_t smask = P.lds_hash_slots - 1u;
  uint32_t slot = (uint32_t)(vh_splitmix64(key) >> 32) & smask;
  for (int probe = 0; probe < 8; ++probe) {
    unsigned long long old = lk[slot];
    if (old == VH_HASH_EMPTY) old = atomicCAS(&lk[slot], (unsigned long long)VH_HASH_EMPTY, (unsigned long long)key);
    if (old == VH_HASH_EMPTY || old == key) { slot_out = slot; return true; }
    slot = (slot + 1u) & smask;
  }
  return false;
}

__device__ __forceinline__ void vh_lds_hash_init(const VhPlanDev& P, char* lds, int nthreads) {
  for (uint32_t g = threadIdx.x; g < P.lds_hash_slots; g += nthreads) {
    reinterpret_cast<uint64_t*>(lds + P.lds_hkeys_off)[g] = VH_HASH_EMPTY;
    for (int j = 0; j < P.nmetric; ++j) {
      const VhMetricDev& m = P.m[j];
      if (vh_sop_bytes(m.sop()) == 4) reinterpret_cast<uint32_t*>(lds + m.lds_off)[g] = (uint32_t)m.ident;
      else reinterpret_cast<uint64_t*>(lds + m.lds_off)[g] = m.ident;
    }
  }
  __syncthreads();
}

// merge the block's LDS table into the HBM hash table: one insert + one update per (block, group, metric)
__device__ __forceinline__ void vh_lds_hash_flush(const VhPlanDev& P, char* lds, int nthreads) {
  __syncthreads();
  unsigned long long nfresh = 0;
  for (uint32_t g = threadIdx.x; g < P.lds_hash_slots; g += nthreads) {
    const uint64_t key = reinterpret_cast<uint64_t*>(lds + P.lds_hkeys_off)[g];
    if (key == VH_HASH_EMPTY) continue;
    bool ok = true, fresh = false;
    const uint64_t gid = vh_hash_insert64(P, key, ok, fresh);
    if (!ok) { atomicOr(P.counters + 2, VH_ERR_HASH_FULL); continue; }
    nfresh += fresh ? 1 : 0;
    for (int j = 0; j < P.nmetric; ++j) {
      const VhMetricDev& m = P.m[j];
      const uint64_t bits = vh_sop_bytes(m.sop()) == 4 ? reinterpret_cast<uint32_t*>(lds + m.lds_off)[g]
                                                     : reinterpret_cast<uint64_t*>(lds + m.lds_off)[g];
      vh_state_update<__HIP_MEMORY_SCOPE_AGENT>(vh_hash_state(P, m, gid), 0, m.sop(), bits);
    }
  }
  if (nfresh) atomicAdd(P.counters + 1, nfresh);
}

template <int MODE, int SCOPE>
__device__ __forceinline__ void vh_consume(const VhPlanDev& P, uint32_t seg, uint32_t row, bool active,
                                           char* lds, uint64_t xoff, unsigned long long& nfresh, VhLdsHashWave& H) {
  uint64_t gid = 0;
  uint64_t key[VH_KEY_WORDS];
  if (MODE == VH_MODE_HASH) {
#pragma unroll
    for (int i = 0; i < VH_KEY_WORDS; ++i) key[i] = 0;
  }
  bool bad = false;
  if (!active) row = 0;
  for (int i = 0; i < P.ngroup; ++i) {
    const VhGroupDev& g = P.g[i];
    uint64_t v = vh_gather(P, g.slot(), seg, row, g.type(), MODE != VH_MODE_HASH);
    if (g.gran() != VH_T_NONE || g.nroll()) v = vh_time_rollup(v, g);
    if (MODE == VH_MODE_HASH) {
      if (g.type() == VH_F32 && (uint32_t)v == 0x80000000u) v = 0;            // -0.0f == 0.0f
      if (g.type() == VH_F64 && v == 0x8000000000000000ull) v = 0;
      key[g.key_word()] |= v << g.key_shift();
    } else {
      const uint64_t d = v - g.lo;
      bad |= d >= g.extent;
      gid += d * g.stride;
    }
  }
  bool in_lds = false;
  if (MODE == VH_MODE_HASH) {
    if (P.heavy_only) {      // the second pass of a hashed partitioning with heavy ranges: only the rows of those ranges count (VhPlanDev::heavy_only)
      const uint32_t idx = (uint32_t)(vh_splitmix64(key[0]) >> 48);
      active = active && ((P.heavy_only[idx >> 5] >> (idx & 31u)) & 1u) != 0u;
    }
    if (P.lds_hash_slots && !H.bypass) {
      uint32_t ls = 0;
      in_lds = active && key[0] != VH_HASH_EMPTY && vh_lds_hash_find(P, lds, key[0], ls);
      const uint32_t nh = __popcll(__ballot(in_lds)), nm = __popcll(__ballot(active && !in_lds));
      H.hits += nh; H.misses += nm;
      if (H.hits + H.misses >= 4096u && H.misses > H.hits) H.bypass = true;
      if (in_lds) {
        for (int j = 0; j < P.nmetric; ++j) {
          const VhMetricDev& m = P.m[j];
          const uint64_t bits = m.slot() == VH_SLOT_ROWID ? (((uint64_t)seg << 32) | row)
                                                        : vh_gather(P, m.slot(), seg, row, m.type(), vh_sop_sext(m.sop()));
          vh_state_update<__HIP_MEMORY_SCOPE_WORKGROUP>(lds + m.lds_off, ls, m.sop(), bits);
        }
      }
      active = active && !in_lds;
    }
    bool ok = true, fresh = false;
    if (active) {
      gid = P.key_words == 1 ? vh_hash_insert64(P, key[0], ok, fresh)
                             : vh_hash_insert_wide(P, key, P.key_words, ok, fresh);
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
  if (MODE == VH_MODE_DENSE_LDS) {
    if (active) reinterpret_cast<uint8_t*>(lds + P.lds_present_off)[gid] = 1;
  } else if (MODE == VH_MODE_DENSE_GLOBAL) {
    if (active && P.present_carrier < 0) P.present[xoff + gid] = 1;
  }
  // (hash organisation: a wave many of whose survivors fall into one group combines their values across the wave — vh_hot_lanes above — and keeps
  // the total in registers until its hot group changes: H.hot)
  uint64_t hot = MODE == VH_MODE_HASH ? vh_hot_lanes(active, gid) : 0ull;
  if (MODE == VH_MODE_HASH && !hot && H.hot.valid) hot = __ballot(active && gid == H.hot.gid);      // (no two neighbours this time: the group the wave already keeps)
  const int lane_ = (int)(threadIdx.x & 63);
  const bool in_hot = ((hot >> lane_) & 1ull) != 0;
  int nvalue = 0;
  for (int j = 0; j < P.nmetric; ++j) nvalue += P.m[j].sop() != SOP_BITSET;
  const bool keep = hot != 0 && nvalue <= VH_HOT_METRICS;      // (wave-uniform) the combined values stay in the wave's accumulators
  const bool speaks = hot != 0 && !keep && lane_ == __builtin_ctzll(hot);
  if (keep) {
    const uint64_t hg = __shfl(gid, __builtin_ctzll(hot));
    if (!H.hot.valid || H.hot.gid != hg) {
      vh_hot_flush(P, H.hot);
      H.hot.gid = hg; H.hot.valid = true; H.hot.card = 0;
      int k = 0;
      for (int j = 0; j < P.nmetric; ++j) if (P.m[j].sop() != SOP_BITSET) H.hot.v[k++] = P.m[j].sop() == SOP_ADD32 || P.m[j].sop() == SOP_ADD64 || P.m[j].sop() == SOP_ADDF32 || P.m[j].sop() == SOP_ADDF64 ? 0ull : P.m[j].ident;
    }
  }
  int kv = 0;
  for (int j = 0; j < P.nmetric; ++j) {
    const VhMetricDev& m = P.m[j];
    if (m.sop() == SOP_BITSET) {   // slot() is the bitset index, m.state the u64 cardinality per group
      unsigned long long* const card = MODE == VH_MODE_HASH ? reinterpret_cast<unsigned long long*>(vh_hash_state(P, m, gid))
                                                            : reinterpret_cast<unsigned long long*>(m.state) + xoff + gid;
      unsigned long long mine = 0;
      if (active) vh_distinct_update(P, m.slot(), card, MODE == VH_MODE_HASH ? gid : xoff + gid, seg, row, H.npairs, in_hot ? &mine : nullptr);
      if (hot) {       // (wave-uniform) the hot group's first sights of this drain: into the accumulator, or one atomic
        const unsigned long long tot = vh_wave_combine(SOP_ADD64, mine, in_hot);
        if (keep) H.hot.card += tot;
        else if (speaks && tot) atomicAdd(card, tot);
      }
      continue;
    }
    uint64_t bits;
    if (m.slot() == VH_SLOT_ROWID) bits = ((uint64_t)seg << 32) | row;   // storage order of the row (search: first occurrence)
    else bits = vh_gather(P, m.slot(), seg, row, m.type(), vh_sop_sext(m.sop()));
    bool upd = active;
    if (hot) {
      const uint64_t tot = vh_wave_combine(m.sop(), bits, in_hot);
      if (keep) { H.hot.v[kv] = vh_combine(m.sop(), H.hot.v[kv], tot); if (in_hot) upd = false; }
      else if (in_hot) { bits = tot; upd = speaks; }
    }
    ++kv;
    if (upd) {
      if (MODE == VH_MODE_DENSE_LDS) {
        vh_state_update<__HIP_MEMORY_SCOPE_WORKGROUP>(lds + m.lds_off, gid, m.sop(), bits);
      } else if (MODE == VH_MODE_DENSE_GLOBAL) {
        vh_state_update<SCOPE>(m.state, xoff + gid, m.sop(), bits);
      } else {
        vh_state_update<__HIP_MEMORY_SCOPE_AGENT>(vh_hash_state(P, m, gid), 0, m.sop(), bits);
      }
    }
  }
}

template <int BLOCK>
struct VhScanCfg {
  static constexpr int kWaves = BLOCK / 64;
  static constexpr int kStepRows = BLOCK * VH_LANE_ROWS;
  static constexpr int kQueueCap = 64 + 256;  // carry-over (<64) + one sub-step (<=256)
};

// The fused kernel. Grid-stride over work units (unit = unit_rows consecutive rows of one
// segment). Per wave and step: stream the predicate columns (coalesced 16 B/lane loads,
// 4 in flight per column), evaluate the filter to a 16-bit mask per lane, compact the
// passing row offsets into this wave's LDS queue with wave64 ballots + mbcnt prefix
// sums, and whenever 64 survivors are queued let all 64 lanes gather the group/metric
// values of one survivor each and update the aggregate table. No block-wide barrier in
// the loop: queues are per wave.
__device__ __forceinline__ void vh_scan_block_end(const VhPlanDev& P, unsigned long long npassed, unsigned long long nfresh, unsigned long long npairs, uint32_t ext_end);     // (below, with the partition helpers)
template <int MODE, int BLOCK, int SCOPE>
__global__ __launch_bounds__(BLOCK) void scan_agg_kernel(const VhPlanDev P) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  typedef VhScanCfg<BLOCK> C;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  // queues live behind the (optional) LDS aggregate table
  uint32_t* q = reinterpret_cast<uint32_t*>(lds + ((MODE == VH_MODE_DENSE_LDS || MODE == VH_MODE_HASH) ? P.lds_bytes : 0)) +
                wave * C::kQueueCap;
  VhLdsHashWave H{0u, 0u, false, 0ull};
  if (MODE == VH_MODE_HASH && P.lds_hash_slots) vh_lds_hash_init(P, lds, BLOCK);

  if (MODE == VH_MODE_DENSE_LDS) {
    // identities: 0 for SUM/AVG/COUNT, type max for MIN, cpp_min_value for MAX
    // (src/codegen/db/store.cc:107-117)
    for (int j = 0; j < P.nmetric; ++j) {
      const VhMetricDev& m = P.m[j];
      const uint64_t ident = m.ident;
      if (vh_sop_bytes(m.sop()) == 4) {
        for (uint64_t g = threadIdx.x; g < P.G; g += BLOCK)
          reinterpret_cast<uint32_t*>(lds + m.lds_off)[g] = (uint32_t)ident;
      } else {
        for (uint64_t g = threadIdx.x; g < P.G; g += BLOCK)
          reinterpret_cast<uint64_t*>(lds + m.lds_off)[g] = ident;
      }
    }
    for (uint64_t g = threadIdx.x; g < P.G; g += BLOCK) reinterpret_cast<uint8_t*>(lds + P.lds_present_off)[g] = 0;
    __syncthreads();
  }

  const uint64_t xoff = (MODE == VH_MODE_DENSE_GLOBAL && P.nxcd > 1) ? (uint64_t)(vh_xcc_id() % P.nxcd) * P.xcd_stride : 0;
  const uint64_t lanemask_lt = (1ull << lane) - 1ull;
  unsigned long long npassed = 0, nfresh = 0;

  for (uint32_t unit = blockIdx.x; unit < P.total_units; unit += gridDim.x) {
    // An aggregate table that has overflowed makes every further insert walk its whole probe limit (a 1 M-slot table under
    // 37 M groups: seconds instead of milliseconds before the host could re-plan). A wave that has seen an insert fail
    // knows the attempt is void and stops; no wave looks at the global flag (a vector load in the scan loop would drain
    // the prefetched predicate columns behind it: measured 0.4 ms on C3).
    if (MODE == VH_MODE_HASH && H.dead) break;
    const uint32_t seg = unit / P.units_per_seg;
    const uint32_t unit_base = (unit - seg * P.units_per_seg) * P.unit_rows;
    const uint32_t seg_rows = P.seg_rows[seg];
    if (unit_base >= seg_rows) continue;
    uint32_t cnt = 0;  // queued survivors of this wave (wave-uniform)
    for (uint32_t step_base = unit_base; step_base < unit_base + P.unit_rows && step_base < seg_rows;
         step_base += C::kStepRows) {
      const uint32_t wave_base = step_base + wave * VH_WAVE_STEP_ROWS;
      if (wave_base >= seg_rows) break;
      const uint32_t row_l = wave_base + lane * 4;
      const bool full = wave_base + VH_WAVE_STEP_ROWS <= seg_rows;
      const uint32_t mask = full ? vh_eval_filter<true>(P, seg, row_l, seg_rows)
                                 : vh_eval_filter<false>(P, seg, row_l, seg_rows);
      npassed += __popc(mask);
#pragma unroll
      for (int k = 0; k < VH_SUBSTEPS; ++k) {
        const uint32_t mk = (mask >> (4 * k)) & 0xFu;
        if (__ballot(mk != 0) == 0) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool b = (mk >> j) & 1u;
          const uint64_t bal = __ballot(b);
          if (b) q[cnt + __popcll(bal & lanemask_lt)] = row_l + k * 256u + j;
          cnt += __popcll(bal);
        }
        __builtin_amdgcn_wave_barrier();
        while (cnt >= 64) {
          cnt -= 64;
          const uint32_t r = q[cnt + lane];
          vh_consume<MODE, SCOPE>(P, seg, r, true, lds, xoff, nfresh, H);
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
    if (cnt) {
      const bool act = lane < (int)cnt;
      const uint32_t r = act ? q[lane] : 0;
      vh_consume<MODE, SCOPE>(P, seg, r, act, lds, xoff, nfresh, H);
      __builtin_amdgcn_wave_barrier();
    }
  }

  // wave totals -> one atomic per wave
  if (MODE == VH_MODE_HASH) vh_hot_flush(P, H.hot);      // what the wave still keeps of its hot group
  for (int off = 32; off > 0; off >>= 1) { npassed += __shfl_down(npassed, off); H.npairs += __shfl_down(H.npairs, off); }
  vh_scan_block_end(P, npassed, nfresh, H.npairs, 0u);     // (one set of atomics per block; this kernel writes no tuples)
  if (MODE == VH_MODE_HASH && P.lds_hash_slots) vh_lds_hash_flush(P, lds, BLOCK);

  if (MODE == VH_MODE_DENSE_LDS) {
    // flush this block's LDS table: one global update per (block, present group)
    __syncthreads();
    const uint64_t xo = P.nxcd > 1 ? (uint64_t)(vh_xcc_id() % P.nxcd) * P.xcd_stride : 0;
    for (uint64_t g = threadIdx.x; g < P.G; g += BLOCK) {
      if (!reinterpret_cast<uint8_t*>(lds + P.lds_present_off)[g]) continue;
      P.present[xo + g] = 1;
      for (int j = 0; j < P.nmetric; ++j) {
        const VhMetricDev& m = P.m[j];
        const uint64_t bits = vh_sop_bytes(m.sop()) == 4 ? reinterpret_cast<uint32_t*>(lds + m.lds_off)[g]
                                                       : reinterpret_cast<uint64_t*>(lds + m.lds_off)[g];
        vh_state_update<SCOPE>(m.state, xo + g, m.sop(), bits);
      }
    }
  }
}



// ------------------------------------------------- partitioned aggregation, phase 1
// Global atomics execute at the memory side on this part (per-XCD L2s are not coherent with each other): ~23 G
// read-modify-writes per second whatever the table size (profiles/r01/NOTES.md, "The atomic roofline"). For group-id
// spaces that do not fit one CU's LDS the survivors are therefore NOT aggregated with global atomics: each becomes a
// (gid, values) tuple appended to its partition's current extent in HBM; part_agg_kernel then aggregates every
// partition with LDS atomics only. The compiled kernels' first attempt writes through the block's ring writer (vh_ring_add_tb below:
// extents by position); re-runs and the pre-built kernels use the two older writers:
//  - the compaction kernels hand the 64 survivors of one drain to vh_part_direct_add: a ballot per partition-id bit
//    gives every lane its rank inside its partition, the partition's cursor is fetched from the lane that owns it
//    (ds_bpermute) and each lane stores its tuple with one 16-byte store — no LDS tile, no second pass;
//  - the lanes form (no compaction, survivors arrive a few per step) collects VH_PART_TILE tuples in LDS,
//    counting-sorts the tile by partition (histogram -> wave prefix -> scatter) and writes every partition's run as
//    one contiguous, coalesced store (vh_part_tile_write / vh_part_tile_runs).
// Per-partition state (current extent, fill) lives in lane p's registers and is handed out with v_readlane: an
// earlier version that went through LDS arrays lost a tuple now and then when an extent was opened inside a tile.
#define VH_PART_TILE 256     // tuples per wave tile
struct VhPartWave {
  uint32_t chunk_next, chunk_end;   // extents this wave has reserved and not yet opened
  uint32_t base, limit;             // level 2: the range of pool-2 extents of the partition being split ... (level 1 with chunks by position,
                                    // VhPlanDev::ext_waves: base = chunks taken so far, limit = this wave's number in the launch)
  uint32_t* cursor;                 // ... and its allocation cursor (VhPlanDev::l2)
};
struct VhPartTile {
  uint64_t* sorted;    // LDS [VH_PART_TILE][tw], lanes form only: the tile's tuples ordered by partition
  uint32_t* hist;      // LDS [64], lanes form only: counts, then scatter cursors
  uint32_t r_ext;      // lane p: partition p's current extent (~0u: none) ...
  uint32_t r_fill;     // ... and the tuples already in it
};
// LEVEL 1: phase 1 writes pool 1 (partition = gid >> part_shift); LEVEL 2: part_split_kernel writes pool 2 (sub-partition 0..63)
template <int LEVEL> __device__ __forceinline__ uint64_t* vh_pool_tuples(const VhPlanDev& P) { return LEVEL == 1 ? P.tuples : P.tuples2; }
template <int LEVEL> __device__ __forceinline__ uint16_t* vh_pool_missing(const VhPlanDev& P) { return LEVEL == 1 ? P.extent_missing : P.extent_missing2; }
template <int LEVEL> __device__ __forceinline__ uint8_t* vh_pool_tags(const VhPlanDev& P) { return LEVEL == 1 ? P.extent_part : P.extent_part2; }
template <int LEVEL> __device__ __forceinline__ uint32_t vh_pool_et(const VhPlanDev& P) { return (uint32_t)(LEVEL == 1 ? P.ext_tuples : P.ext_tuples2); }
template <int LEVEL> __device__ __forceinline__ uint32_t vh_pool_es(const VhPlanDev& P) { return LEVEL == 1 ? (uint32_t)P.ext_stride : vh_pool_et<LEVEL>(P); }   // tuples between extent starts
template <int LEVEL> __device__ __forceinline__ uint32_t vh_pool_npart(const VhPlanDev& P) { return LEVEL == 1 ? (uint32_t)P.npart : 64u; }
// the two pools as run-time values (plan / split kernels)
struct VhPools {
  uint64_t* t1; uint16_t* miss1; uint8_t* tag1; uint32_t max1; unsigned long long allocated1;
  uint64_t* t2; uint16_t* miss2; uint8_t* tag2; uint32_t max2; uint32_t* l2;
};
__device__ __forceinline__ VhPools vh_pools(const VhPlanDev& P) {
  VhPools Q;
  Q.t1 = P.tuples; Q.miss1 = P.extent_missing; Q.tag1 = P.extent_part; Q.max1 = P.max_extents; Q.allocated1 = P.counters[5];
  Q.t2 = P.tuples2; Q.miss2 = P.extent_missing2; Q.tag2 = P.extent_part2; Q.max2 = P.max_extents2; Q.l2 = P.l2;
  return Q;
}
__host__ __device__ __forceinline__ size_t vh_part_tile_bytes(const VhPlanDev& P) {
  return ((size_t)VH_PART_TILE * P.tw * 8 + 64 * 4 + 15) / 16 * 16;
}
__device__ __forceinline__ void vh_part_tile_init(const VhPlanDev& P, char* area, VhPartTile& T, VhPartWave& W) {
  T.sorted = reinterpret_cast<uint64_t*>(area);      // (the compacting kernels pass no area: they only use the register state)
  T.hist = reinterpret_cast<uint32_t*>(area + (size_t)VH_PART_TILE * P.tw * 8);
  T.r_ext = ~0u; T.r_fill = 0;
  W.chunk_next = W.chunk_end = 0;
  W.base = 0; W.limit = P.ext_waves ? blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6) : 0u; W.cursor = nullptr;
}
// Phase 1, chunks by position: what the shared cursor would have said at the end for THIS wave (0: it took no chunk) — vh_scan_block_end
// publishes the block's largest.
__device__ __forceinline__ uint32_t vh_part_wave_end(const VhPlanDev& P, const VhPartWave& W) {
  return P.ext_waves && W.base != 0 ? (W.chunk_end < P.max_extents ? W.chunk_end : P.max_extents) : 0u;
}
// The end of a scan block: its waves' row counters (lane 0 of every wave holds its wave's) as ONE set of device atomics per block. Atomics
// of every wave on the same few words are served one after the other (~6.5 ns each), and the kernel is not over before the last one: 3 072
// waves x 2 words = 40 us at the end of every C3 scan, 8 192 x 1 = 53 us of C1's 68 (tools/r04/run26.sh, profiles/r04/NOTES.md).
__device__ __forceinline__ void vh_scan_block_end(const VhPlanDev& P, unsigned long long npassed, unsigned long long nfresh, unsigned long long npairs, uint32_t ext_end) {
  __shared__ unsigned long long s_cnt[3][16];
  __shared__ uint32_t s_ext[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (lane == 0) { s_cnt[0][wave] = npassed; s_cnt[1][wave] = nfresh; s_cnt[2][wave] = npairs; s_ext[wave] = ext_end; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long a = 0, b = 0, c = 0;
    uint32_t e = 0;
    for (int w = 0; w < nw; ++w) { a += s_cnt[0][w]; b += s_cnt[1][w]; c += s_cnt[2][w]; e = s_ext[w] > e ? s_ext[w] : e; }
    if (a) atomicAdd(P.counters + 0, a);
    if (b) atomicAdd(P.counters + 1, b);
    if (c) atomicAdd(P.counters + 4, c);
    if (e) atomicMax(P.counters + 5, (unsigned long long)e);
  }
}

// Open a new extent for partition p (wave-uniform). Returns ~0u when the buffer is exhausted.
template <int LEVEL>
__device__ __forceinline__ uint32_t vh_part_new_extent(const VhPlanDev& P, VhPartWave& W, int p, int lane) {
  if (W.chunk_next == W.chunk_end) {
    if (LEVEL == 1 && P.ext_waves) {
      const unsigned long long c = ((unsigned long long)W.base * P.ext_waves + W.limit) * VH_EXT_CHUNK;
      W.chunk_next = c < 0xFFFFFF00ull && W.limit < P.ext_waves ? (uint32_t)c : 0xFFFFFF00u;      // (beyond any pool, or a launch that is not the one the host
                                                                                                      //  laid the chunks out for: refused below, the re-run takes the cursor)
      ++W.base;
    } else if (LEVEL == 1) {
      unsigned long long c = 0;
      if (lane == 0) c = atomicAdd(P.counters + 5, (unsigned long long)VH_EXT_CHUNK);
      c = __shfl(c, 0);
      W.chunk_next = (uint32_t)c;
    } else {
      uint32_t c = 0;
      if (lane == 0) c = atomicAdd(W.cursor, (uint32_t)VH_EXT_CHUNK);
      c = __shfl(c, 0);
      W.chunk_next = c > W.limit - W.base ? W.limit : W.base + c;      // (a saturated cursor must not wrap into another range)
    }
    W.chunk_end = W.chunk_next + VH_EXT_CHUNK;
  }
  const uint32_t ext = W.chunk_next++;
  const bool ok = ext < (LEVEL == 1 ? P.max_extents : W.limit);
  if (ok && lane == 0) vh_pool_tags<LEVEL>(P)[ext] = (uint8_t)p;
  if (!ok) {
    if (lane == 0) atomicOr(P.counters + 2, VH_ERR_PART_FULL);  // the host re-runs with a larger tuple buffer
    return ~0u;
  }
  return ext;
}

template <int LEVEL>
__device__ __forceinline__ void vh_part_tile_runs(const VhPlanDev& P, VhPartTile& T, VhPartWave& W, uint32_t cnt, uint32_t base, int lane);
// One tile: every lane brings up to four tuples (part[r] == ~0u: none) in registers; they leave for HBM grouped by partition.
template <int NW, int LEVEL = 1>
__device__ __forceinline__ void vh_part_tile_write(const VhPlanDev& P, VhPartTile& T, VhPartWave& W, const uint64_t (&words)[4][NW],
                                                   const uint32_t (&part)[4], int lane) {
  uint32_t* hist = T.hist;
  hist[lane] = 0;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (part[r] != 0xFFFFFFFFu) atomicAdd(&hist[part[r]], 1u);
  __builtin_amdgcn_wave_barrier();
  // lane p owns partition p: count, exclusive prefix (wave scan), cursor
  const uint32_t cnt = hist[lane];                       // lanes >= npart read 0
  uint32_t incl = cnt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(incl, off); if (lane >= off) incl += o; }
  const uint32_t base = incl - cnt;
  __builtin_amdgcn_wave_barrier();
  hist[lane] = base;                                     // becomes the scatter cursor
  __builtin_amdgcn_wave_barrier();
  const uint32_t tw = (uint32_t)P.tw;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (part[r] != 0xFFFFFFFFu) {
      const uint32_t pos = atomicAdd(&hist[part[r]], 1u);
#pragma unroll
      for (int w = 0; w < NW; ++w)
        if ((uint32_t)w < tw) T.sorted[pos * tw + w] = words[r][w];
    }
  }
  vh_part_tile_runs<LEVEL>(P, T, W, cnt, base, lane);
}

// Run write-out shared by both forms: T.sorted holds the tile ordered by partition; lane p holds partition p's count and
// exclusive prefix.
template <int LEVEL>
__device__ __forceinline__ void vh_part_tile_runs(const VhPlanDev& P, VhPartTile& T, VhPartWave& W, uint32_t cnt, uint32_t base, int lane) {
  const uint32_t tw = (uint32_t)P.tw;
  // room in the partitions' current extents? (lane p decides for partition p; a run never straddles extents)
  const uint32_t et = vh_pool_et<LEVEL>(P);
  uint64_t* const pool = vh_pool_tuples<LEVEL>(P);
  uint64_t need = __ballot(cnt != 0 && (T.r_ext == ~0u || T.r_fill + cnt > et));
  while (need) {
    const int p = __builtin_ctzll(need);
    need &= need - 1;
    const uint32_t old = __builtin_amdgcn_readlane(T.r_ext, p), oldfill = __builtin_amdgcn_readlane(T.r_fill, p);
    if (old != ~0u && lane == 0) vh_pool_missing<LEVEL>(P)[old] = (uint16_t)(et - oldfill);
    const uint32_t ext = vh_part_new_extent<LEVEL>(P, W, p, lane);
    if (lane == p) { T.r_ext = ext; T.r_fill = 0; }
  }
  // lane p holds (extent, fill, base, count) of partition p, the loop is wave-uniform over the partitions present in this tile
  const uint64_t mydst = T.r_ext == ~0u ? ~0ull : (uint64_t)T.r_ext * vh_pool_es<LEVEL>(P) + T.r_fill;
  if (T.r_ext != ~0u) T.r_fill += cnt;
  uint64_t runs = __ballot(cnt != 0);
  __builtin_amdgcn_wave_barrier();
  while (runs) {
    const int p = __builtin_ctzll(runs);
    runs &= runs - 1;
    const uint32_t pb = __builtin_amdgcn_readlane(base, p), pc = __builtin_amdgcn_readlane(cnt, p);
    const uint64_t pd = ((uint64_t)__builtin_amdgcn_readlane((uint32_t)(mydst >> 32), p) << 32) | __builtin_amdgcn_readlane((uint32_t)mydst, p);
    if (pd == ~0ull) continue;                           // tuple buffer exhausted: the host re-runs (VH_ERR_PART_FULL)
    if (tw == 2) {                                       // the common shape (gid + one 32-bit and one 64-bit value): 16 B per lane
      typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
      for (uint32_t i = lane; i < pc; i += 64)
        *reinterpret_cast<u64x2*>(pool + (pd + i) * 2) = *reinterpret_cast<const u64x2*>(T.sorted + (size_t)(pb + i) * 2);
    } else {
      for (uint32_t i = lane; i < pc * tw; i += 64) pool[pd * tw + i] = T.sorted[(size_t)pb * tw + i];
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// Compacting kernels: no tile at all. The 64 survivors of a drain find their peers of the same partition
// with one ballot per partition-id bit (rank = mbcnt among the peers), fetch their partition's write cursor from the
// lane that owns it (two ds_bpermute) and store their 16-byte tuple straight into the extent: ~60 VALU and no LDS
// atomics, barriers or staging per drain. A partition's tuples of one drain are adjacent (a few tens of bytes); the next
// drain of the wave continues the same line, and L2 merges the pieces before the line leaves for HBM (a wave keeps
// npart lines open: 4096 waves x 13 partitions x 128 B = 7 MB over eight L2s). Measured against collecting 256 tuples in LDS
// and counting-sorting them like the lanes form does: 4.16 vs 4.38 ms on C3 (profiles/r02/NOTES.md).
// TWC: the tuple width when the caller knows it at compile time (the per-query kernels of vh_jit.hip), 0 = P.tw
template <int NW, int LEVEL = 1, int TWC = 0>
__device__ __forceinline__ void vh_part_direct_add(const VhPlanDev& P, VhPartTile& T, VhPartWave& W, bool active,
                                                   const uint64_t (&words)[NW], uint32_t p, int lane) {
  const uint32_t npart = vh_pool_npart<LEVEL>(P), et = vh_pool_et<LEVEL>(P), tw = TWC ? (uint32_t)TWC : (uint32_t)P.tw;
  const uint64_t act = __ballot(active);
  uint64_t peers = act, mine = act;          // lanes in my survivor's partition / lanes in the partition this lane OWNS
#pragma unroll
  for (int b = 0; b < 6; ++b) {
    if ((npart - 1u) >> b) {                 // wave-uniform: only the bits partition ids use
      const uint64_t bal = __ballot((p >> b) & 1u);
      peers &= ((p >> b) & 1u) ? bal : ~bal;
      mine &= (((uint32_t)lane >> b) & 1u) ? bal : ~bal;
    }
  }
  const uint32_t rank = __popcll(peers & ((1ull << lane) - 1ull));
  const uint32_t cnt = (uint32_t)lane < npart ? (uint32_t)__popcll(mine) : 0u;
  if (VH_ABLATE & 16) { if (rank + cnt == 0x12345678u) P.counters[7] = rank; return; }      // measurement build: ranks and counts, nothing behind them
  // room in the owned partition's extent? (a drain's tuples of one partition never straddle extents)
  uint64_t need = __ballot(cnt != 0 && (T.r_ext == ~0u || T.r_fill + cnt > et));
  while (need) {
    const int q = __builtin_ctzll(need);
    need &= need - 1;
    const uint32_t old = __builtin_amdgcn_readlane(T.r_ext, q), oldfill = __builtin_amdgcn_readlane(T.r_fill, q);
    if (old != ~0u && lane == 0) vh_pool_missing<LEVEL>(P)[old] = (uint16_t)(et - oldfill);
    const uint32_t ext = vh_part_new_extent<LEVEL>(P, W, q, lane);
    if (lane == q) { T.r_ext = ext; T.r_fill = 0; }
  }
  const uint32_t pe = (uint32_t)__shfl((int)T.r_ext, (int)(active ? p : 0u)), pf = (uint32_t)__shfl((int)T.r_fill, (int)(active ? p : 0u));
  if (T.r_ext != ~0u) T.r_fill += cnt;
  if (active && pe != ~0u) {                 // ~0: tuple buffer exhausted, the host re-runs (VH_ERR_PART_FULL)
    uint64_t* d = vh_pool_tuples<LEVEL>(P) + ((uint64_t)pe * vh_pool_es<LEVEL>(P) + pf + rank) * tw;
    if (VH_ABLATE & 4) { if (words[0] == 0x123456789ABCDEFull) d[0] = 1; }   // measurement build: everything but the tuple store
    else if (tw == 2) {
      typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
      u64x2 v; v.x = words[0]; v.y = words[1];
      if (VH_ABLATE & 8) __builtin_nontemporal_store(v, reinterpret_cast<u64x2*>(d));
      else *reinterpret_cast<u64x2*>(d) = v;
    } else if (tw == 4) {
      typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
      u64x2 v0, v1; v0.x = words[0]; v0.y = words[1]; v1.x = words[NW > 2 ? 2 : 0]; v1.y = words[NW > 3 ? 3 : 0];   // four-word tuples: two 16-byte stores
      if (VH_ABLATE & 8) { __builtin_nontemporal_store(v0, reinterpret_cast<u64x2*>(d)); __builtin_nontemporal_store(v1, reinterpret_cast<u64x2*>(d) + 1); }
      else { reinterpret_cast<u64x2*>(d)[0] = v0; reinterpret_cast<u64x2*>(d)[1] = v1; }
    } else {
#pragma unroll
      for (int w = 0; w < NW; ++w)
        if ((uint32_t)w < tw) d[w] = words[w];
    }
  }
}

// ------------------------------------------------- tuples that leave as whole 128-byte lines (the ring writer below)
#ifndef VJ_ABL
#define VJ_ABL 0     // measurement builds of the compiled kernels only (vh_jit_body.h)
#endif
#define VH_RING_PARTS 16                          // DENSE_PART through the ring writer: partitions a block keeps waiting lines for — the small form ...
#define VH_RING_PARTS_MAX 64                      // ... and the wide one (group-id spaces of 17-64 LDS-sized ranges)
typedef uint64_t vh_u64x2 __attribute__((ext_vector_type(2)));
// TW = 64-bit words per tuple: 2 (eight tuples per line) or 1 (sixteen: the planner packed gid and values into one word, VhPlanDev::gid_bits)
template <int TW> struct VhStageTuple;
template <> struct VhStageTuple<2> { typedef vh_u64x2 type; static __device__ __forceinline__ type make(const uint64_t (&w)[2]) { type v; v.x = w[0]; v.y = w[1]; return v; } };
template <> struct VhStageTuple<1> { typedef uint64_t type; static __device__ __forceinline__ type make(const uint64_t (&w)[1]) { return w[0]; } };

template <int LEVEL = 1>
__device__ __forceinline__ void vh_part_tile_finish(const VhPlanDev& P, VhPartTile& T, int lane) {   // open extents are closed with what they hold
  const uint32_t et = vh_pool_et<LEVEL>(P);
  if (T.r_ext != ~0u && T.r_fill < et) vh_pool_missing<LEVEL>(P)[T.r_ext] = (uint16_t)(et - T.r_fill);
}

// ------------------------------------------------- phase 2's blocks by the partitions' tuple counts
// With skewed group keys one partition of DENSE_PART receives a multiple of its share of the tuples (C3 with a Zipf-like first group column: 62 % in
// one of 13), and with the same number of phase-2 blocks per partition that partition's blocks work eight times as long as the others'. Where phase 1
// counted its tuples per partition (the ring writer does, at the blocks' ends: VhPlanDev::part_count) the `blocks` blocks of phase 2 — and with them the
// private copies of a partition's range in HBM — are shared out in proportion: every partition one block, the rest by count (rounded down), what
// rounding left over one each to the first partitions with tuples, never more than `cap` (the copies there is memory for). Computed redundantly by wave 0
// of every phase-2 block and by every block of the merge kernel from the same counts: the same answer everywhere, no launch of its own.
// Wave-level (all 64 lanes): lane p returns partition p's share; *start = the blocks before it.
#define VH_PART_COUNT_WAYS 8u      // counters per partition (vh_ring_finish_tb spreads the blocks over them)
__device__ __forceinline__ uint32_t vh_part_count_of(const uint32_t* count, int npart, int lane) {      // lane p: partition p's tuples (the sum of its counters)
  if (lane >= npart) return 0u;
  const uint4 a = reinterpret_cast<const uint4*>(count)[2 * lane], b = reinterpret_cast<const uint4*>(count)[2 * lane + 1];
  return a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
}
__device__ __forceinline__ uint32_t vh_part_shares(uint32_t c, int npart, uint32_t blocks, uint32_t cap, int lane, uint32_t* start) {      // c: vh_part_count_of
  unsigned long long tot = c;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
  uint32_t share = lane < npart ? 1u : 0u;
  const uint32_t spare = blocks > (uint32_t)npart ? blocks - (uint32_t)npart : 0u;
  if (tot && c) share += (uint32_t)((unsigned long long)spare * c / tot);
  if (share > cap) share = cap;
  uint32_t sum = share;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
  const uint32_t left = blocks > sum ? blocks - sum : 0u;
  const bool elig = c != 0 && share < cap;
  const uint64_t mask = __ballot(elig);
  if (elig && (uint32_t)__popcll(mask & ((1ull << lane) - 1ull)) < left) ++share;
  uint32_t incl = share;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(incl, off); if (lane >= off) incl += o; }
  *start = incl - share;
  return share;
}
// Which (partition, block of the partition, blocks of the partition) a phase-2 block is. false: the block has nothing to do.
// (my_count: vh_part_count_of for the block's first 64 threads, loaded by the caller BEFORE it fills its LDS tables, so that the loads' latency is
// not the first thing every block of the launch waits for)
__device__ __forceinline__ bool vh_part_my_share(const VhPlanDev& P, int blocks_per_part, uint32_t my_count, int& part, int& b, int& bpp) {
  if (!P.part_count || P.nlevel != 1) { part = blockIdx.x / blocks_per_part; b = blockIdx.x % blocks_per_part; bpp = blocks_per_part; return true; }
  __shared__ int s_mine[3];
  if (threadIdx.x == 0) s_mine[0] = -1;
  __syncthreads();
  if (threadIdx.x < 64) {
    uint32_t start = 0;
    const uint32_t share = vh_part_shares(my_count, P.npart, gridDim.x, (uint32_t)P.nxcd, (int)threadIdx.x, &start);
    if ((int)threadIdx.x < P.npart && blockIdx.x >= start && blockIdx.x < start + share) { s_mine[0] = (int)threadIdx.x; s_mine[1] = (int)(blockIdx.x - start); s_mine[2] = (int)share; }
  }
  __syncthreads();
  part = s_mine[0]; b = s_mine[1]; bpp = s_mine[2];
  return part >= 0;
}

// ------------------------------------------------- the ring writer: FAN partitions per BLOCK, whole lines, no barriers
// What every tuple of the partitioning organisations is written with: DENSE_PART's phase 1 (vj_part_ring_add) and second split
// (part_split_ring_kernel), the hashed partitioning's level A — written by the scan itself (vj_fan_add, vh_jit_body.h) — and its barrier-free
// level B (hp_ring_scatter_kernel). Per digit d the block keeps, in LDS,
//   pos[d]   tuples it has appended to digit d so far — a tuple's number `my` comes off it with one returning LDS atomic and says everything:
//            its 128-byte line my / LINE of the (block, digit) stream, its place in the line, and (through the caller's Dest) the extent
//            and line the line goes to;
//   ring     VH_RING_LINES waiting lines: the tuple is written to line (my / LINE) % LINES once that ring place has seen its previous line
//            leave (gen[d][r] counts the lines that left; a lane whose place is still taken — more than LINES * LINE tuples of one digit in
//            flight among the block's waves: rare with mixed keys — tries again in the next round of its wave);
//   done     tuples written into the waiting line: whoever writes the last one owns the line's way out. The owners of one call (about eight
//            of 64 lanes) put (ring line, destination) into the wave's list and the WAVE copies the lines out, eight lanes per 128-byte
//            line: HBM only ever sees whole aligned lines, except for each digit's last, at the block's end (vh_ring_finish).
// WHERE a line goes. The (block, digit) stream is cut into extents of 2^et_shift tuples. Its first Dest::kmax extents lie at POSITIONS the
// Dest computes (extent(d, k): no allocation, no atomics — room for the stream's share of uniformly spread keys and a little more). A stream
// that outgrows its positions — a hot key, clustered survivors — takes further extents from the pool's SHARED overflow region through one global
// atomic per extent (Dest::ovf: a cursor that starts behind the positional extents; consumers find such extents by their tags). The block's waves
// agree on "extent k of digit d" through ovf[d][k & 1] in LDS (k + 1 << 32 | extent): with two waiting lines per digit at most two consecutive
// lines — hence at most extents k and k + 1 — are ever being resolved at once, and only line OWNERS resolve, after their line is complete. An
// owner that finds the entry missing takes an extent and publishes it with a compare-and-swap; a loser of that race drops its extent unused (its
// tag stays "never opened"). Only when the overflow region is exhausted too is the attempt void (VH_ERR_PART_FULL: the host re-plans with a
// bigger pool — the same writer).
// ORDERING (VERDICT r05 #6 / ADVICE r05; pinned by tests/test_jit_compile.py::test_ring_writer_instruction_order). No wave ever waits for a
// barrier. What the protocol needs: (1) a lane's tuple is in the ring before its `done` count is visible, (2) the owner's reads of the line
// come behind the count that made it the owner, (3) its `done` reset and `gen` store come behind those reads, in that order. It rests on the
// hardware executing the LDS (DS) instructions of ONE wave in issue order — the LDS unit of a CU is a single in-order pipeline per wave's
// queue; `lgkmcnt` counts DS operations and they return in order (CDNA3/CDNA4 ISA guide, "Data Share: LDS instructions are issued in order
// and complete in order", and section 4.4 on `s_waitcnt lgkmcnt` being decrement-in-order for LDS-only sequences) — and on an LDS atomic being
// one indivisible step of that unit. This holds on gfx90a / gfx942 / gfx950 (every CDNA part); it is NOT a HIP memory-model guarantee: the
// atomics are relaxed workgroup-scope and the plain ring stores are ordinary LDS stores. What the COMPILER must not reorder is fenced with
// signal fences (no instructions), and the emitted order is asserted on the disassembly by the test named above.
// A kernel must end whatever happens: a lane that waits longer than 2^22 rounds for its ring place gives up, voids the attempt
// (VH_ERR_PART_FULL) and raises the block's `dead` word, which every waiting lane of every wave reads in its wait loop — so one lost line
// costs one bound, not one bound per later tuple of that ring place.
struct VhRing { uint32_t* pos; uint32_t* done; uint32_t* gen; uint32_t* dead; unsigned long long* ovf; char* ring; uint64_t* list; };
// The shared overflow region of a pool, as a Dest carries it: extents [base, base + cap) are handed out through *cur32 / *cur64 (extents taken
// so far, counted from `base`; the counter starts at the number of positional extents, so that consumers that go by "extents used" see them
// all). fill / tag: the pool's per-extent arrays; missing: fill holds what an extent LACKS (DENSE_PART's pools), not what it holds.
struct VhRingOvf {
  uint32_t base, cap;
  unsigned int* cur32; unsigned long long* cur64;
  uint16_t* fill; uint8_t* tag;
  __device__ __forceinline__ uint64_t take() const {
    const unsigned long long got = cur32 ? (unsigned long long)atomicAdd(cur32, 1u) : cur64 ? atomicAdd(cur64, 1ull) : ~0ull;
    return got < (unsigned long long)cap ? (uint64_t)base + got : ~0ull;
  }
};
#define VH_RING_FAN VJ_FAN
#define VH_RING_LINES VJ_FAN_RING
// extent k >= Dest::kmax of digit d: looked up in / published to the block's table (see WHERE above). ~0ull: the pool has no extent left.
template <bool MISSING, class Dest>
__device__ __forceinline__ uint64_t vh_ring_ovf_extent(const VhRing& F, const Dest& dest, uint32_t d, uint32_t k, uint32_t et) {
  unsigned long long* const slot = F.ovf + 2u * d + (k & 1u);
  unsigned long long cur = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  for (;;) {
    if ((uint32_t)(cur >> 32) == k + 1u) return (uint64_t)(uint32_t)cur;
    const uint64_t e = dest.ovf.take();
    if (e == ~0ull) return ~0ull;
    const unsigned long long want = ((unsigned long long)(k + 1u) << 32) | (unsigned long long)e;
    const unsigned long long seen = atomicCAS(slot, cur, want);
    if (seen == cur) {      // mine is the (block, digit)'s extent k: every extent but a stream's last is full when the block ends (vh_ring_finish_tb corrects the last)
      dest.ovf.tag[e] = (uint8_t)d;
      dest.ovf.fill[e] = (uint16_t)(MISSING ? 0u : et);
      return e;
    }
    cur = seen;             // (another owner of the same extent was first; the extent taken here stays unused, its tag "never opened")
  }
}
// TB bytes per tuple: 4 or 8 (DENSE_PART's one-word tuples), 16 or 32; FAN digits and LINES waiting lines per digit (powers of two, FAN * LINES <= 1024:
// a ring line's number takes ten bits of a list entry); et_shift: log2 of the tuples an extent of the pool holds; stride: tuples between extent
// starts (a whole number of 128-byte lines). The hashed partitioning: 256 digits x 2 lines; DENSE_PART: 16 or 64 x 2.
template <int TB, class Dest, int FAN = VH_RING_FAN, int R = VH_RING_LINES, bool MISSING = false>
__device__ __forceinline__ void vh_ring_add_tb(const VhRing& F, char* pool, uint32_t stride, uint32_t et_shift, bool active, const uint64_t (&w)[(TB + 7) / 8], uint32_t d, int lane,
                                               const Dest& dest, unsigned long long* err) {
  static_assert(FAN * R <= 1024 && (TB == 4 || TB == 8 || TB == 16 || TB == 32), "ring geometry");
  constexpr uint32_t LINE = 128u / TB;
  uint32_t my = 0;
  if (active) my = __hip_atomic_fetch_add(&F.pos[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  const uint32_t line = my / LINE, slot = my % LINE, rl = d * R + (line & (R - 1u)), want = line / R;
  char* const cell = F.ring + (size_t)rl * 128u + slot * TB;
  // where my line goes if I turn out to own it: by position ...
  const uint32_t t0 = line * LINE, k = t0 >> et_shift, in_ext = t0 & ((1u << et_shift) - 1u);
  uint64_t e = dest.extent(d, k);
  bool pending = active;
  uint64_t pend = __ballot(pending);
  uint32_t spins = 0;
  while (pend) {
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    const bool can = pending && __hip_atomic_load(&F.gen[rl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == want;
    if (can) {
      if constexpr (TB == 4) *reinterpret_cast<uint32_t*>(cell) = (uint32_t)w[0];
      else if constexpr (TB == 8) *reinterpret_cast<uint64_t*>(cell) = w[0];
      else {
        vh_u64x2 v; v.x = w[0]; v.y = w[1];
        reinterpret_cast<vh_u64x2*>(cell)[0] = v;
        if constexpr (TB == 32) { vh_u64x2 v1; v1.x = w[2]; v1.y = w[3]; reinterpret_cast<vh_u64x2*>(cell)[1] = v1; }
      }
    }
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    uint32_t c = 0;
    if (can) c = __hip_atomic_fetch_add(&F.done[rl], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const bool own = can && c == LINE - 1u;
    const uint64_t om = __ballot(own);
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    if (om) {
      const uint32_t no = (uint32_t)__popcll(om);
      // ... or, beyond the stream's positions, from the pool's shared region (owners only: one lookup per line, one global atomic per extent)
      if (own && e == ~0ull) e = vh_ring_ovf_extent<MISSING>(F, dest, d, k, 1u << et_shift);
      const bool room = e != ~0ull;
      const uint64_t gline = (e * stride + in_ext) / LINE;      // in 128-byte lines from the pool's start (extents start on lines)
      if (own) F.list[__builtin_amdgcn_mbcnt_hi((uint32_t)(om >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)om, 0u))] = (uint64_t)rl | (room ? gline << 10 : ~0ull << 10);
      __atomic_signal_fence(__ATOMIC_SEQ_CST);
      __builtin_amdgcn_wave_barrier();
      for (uint32_t i = (uint32_t)lane >> 3; i < no; i += 8u) {
        const uint64_t ent = F.list[i];
        const uint32_t piece = (uint32_t)lane & 7u;
        const vh_u64x2 v = reinterpret_cast<const vh_u64x2*>(F.ring)[((uint32_t)ent & 1023u) * 8u + piece];
        if ((ent >> 10) != (~0ull >> 10)) reinterpret_cast<vh_u64x2*>(pool)[(ent >> 10) * 8u + piece] = v;
      }
      __atomic_signal_fence(__ATOMIC_SEQ_CST);
      __builtin_amdgcn_wave_barrier();
      if (own) {
        __hip_atomic_store(&F.done[rl], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __hip_atomic_store(&F.gen[rl], want + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (!room) atomicOr(err, VH_ERR_PART_FULL);
      }
    }
    pending = pending && !can;
    pend = __ballot(pending);
    if (pend) {
      __builtin_amdgcn_s_sleep(1);
      // (a ring place frees itself as soon as the eight writers of the line before have written, and none of them waits for anything later: the
      // wait is a few rounds. A bound all the same, and a block-wide word that ends every other wait at once — see the header comment)
      if (++spins > (1u << 22) || __hip_atomic_load(F.dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) {
        if (pending) { atomicOr(err, VH_ERR_PART_FULL); __hip_atomic_store(F.dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        break;
      }
    }
  }
}
// (the hashed partitioning's form: U 16-byte units per tuple, extents of VJ_FAN_ET units)
template <int U, class Dest>
__device__ __forceinline__ void vh_ring_add(const VhRing& F, vh_u64x2* pool, uint32_t stride, bool active, const uint64_t (&w)[2 * U], uint32_t d, int lane,
                                            const Dest& dest, unsigned long long* err) {
  vh_ring_add_tb<16 * U, Dest>(F, reinterpret_cast<char*>(pool), stride, U == 1 ? 12u : 11u, active, w, d, lane, dest, err);
  static_assert(VJ_FAN_ET == 4096, "extent shifts above");
}
template <int BLOCK, int FAN = VH_RING_FAN, int R = VH_RING_LINES>
__device__ __forceinline__ void vh_ring_init(char* area, VhRing& F, int wave) {      // area: VH_RING_LDS_BYTES(FAN, R, BLOCK) of LDS
  F.pos = reinterpret_cast<uint32_t*>(area);
  F.done = F.pos + FAN;
  F.gen = F.done + FAN * R;
  F.dead = F.gen + FAN * R;                                                           // (+ 3 words of padding)
  F.ovf = reinterpret_cast<unsigned long long*>(area + (size_t)FAN * 4 * (1 + 2 * R) + 16);
  F.ring = reinterpret_cast<char*>(F.ovf + 2 * FAN);
  F.list = reinterpret_cast<uint64_t*>(F.ring + (size_t)FAN * R * 128 + (size_t)wave * VJ_FAN_LIST_BYTES);
  for (uint32_t i = threadIdx.x; i < (uint32_t)FAN * (1 + 2 * R) + 4u + 4u * (uint32_t)FAN; i += BLOCK) F.pos[i] = 0u;      // counters, `dead`, the overflow table
  __syncthreads();
}
// The block's end: every digit's last, partial line, and the fill and tag (digit) of every extent the block wrote to BY POSITION (an overflow
// extent got both when it was taken; a stream's last extent is corrected here). MISSING: the fill array holds what an extent LACKS (DENSE_PART's
// pools: extent_missing), not what it holds (the hashed partitioning's).
template <int TB, int BLOCK, class Dest, int FAN = VH_RING_FAN, int R = VH_RING_LINES, bool MISSING = false>
__device__ __forceinline__ void vh_ring_finish_tb(const VhRing& F, char* pool, uint32_t stride, uint32_t et_shift, uint16_t* fill, uint8_t* tag, const Dest& dest, unsigned long long* err,
                                                  uint32_t* count = nullptr) {      // count: per digit, the tuples of all blocks (phase 2's shares: vh_part_shares)
  constexpr uint32_t LINE = 128u / TB;
  const uint32_t ET = 1u << et_shift;
  __syncthreads();
  for (uint32_t d = threadIdx.x; d < (uint32_t)FAN; d += BLOCK) {
    const uint32_t n = F.pos[d];
    if (!n) continue;
    if (count) atomicAdd(&count[d * VH_PART_COUNT_WAYS + (blockIdx.x & (VH_PART_COUNT_WAYS - 1u))], n);      // (eight counters per digit: a thousand blocks end at about the same time, and atomics on ONE address are served one after the other)
    bool full = false;
    const uint32_t left = n % LINE, line = n / LINE;
    const uint32_t klast = (n - 1u) >> et_shift;
    // the stream's last extent: by position, or out of the block's table (a last extent no complete line ever reached is taken here)
    uint64_t elast = dest.extent(d, klast);
    if (elast == ~0ull) elast = vh_ring_ovf_extent<MISSING>(F, dest, d, klast, ET);
    if (left) {
      const uint32_t t0 = line * LINE;
      if (elast != ~0ull) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(F.ring + (size_t)(d * R + (line & (R - 1u))) * 128u);
        uint32_t* dst = reinterpret_cast<uint32_t*>(pool + (elast * stride + (t0 & (ET - 1u))) * TB);
        for (uint32_t i = 0; i < left * (TB / 4u); ++i) dst[i] = src[i];
      } else full = true;
    }
    for (uint32_t k = 0; k <= klast; ++k) {
      const uint64_t e = k == klast ? elast : dest.extent(d, k);
      if (e == ~0ull) { if (k == klast) full = true; continue; }      // (an overflow extent before the last: marked full when it was taken)
      const uint32_t in = n - (k << et_shift), held = in < ET ? in : ET;
      fill[e] = (uint16_t)(MISSING ? ET - held : held);
      tag[e] = (uint8_t)d;
    }
    if (full) atomicOr(err, VH_ERR_PART_FULL);
  }
}
template <int U, int BLOCK, class Dest>
__device__ __forceinline__ void vh_ring_finish(const VhRing& F, vh_u64x2* pool, uint32_t stride, uint16_t* fill, uint8_t* tag, const Dest& dest, unsigned long long* err) {
  vh_ring_finish_tb<16 * U, BLOCK, Dest>(F, reinterpret_cast<char*>(pool), stride, U == 1 ? 12u : 11u, fill, tag, dest, err);
}

// =====================================================================================
// Fast variant: every predicate column is 4 bytes wide (u32 / i32 / f32 — dict codes, uint
// dims, time) and at most VH_MAX_PRED distinct ones are referenced. Differences to the generic
// kernel above, all aimed at keeping more HBM bytes in flight per wave:
//   * all predicate columns of a step are loaded up front into registers (NP x 4 x 16 B per
//     lane in flight at once) instead of one column per filter leaf;
//   * the loads of the NEXT step are issued right after the current step's masks are computed,
//     so they are in flight during the compaction and the survivor gathers / atomics;
//   * a survivor's group and metric values are all gathered before the first dependent use.
// Same results, same table organisations, same C-ABI.
template <typename T>
__device__ __forceinline__ uint32_t vh_cmp16_bits(const uint32_t (&bits)[VH_LANE_ROWS], uint64_t litbits, int op) {
  T v[VH_LANE_ROWS];
#pragma unroll
  for (int i = 0; i < VH_LANE_ROWS; ++i) v[i] = __builtin_bit_cast(T, bits[i]);
  return vh_cmp16<T>(v, vh_lit<T>(litbits), op);
}

__device__ __forceinline__ uint32_t vh_leaf_bits(const VhPlanDev& P, const VhProgOp o, const uint32_t (&bits)[VH_LANE_ROWS]) {
  if (o.kind() == VH_F_REL) {
    switch (o.type()) {
      case VH_I32: return vh_cmp16_bits<int32_t>(bits, P.ilits[o.lit()], o.op());
      case VH_F32: return vh_cmp16_bits<float>(bits, P.ilits[o.lit()], o.op());
      default: return vh_cmp16_bits<uint32_t>(bits, P.ilits[o.lit()], o.op());
    }
  }
  uint32_t m = o.op() ? 0u : VH_ROWMASK;
  for (int i = 0; i < o.count(); ++i) {
    const uint64_t lit = P.ilits[o.lit() + i];
    uint32_t e;
    switch (o.type()) {
      case VH_I32: e = vh_cmp16_bits<int32_t>(bits, lit, o.op() ? VH_OP_EQ : VH_OP_NE); break;
      case VH_F32: e = vh_cmp16_bits<float>(bits, lit, o.op() ? VH_OP_EQ : VH_OP_NE); break;
      default: e = vh_cmp16_bits<uint32_t>(bits, lit, o.op() ? VH_OP_EQ : VH_OP_NE); break;
    }
    if (o.op()) m |= e; else m &= e;
  }
  return m;
}

#ifndef VH_PRELOAD_WAIT
#define VH_PRELOAD_WAIT 0      // measurement: 1 = wait for every load of vh_preload before issuing the next, 2 = for every column's four
#endif
// Where the packed values of narrow predicate copies are widened to one register each (C3, five processes each,
// profiles/r02/NOTES.md "Narrow copies"): 3 = per column, right behind that column's four loads (125 VGPRs, 4 waves per SIMD:
// 3.12-3.26 ms); 2 = behind every single load (105 VGPRs, but the twelve loads of a step queue behind each other: 3.49-3.61);
// 1 = after all loads of the step (130 VGPRs, 3 waves: 3.89-3.97); 0 = at the top of the step that evaluates them (141 VGPRs: 3.88-3.92).
#ifndef VH_WIDEN_EARLY
#define VH_WIDEN_EARLY 3
#endif
// NARROW = false: the caller's plans never carry narrow copies (the lanes kernels: the host gives them the 4-byte arenas back —
// the extra paths cost them registers they do not have, C2 0.36 -> 0.50 ms with spills)
template <int NP, bool NARROW = true>
__device__ __forceinline__ void vh_preload(const VhPlanDev& P, uint32_t seg, uint32_t row_l, uint32_t seg_rows,
                                           uint32_t (&v)[NP][VH_LANE_ROWS]) {
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const uint32_t* col = reinterpret_cast<const uint32_t*>(P.colbase[P.pred_slot[p]] + (uint64_t)seg * P.colstride[P.pred_slot[p]]);
    const bool used = p < P.npred;     // a filter-less query runs the NP = 1 instantiation and loads nothing here
#pragma unroll
    for (int k = 0; k < VH_SUBSTEPS; ++k) {
      const uint32_t r = row_l + k * 256u;
      if (used && r < seg_rows) {
        if (!NARROW || P.pred_width[p] == 4) vh_load4<uint32_t>(col + r, &v[p][k * 4]);
        else if (P.pred_width[p] == 2) {       // narrow copy: four 16-bit values in one 8-byte load. They stay packed in the first two
                                               // registers until vh_widen, at the top of the step that evaluates them: unpacking here would
                                               // make every load wait for its data and the twelve loads of a step queue behind each other
          const uint64_t w = __builtin_nontemporal_load(reinterpret_cast<const uint64_t*>(reinterpret_cast<const uint16_t*>(col) + r));
          if (VH_WIDEN_EARLY == 2) {           // measurement: unpack behind every single load (the loads of a step queue behind each other)
            v[p][k * 4] = (uint32_t)w & 0xFFFFu; v[p][k * 4 + 1] = (uint32_t)(w >> 16) & 0xFFFFu;
            v[p][k * 4 + 2] = (uint32_t)(w >> 32) & 0xFFFFu; v[p][k * 4 + 3] = (uint32_t)(w >> 48);
          } else { v[p][k * 4] = (uint32_t)w; v[p][k * 4 + 1] = (uint32_t)(w >> 32); }
        } else {                               // ... four 8-bit values in one 4-byte load
          const uint32_t w = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(col) + r));
          if (VH_WIDEN_EARLY == 2) { v[p][k * 4] = w & 0xFFu; v[p][k * 4 + 1] = (w >> 8) & 0xFFu; v[p][k * 4 + 2] = (w >> 16) & 0xFFu; v[p][k * 4 + 3] = w >> 24; }
          else v[p][k * 4] = w;
        }
      } else {
        v[p][k * 4] = v[p][k * 4 + 1] = v[p][k * 4 + 2] = v[p][k * 4 + 3] = 0u;
      }
      if (VH_PRELOAD_WAIT == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (VH_PRELOAD_WAIT == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (NARROW && VH_WIDEN_EARLY == 3 && used && P.pred_width[p] != 4) {       // this column's four loads are out: unpack them (the wave waits here, once per column)
#pragma unroll
      for (int k = 0; k < VH_SUBSTEPS; ++k) {
        const uint32_t a = v[p][k * 4], b = v[p][k * 4 + 1];
        if (P.pred_width[p] == 2) { v[p][k * 4] = a & 0xFFFFu; v[p][k * 4 + 1] = a >> 16; v[p][k * 4 + 2] = b & 0xFFFFu; v[p][k * 4 + 3] = b >> 16; }
        else { v[p][k * 4] = a & 0xFFu; v[p][k * 4 + 1] = (a >> 8) & 0xFFu; v[p][k * 4 + 2] = (a >> 16) & 0xFFu; v[p][k * 4 + 3] = a >> 24; }
      }
    }
  }
}

// Values of narrow predicate copies, as vh_preload left them (packed), to one 32-bit register each.
template <int NP>
__device__ __forceinline__ void vh_widen(const VhPlanDev& P, uint32_t (&v)[NP][VH_LANE_ROWS]) {
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    if (P.pred_width[p] == 2) {
#pragma unroll
      for (int k = 0; k < VH_SUBSTEPS; ++k) {
        const uint32_t a = v[p][k * 4], b = v[p][k * 4 + 1];
        v[p][k * 4] = a & 0xFFFFu; v[p][k * 4 + 1] = a >> 16; v[p][k * 4 + 2] = b & 0xFFFFu; v[p][k * 4 + 3] = b >> 16;
      }
    } else if (P.pred_width[p] == 1) {
#pragma unroll
      for (int k = 0; k < VH_SUBSTEPS; ++k) {
        const uint32_t a = v[p][k * 4];
        v[p][k * 4] = a & 0xFFu; v[p][k * 4 + 1] = (a >> 8) & 0xFFu; v[p][k * 4 + 2] = (a >> 16) & 0xFFu; v[p][k * 4 + 3] = a >> 24;
      }
    }
  }
}

template <int NP>
__device__ __forceinline__ uint32_t vh_eval_filter_fast(const VhPlanDev& P, const uint32_t (&v)[NP][VH_LANE_ROWS], uint32_t row_l,
                                                        uint32_t seg_rows) {
  uint32_t m;
  if (P.prog_flat) {      // leaves under one AND (or one OR): fold as they come — a register array indexed by a run-time stack pointer
                          // costs a chain of selects per push and pop
    const bool conj = P.prog_flat == 1;
    m = conj ? VH_ROWMASK : 0u;
    const int nleaf = P.nprog > 1 ? P.nprog - 1 : 1;
    for (int pc = 0; pc < nleaf; ++pc) {
      const VhProgOp o = P.iprog[pc];
      uint32_t e = VH_ROWMASK;
      if (o.kind() != VH_F_TRUE) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
          if (o.pslot() == p) e = vh_leaf_bits(P, o, v[p]);
      }
      m = conj ? (m & e) : (m | e);
    }
  } else {
    uint32_t st[VH_MAX_STACK];
    int sp = 0;
    for (int pc = 0; pc < P.nprog; ++pc) {
      const VhProgOp o = P.iprog[pc];
      switch (o.kind()) {
        case VH_F_TRUE: st[sp++] = VH_ROWMASK; break;
        case VH_F_AND: {
          uint32_t a = st[--sp];
          for (int i = 1; i < o.count(); ++i) a &= st[--sp];
          st[sp++] = a;
        } break;
        case VH_F_OR: {
          uint32_t a = st[--sp];
          for (int i = 1; i < o.count(); ++i) a |= st[--sp];
          st[sp++] = a;
        } break;
        default: {
          uint32_t e = 0;
#pragma unroll
          for (int p = 0; p < NP; ++p)
            if (o.pslot() == p) e = vh_leaf_bits(P, o, v[p]);
          st[sp++] = e;
        } break;
      }
    }
    m = st[0];
  }
  if (row_l + (VH_SUBSTEPS - 1) * 256u + 4u > seg_rows) {
#pragma unroll
    for (int k = 0; k < VH_SUBSTEPS; ++k) {
      const uint32_t r = row_l + k * 256u;
      const uint32_t n = r >= seg_rows ? 0u : (seg_rows - r >= 4u ? 4u : seg_rows - r);
      m &= ~(((0xFu << n) & 0xFu) << (k * 4));
    }
  }
  return m;
}

// SHAPE 1 / 2 (DENSE_PART only; the host checks the plan, query_launch_locked): GROUP BY one or two unsigned columns of up to 32 bits
// (one: the host points the second digit at the first column with stride 0)
// (dictionary codes, booleans, uint dimensions) without time arithmetic, SUM of a 64-bit column (metric 0) + SUM of a 32-bit one (metric 1) — "GROUP BY two dimensions, SUM + COUNT", the
// reference's bread and butter. The generic drain walks the plan's column descriptors per survivor (element types, rollup rules,
// 64-bit digit arithmetic, tuple word / shift of every metric): ~300 VALU instructions per drain on a SIMD that is busy issuing
// 40 % of the time (profiles/r02/NOTES.md, "Instruction counts"); this one knows the answers.
template <int MODE, int SCOPE, int SHAPE = 0>
__device__ __forceinline__ void vh_consume_fast(const VhPlanDev& P, uint32_t seg, uint32_t row, bool active, char* lds,
                                                uint64_t xoff, unsigned long long& nfresh, VhPartWave& W, VhPartTile& T, VhLdsHashWave& H) {
  if (!active) row = 0;
  if (SHAPE != 0 && MODE == VH_MODE_DENSE_PART) {
    constexpr int J64 = SHAPE == 2 ? 1 : 0, J32 = 1 - J64;     // SHAPE 2: the same with the metrics in the other order
    uint32_t s0, s1, s2, s3;
    const uint64_t r0 = vh_gather_raw(P, P.g[0].slot(), seg, row, s0);
    const uint64_t r1 = vh_gather_raw(P, P.g[1].slot(), seg, row, s1);
    const uint64_t r2 = vh_gather_raw(P, P.m[J64].slot(), seg, row, s2);
    const uint64_t r3 = vh_gather_raw(P, P.m[J32].slot(), seg, row, s3);
    if (VH_ABLATE & 2) {       // measurement build: the gathers and nothing behind them ((VH_ABLATE & 1): not even those)
      const uint64_t acc = (VH_ABLATE & 1) ? (uint64_t)row : r0 + r1 + r2 + r3;
      if (acc == 0x123456789ABCDEFull) P.counters[7] = acc;
      return;
    }
    // unsigned group columns of 1, 2 or 4 bytes (dictionary codes, booleans, uint dimensions): the host left the element's width
    // as a right shift in key_shift (unused on the dense paths): 0, 16 or 24
    const uint32_t d0 = (((uint32_t)(r0 >> s0) << P.g[0].key_shift()) >> P.g[0].key_shift()) - (uint32_t)P.g[0].lo;
    const uint32_t d1 = (((uint32_t)(r1 >> s1) << P.g[1].key_shift()) >> P.g[1].key_shift()) - (uint32_t)P.g[1].lo;
    const bool bad = d0 >= (uint32_t)P.g[0].extent || d1 >= (uint32_t)P.g[1].extent;
    if (__ballot(active && bad)) {
      if (active && bad) atomicOr(P.counters + 2, VH_ERR_RANGE);
    }
    active = active && !bad;
    const uint32_t gid = d0 * (uint32_t)P.g[0].stride + d1 * (uint32_t)P.g[1].stride;
    const uint64_t words[2] = {(uint64_t)gid | ((r3 >> s3) << 32), r2 >> s2};
    vh_part_direct_add<2>(P, T, W, active, words, gid >> P.part_shift, (int)(threadIdx.x & 63));
    return;
  }
  // All of a survivor's values are requested before the first one is looked at: the aligned 8-byte word around each
  // element is loaded whatever the column's type (no type switch, hence no branch and no wait, between the loads), and
  // only then shifted / masked / sign-extended. One memory round trip per drain instead of one per column — the drain
  // is a dependent chain (queue -> gather -> table), so its latency is what a wave's survivors cost.
  uint64_t gv[VH_FAST_COLS], mv[VH_FAST_COLS];
  uint32_t gsh[VH_FAST_COLS], msh[VH_FAST_COLS];
#pragma unroll
  for (int i = 0; i < VH_FAST_COLS; ++i) {
    gv[i] = 0; gsh[i] = 0;
    if (i < P.ngroup) { if (VH_ABLATE & 1) gv[i] = P.g[i].lo + (row & 63u); else gv[i] = vh_gather_raw(P, P.g[i].slot(), seg, row, gsh[i]); }
  }
#pragma unroll
  for (int j = 0; j < VH_FAST_COLS; ++j) {
    mv[j] = 0; msh[j] = 0;
    if (j < P.nmetric && P.m[j].slot() != VH_SLOT_ROWID) { if (VH_ABLATE & 1) mv[j] = row; else mv[j] = vh_gather_raw(P, P.m[j].slot(), seg, row, msh[j]); }
  }
#pragma unroll
  for (int i = 0; i < VH_FAST_COLS; ++i)
    if (i < P.ngroup) gv[i] = vh_decode_bits(gv[i] >> gsh[i], P.g[i].type(), MODE != VH_MODE_HASH);
#pragma unroll
  for (int j = 0; j < VH_FAST_COLS; ++j) {
    if (j < P.nmetric) {
      const VhMetricDev& m = P.m[j];
      mv[j] = m.slot() == VH_SLOT_ROWID ? (((uint64_t)seg << 32) | row) : vh_decode_bits(mv[j] >> msh[j], m.type(), vh_sop_sext(m.sop()));
    }
  }
  if (VH_ABLATE & 2) {     // keep the loads alive, drop everything behind them
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < VH_FAST_COLS; ++i) acc += gv[i] + mv[i];
    if (acc == 0x123456789ABCDEFull) P.counters[7] = acc;
    return;
  }
  uint64_t gid = 0;
  uint64_t key[VH_KEY_WORDS];
  if (MODE == VH_MODE_HASH) {
#pragma unroll
    for (int i = 0; i < VH_KEY_WORDS; ++i) key[i] = 0;
  }
  bool bad = false;
#pragma unroll
  for (int i = 0; i < VH_FAST_COLS; ++i) {
    if (i < P.ngroup) {
      const VhGroupDev& g = P.g[i];
      uint64_t v = gv[i];
      if (g.gran() != VH_T_NONE || g.nroll()) v = vh_time_rollup(v, g);
      if (MODE == VH_MODE_HASH) {
        if (g.type() == VH_F32 && (uint32_t)v == 0x80000000u) v = 0;
        if (g.type() == VH_F64 && v == 0x8000000000000000ull) v = 0;
        key[g.key_word()] |= v << g.key_shift();
      } else {
        const uint64_t d = v - g.lo;
        bad |= d >= g.extent;
        gid += d * g.stride;
      }
    }
  }
  if (MODE == VH_MODE_HASH) {
    if (P.lds_hash_slots && !H.bypass) {
      uint32_t ls = 0;
      const bool in_lds = active && key[0] != VH_HASH_EMPTY && vh_lds_hash_find(P, lds, key[0], ls);
      const uint32_t nh = __popcll(__ballot(in_lds)), nm = __popcll(__ballot(active && !in_lds));
      H.hits += nh; H.misses += nm;
      if (H.hits + H.misses >= 4096u && H.misses > H.hits) H.bypass = true;
      if (in_lds) {
#pragma unroll
        for (int j = 0; j < VH_FAST_COLS; ++j)
          if (j < P.nmetric) vh_state_update<__HIP_MEMORY_SCOPE_WORKGROUP>(lds + P.m[j].lds_off, ls, P.m[j].sop(), mv[j]);
      }
      active = active && !in_lds;
    }
    bool ok = true, fresh = false;
    if (active) {
      gid = P.key_words == 1 ? vh_hash_insert64(P, key[0], ok, fresh) : vh_hash_insert_wide(P, key, P.key_words, ok, fresh);
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
  if (MODE == VH_MODE_DENSE_PART) {
    uint64_t words[1 + VH_FAST_COLS];
    words[0] = gid & 0xFFFFFFFFull;
#pragma unroll
    for (int w = 1; w < 1 + VH_FAST_COLS; ++w) words[w] = 0;
#pragma unroll
    for (int j = 0; j < VH_FAST_COLS; ++j) {
      if (j < P.nmetric) {
        const VhMetricDev& m = P.m[j];
        const uint64_t v = (vh_sop_bytes(m.sop()) == 4 ? (mv[j] & 0xFFFFFFFFull) : mv[j]) << m.tshift();
#pragma unroll
        for (int w = 0; w < 1 + VH_FAST_COLS; ++w)
          if (m.tword() == w) words[w] |= v;
      }
    }
    vh_part_direct_add<1 + VH_FAST_COLS>(P, T, W, active, words, (uint32_t)(gid >> P.part_shift), (int)(threadIdx.x & 63));
    return;
  }
  if (MODE == VH_MODE_DENSE_LDS) {
    if (active) reinterpret_cast<uint8_t*>(lds + P.lds_present_off)[gid] = 1;
  } else if (MODE == VH_MODE_DENSE_GLOBAL) {
    if (active && P.present_carrier < 0) P.present[xoff + gid] = 1;
  }
  if (MODE == VH_MODE_HASH) {      // a wave's hot group: one lane speaks for its lanes (vh_hot_lanes)
    const uint64_t hot = vh_hot_lanes(active, gid);
    if (hot) {       // (wave-uniform)
      const int lane_ = (int)(threadIdx.x & 63);
      const bool in_hot = ((hot >> lane_) & 1ull) != 0, speaks = lane_ == __builtin_ctzll(hot);
#pragma unroll
      for (int j = 0; j < VH_FAST_COLS; ++j) {
        if (j < P.nmetric) {
          const VhMetricDev& m = P.m[j];
          const uint64_t tot = vh_wave_combine(m.sop(), mv[j], in_hot);
          if (in_hot ? speaks : active) vh_state_update<__HIP_MEMORY_SCOPE_AGENT>(vh_hash_state(P, m, gid), 0, m.sop(), in_hot ? tot : mv[j]);
        }
      }
      return;
    }
  }
#pragma unroll
  for (int j = 0; j < VH_FAST_COLS; ++j) {
    if (j < P.nmetric && active) {
      const VhMetricDev& m = P.m[j];
      if (MODE == VH_MODE_DENSE_LDS) vh_state_update<__HIP_MEMORY_SCOPE_WORKGROUP>(lds + m.lds_off, gid, m.sop(), mv[j]);
      else if (MODE == VH_MODE_DENSE_GLOBAL) vh_state_update<SCOPE>(m.state, xoff + gid, m.sop(), mv[j]);
      else vh_state_update<__HIP_MEMORY_SCOPE_AGENT>(vh_hash_state(P, m, gid), 0, m.sop(), mv[j]);
    }
  }
}

#ifndef VH_FAST_WAVES
#define VH_FAST_WAVES(MODE, BLOCK, NP) 0      // waves per SIMD asked of the compiler (0: no request); experiments override it
#endif
template <int MODE, int BLOCK, int SCOPE, int NP, int SHAPE>
__device__ __forceinline__ void vh_scan_fast_body(const VhPlanDev& P) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  typedef VhScanCfg<BLOCK> C;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  uint32_t* q = reinterpret_cast<uint32_t*>(lds + ((MODE == VH_MODE_DENSE_LDS || MODE == VH_MODE_HASH) ? P.lds_bytes : 0)) + wave * C::kQueueCap;
  VhLdsHashWave H{0u, 0u, false, 0ull};
  if (MODE == VH_MODE_HASH && P.lds_hash_slots) vh_lds_hash_init(P, lds, BLOCK);
  VhPartWave W;
  VhPartTile T;
  if (MODE == VH_MODE_DENSE_PART)
    vh_part_tile_init(P, lds, T, W);

  if (MODE == VH_MODE_DENSE_LDS) {
    for (int j = 0; j < P.nmetric; ++j) {
      const VhMetricDev& m = P.m[j];
      const uint64_t ident = m.ident;
      if (vh_sop_bytes(m.sop()) == 4) {
        for (uint64_t g = threadIdx.x; g < P.G; g += BLOCK) reinterpret_cast<uint32_t*>(lds + m.lds_off)[g] = (uint32_t)ident;
      } else {
        for (uint64_t g = threadIdx.x; g < P.G; g += BLOCK) reinterpret_cast<uint64_t*>(lds + m.lds_off)[g] = ident;
      }
    }
    for (uint64_t g = threadIdx.x; g < P.G; g += BLOCK) reinterpret_cast<uint8_t*>(lds + P.lds_present_off)[g] = 0;
    __syncthreads();
  }

  const uint64_t xoff = (MODE == VH_MODE_DENSE_GLOBAL && P.nxcd > 1) ? (uint64_t)(vh_xcc_id() % P.nxcd) * P.xcd_stride : 0;
  uint32_t npassed32 = 0;                    // per lane: a lane sees a 64th of this block's rows (< 2^32); one register less than a 64-bit count
  unsigned long long nfresh = 0;
  const uint32_t spu = P.unit_rows / C::kStepRows;  // steps per unit

  // this block's units are blockIdx.x, + gridDim.x, ...; a unit is `spu` steps inside one segment. (segment, unit inside the
  // segment, step inside the unit) advance by additions: three integer divisions per step were ~75 instructions of every step.
  const uint32_t gdiv = gridDim.x / P.units_per_seg, gmod = gridDim.x % P.units_per_seg;
  uint32_t unit = blockIdx.x, seg = 0, useg = 0, ustep = 0, wave_base = 0, seg_rows = 0;    // useg: unit index inside its segment
  bool have = unit < P.total_units;
  if (have) {
    seg = unit / P.units_per_seg;
    useg = unit - seg * P.units_per_seg;
    seg_rows = P.seg_rows[seg];
    wave_base = useg * P.unit_rows + wave * VH_WAVE_STEP_ROWS;
  }
  uint32_t v[NP][VH_LANE_ROWS];
  if (have) { vh_preload<NP>(P, seg, wave_base + lane * 4, seg_rows, v); if (VH_WIDEN_EARLY == 1) vh_widen<NP>(P, v); }
  uint32_t cnt = 0;
  while (have) {
    const uint32_t row_l = wave_base + lane * 4;
    if (!VH_WIDEN_EARLY) vh_widen<NP>(P, v);
    const uint32_t mask = vh_eval_filter_fast<NP>(P, v, row_l, seg_rows);
    npassed32 += __popc(mask);
    // locate the next step and put its predicate columns in flight now
    uint32_t nseg = seg, nwave_base = wave_base + C::kStepRows, nseg_rows = seg_rows;
    bool nhave = true;
    if (++ustep == spu) {
      ustep = 0;
      unit += gridDim.x;
      nhave = unit < P.total_units;
      useg += gmod; nseg = seg + gdiv;
      if (useg >= P.units_per_seg) { useg -= P.units_per_seg; ++nseg; }
      if (nhave) {
        nseg_rows = P.seg_rows[nseg];
        nwave_base = useg * P.unit_rows + wave * VH_WAVE_STEP_ROWS;
      }
    }
    if (nhave) { vh_preload<NP>(P, nseg, nwave_base + lane * 4, nseg_rows, v); if (VH_WIDEN_EARLY == 1) vh_widen<NP>(P, v); }
    // One textual drain for the four sub-steps and for the flush at a segment's end (k == VH_SUBSTEPS): with the loop unrolled the
    // drain — gathers, rollup, table update, every table organisation's alternatives — was inlined five times and the kernel ran to
    // 26 K instructions, several times the instruction cache (profiles/r02/NOTES.md, "Code size").
#pragma unroll 1
    for (int k = 0; k <= VH_SUBSTEPS; ++k) {
      bool flush = false;
      if (k < VH_SUBSTEPS) {
        const uint32_t mk = (mask >> (4 * k)) & 0xFu;
        if (__ballot(mk != 0) == 0) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool b = (mk >> j) & 1u;
          const uint64_t bal = __ballot(b);
          if (b) q[cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] = row_l + k * 256u + j;
          cnt += __popcll(bal);
        }
        __builtin_amdgcn_wave_barrier();
      } else {
        flush = !nhave || nseg != seg;       // queue entries are rows of the current segment
      }
      // Tried and measured slower (profiles/r01/NOTES.md): draining two survivors per lane with both gathers
      // in flight (+8..50 %: register pressure), and a "dense lane" path with coalesced 4-row payload loads
      // for wave steps where most rows pass (179-242 VGPRs, scratch spills in the LDS variant).
      while (cnt >= 64 || (flush && cnt)) {
        const uint32_t take = cnt >= 64 ? 64u : cnt;
        cnt -= take;
        const bool act = (uint32_t)lane < take;
        const uint32_t r = act ? q[cnt + lane] : 0u;
        vh_consume_fast<MODE, SCOPE, SHAPE>(P, seg, r, act, lds, xoff, nfresh, W, T, H);
        __builtin_amdgcn_wave_barrier();
      }
    }
    have = nhave; seg = nseg; wave_base = nwave_base; seg_rows = nseg_rows;
    if (MODE == VH_MODE_HASH && H.dead) have = false;     // this wave saw the table overflow: the attempt is void (see scan_agg_kernel)
  }

  if (MODE == VH_MODE_DENSE_PART) vh_part_tile_finish(P, T, lane);
  unsigned long long npassed = npassed32;
  for (int off = 32; off > 0; off >>= 1) npassed += __shfl_down(npassed, off);
  vh_scan_block_end(P, npassed, nfresh, 0ull, MODE == VH_MODE_DENSE_PART ? vh_part_wave_end(P, W) : 0u);
  if (MODE == VH_MODE_HASH && P.lds_hash_slots) vh_lds_hash_flush(P, lds, BLOCK);
  if (MODE == VH_MODE_DENSE_LDS) {
    __syncthreads();
    const uint64_t xo = P.nxcd > 1 ? (uint64_t)(vh_xcc_id() % P.nxcd) * P.xcd_stride : 0;
    for (uint64_t g = threadIdx.x; g < P.G; g += BLOCK) {
      if (!reinterpret_cast<uint8_t*>(lds + P.lds_present_off)[g]) continue;
      P.present[xo + g] = 1;
      for (int j = 0; j < P.nmetric; ++j) {
        const VhMetricDev& m = P.m[j];
        const uint64_t bits = vh_sop_bytes(m.sop()) == 4 ? reinterpret_cast<uint32_t*>(lds + m.lds_off)[g]
                                                       : reinterpret_cast<uint64_t*>(lds + m.lds_off)[g];
        vh_state_update<SCOPE>(m.state, xo + g, m.sop(), bits);
      }
    }
  }
}

template <int MODE, int BLOCK, int SCOPE, int NP>
__global__ __launch_bounds__(BLOCK, VH_FAST_WAVES(MODE, BLOCK, NP)) void scan_agg_fast_kernel(const VhPlanDev P) {
  vh_scan_fast_body<MODE, BLOCK, SCOPE, NP, 0>(P);
}
// the same scan with a drain specialised for one plan shape (vh_consume_fast, SHAPE)
// (at most 4 waves per SIMD asked for: left alone, the compiler aims at 5-6 for the one- and two-column instantiations and
// pays for it with scratch spills inside the scan loop)
#ifndef VH_SHAPE_WAVES
#define VH_SHAPE_WAVES 4
#endif
#ifndef VH_SHAPE_MINWAVES
#define VH_SHAPE_MINWAVES 1
#endif
template <int MODE, int BLOCK, int SCOPE, int NP, int SHAPE>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(VH_SHAPE_MINWAVES, VH_SHAPE_WAVES))) void scan_agg_shape_kernel(const VhPlanDev P) {
  vh_scan_fast_body<MODE, BLOCK, SCOPE, NP, SHAPE>(P);
}

// ------------------------------------------------- "lanes" variant: no compaction (high selectivity, LDS table)
// When a large fraction of the rows pass (C2: 50 %), compacting survivors into a queue and gathering their payload
// one scalar load per survivor and column costs more than it saves: every payload line is touched anyway. This
// variant keeps a lane on its own 4 consecutive rows of each sub-step, loads the group / metric columns for them
// with the same coalesced 16 B/lane vector loads as the predicate columns, and lets the lanes whose mask bit is set
// update the LDS table directly. Restricted (host side) to DENSE_LDS plans with <= VH_LANES_COLS group and metric
// columns of 4 or 8 bytes and no time truncation, chosen when the selectivity probe says >= 25 % of the rows pass.
#define VH_LANES_COLS 2
#ifndef VH_LANES_WAVES
#define VH_LANES_WAVES(MODE, BLOCK, NP) ((MODE) == VH_MODE_DENSE_LDS && (BLOCK) == 256 && (NP) == 1 ? 6 : 0)
#endif
__device__ __forceinline__ void vh_load_rows4(const char* base, int type, uint32_t r0, bool sext, uint64_t (&out)[4]) {
  if (type == VH_U64 || type == VH_I64 || type == VH_F64) {
    vh_load4<uint64_t>(reinterpret_cast<const uint64_t*>(base) + r0, out);
  } else {
    uint32_t t[4];
    vh_load4<uint32_t>(reinterpret_cast<const uint32_t*>(base) + r0, t);
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = (type == VH_I32 && sext) ? (uint64_t)(int64_t)(int32_t)t[j] : (uint64_t)t[j];
  }
}

template <int MODE, int BLOCK, int SCOPE, int NP>
__global__ __launch_bounds__(BLOCK, VH_LANES_WAVES(MODE, BLOCK, NP)) void scan_agg_lanes_kernel(const VhPlanDev P) {
  // MODE = VH_MODE_DENSE_LDS: direct-indexed LDS table. MODE = VH_MODE_HASH: the LDS front table (time buckets,
  // float keys, ...), rows that find no slot there go to the HBM table — same rules as the compacting kernel.
  extern __shared__ __attribute__((aligned(16))) char lds[];
  typedef VhScanCfg<BLOCK> C;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  VhPartWave W;
  VhPartTile T;
  if (MODE == VH_MODE_DENSE_PART) {       // phase 1 of the radix-partitioned aggregation: one LDS tile per wave
    vh_part_tile_init(P, lds + (size_t)wave * vh_part_tile_bytes(P), T, W);
  } else if (MODE == VH_MODE_HASH) {
    vh_lds_hash_init(P, lds, BLOCK);
  } else {
    for (int j = 0; j < P.nmetric; ++j) {
      const VhMetricDev& m = P.m[j];
      const uint64_t ident = m.ident;
      if (vh_sop_bytes(m.sop()) == 4) {
        for (uint64_t g = threadIdx.x; g < P.G; g += BLOCK) reinterpret_cast<uint32_t*>(lds + m.lds_off)[g] = (uint32_t)ident;
      } else {
        for (uint64_t g = threadIdx.x; g < P.G; g += BLOCK) reinterpret_cast<uint64_t*>(lds + m.lds_off)[g] = ident;
      }
    }
    for (uint64_t g = threadIdx.x; g < P.G; g += BLOCK) reinterpret_cast<uint8_t*>(lds + P.lds_present_off)[g] = 0;
    __syncthreads();
  }

  unsigned long long npassed = 0, nfresh = 0;
  uint32_t lane_hits = 0, lane_misses = 0;     // HASH: this lane's luck with the LDS front table
  const uint32_t spu = P.unit_rows / C::kStepRows;
  uint32_t t = 0, seg = 0, unit_base = 0, wave_base = 0, seg_rows = 0;
  bool have;
  {
    const uint32_t unit = blockIdx.x;
    have = unit < P.total_units;
    if (have) {
      seg = unit / P.units_per_seg;
      unit_base = (unit - seg * P.units_per_seg) * P.unit_rows;
      seg_rows = P.seg_rows[seg];
      wave_base = unit_base + wave * VH_WAVE_STEP_ROWS;
    }
  }
  uint32_t v[NP][VH_LANE_ROWS];
  if (have) vh_preload<NP, false>(P, seg, wave_base + lane * 4, seg_rows, v);
  bool range_err = false, full_err = false;
  while (have) {
    const uint32_t row_l = wave_base + lane * 4;
    const uint32_t mask = vh_eval_filter_fast<NP>(P, v, row_l, seg_rows);
    npassed += __popc(mask);
    ++t;
    uint32_t nseg = seg, nunit_base = unit_base, nwave_base = 0, nseg_rows = seg_rows;
    bool nhave;
    {
      const uint32_t unit = blockIdx.x + (t / spu) * gridDim.x;
      nhave = unit < P.total_units;
      if (nhave) {
        nseg = unit / P.units_per_seg;
        nunit_base = (unit - nseg * P.units_per_seg) * P.unit_rows;
        nseg_rows = P.seg_rows[nseg];
        nwave_base = nunit_base + (t % spu) * C::kStepRows + wave * VH_WAVE_STEP_ROWS;
      }
    }
    if (nhave) vh_preload<NP, false>(P, nseg, nwave_base + lane * 4, nseg_rows, v);
#pragma unroll
    for (int k = 0; k < VH_SUBSTEPS; ++k) {
      const uint32_t mk = (mask >> (4 * k)) & 0xFu;
      const uint32_t r0 = row_l + k * 256u;
      uint64_t gv[VH_LANES_COLS][4], mv[VH_LANES_COLS][4];
#pragma unroll
      for (int i = 0; i < VH_LANES_COLS; ++i) {
        gv[i][0] = gv[i][1] = gv[i][2] = gv[i][3] = 0;
        if (i < P.ngroup && mk) {     // mk == 0 also covers rows at or beyond size(): nothing is loaded out of bounds
          const VhGroupDev& g = P.g[i];
          vh_load_rows4(P.colbase[g.slot()] + (uint64_t)seg * P.colstride[g.slot()], g.type(), r0, MODE != VH_MODE_HASH, gv[i]);
        }
      }
#pragma unroll
      for (int j = 0; j < VH_LANES_COLS; ++j) {
        mv[j][0] = mv[j][1] = mv[j][2] = mv[j][3] = 0;
        if (j < P.nmetric && mk) {
          const VhMetricDev& m = P.m[j];
          vh_load_rows4(P.colbase[m.slot()] + (uint64_t)seg * P.colstride[m.slot()], m.type(), r0, vh_sop_sext(m.sop()), mv[j]);
        }
      }
      if (MODE == VH_MODE_DENSE_PART) {
        // Phase 1 of the radix-partitioned aggregation, one wave tile = this sub-step's 256 row slots (vh_part_tile_write).
        if (__ballot(mk != 0) == 0) continue;
        uint64_t words[4][1 + VH_LANES_COLS];
        uint32_t part[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          uint64_t gid = 0;
          bool bad = false;
#pragma unroll
          for (int i = 0; i < VH_LANES_COLS; ++i) {
            if (i < P.ngroup) {
              const VhGroupDev& g = P.g[i];
              const uint64_t d = gv[i][r] - g.lo;
              bad |= d >= g.extent;
              gid += d * g.stride;
            }
          }
          const bool act = ((mk >> r) & 1u) && !bad;
          if (((mk >> r) & 1u) && bad) range_err = true;
          words[r][0] = gid & 0xFFFFFFFFull;
#pragma unroll
          for (int w = 1; w < 1 + VH_LANES_COLS; ++w) words[r][w] = 0;
#pragma unroll
          for (int j = 0; j < VH_LANES_COLS; ++j) {
            if (j < P.nmetric) {
              const VhMetricDev& m = P.m[j];
              const uint64_t x = (vh_sop_bytes(m.sop()) == 4 ? (mv[j][r] & 0xFFFFFFFFull) : mv[j][r]) << m.tshift();
#pragma unroll
              for (int w = 0; w < 1 + VH_LANES_COLS; ++w)
                if (m.tword() == (uint32_t)w) words[r][w] |= x;
            }
          }
          part[r] = act ? (uint32_t)(gid >> P.part_shift) : 0xFFFFFFFFu;
        }
        vh_part_tile_write<1 + VH_LANES_COLS>(P, T, W, words, part, lane);
        continue;
      }
      if (VH_ABLATE & 64) {      // measurement build: the loads of a sub-step, nothing behind them
        uint64_t acc = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc += gv[0][r] + gv[1][r] + mv[0][r] + mv[1][r];
        if (acc == 0x123456789ABCDEFull) P.counters[7] = acc;
        continue;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (!((mk >> r) & 1u)) continue;
        if (MODE == VH_MODE_HASH) {
          uint64_t key = 0;
#pragma unroll
          for (int i = 0; i < VH_LANES_COLS; ++i) {
            if (i < P.ngroup) {
              const VhGroupDev& g = P.g[i];
              uint64_t x = gv[i][r];
              if (g.gran() != VH_T_NONE || g.nroll()) x = vh_time_rollup(x, g);
              if (g.type() == VH_F32 && (uint32_t)x == 0x80000000u) x = 0;            // -0.0f == 0.0f
              if (g.type() == VH_F64 && x == 0x8000000000000000ull) x = 0;
              key |= x << g.key_shift();
            }
          }
          uint32_t ls = 0;
          const bool bypass = lane_hits + lane_misses >= 64u && lane_misses > lane_hits;
          if (!bypass && key != VH_HASH_EMPTY && vh_lds_hash_find(P, lds, key, ls)) {
            ++lane_hits;
#pragma unroll
            for (int j = 0; j < VH_LANES_COLS; ++j)
              if (j < P.nmetric) vh_state_update<__HIP_MEMORY_SCOPE_WORKGROUP>(lds + P.m[j].lds_off, ls, P.m[j].sop(), mv[j][r]);
          } else {
            ++lane_misses;
            bool ok = true, fresh = false;
            const uint64_t gid = vh_hash_insert64(P, key, ok, fresh);
            if (!ok) { full_err = true; continue; }
            nfresh += fresh ? 1 : 0;
#pragma unroll
            for (int j = 0; j < VH_LANES_COLS; ++j)
              if (j < P.nmetric) vh_state_update<__HIP_MEMORY_SCOPE_AGENT>(vh_hash_state(P, P.m[j], gid), 0, P.m[j].sop(), mv[j][r]);
          }
        } else {
          uint64_t gid = 0;
          bool bad = false;
#pragma unroll
          for (int i = 0; i < VH_LANES_COLS; ++i) {
            if (i < P.ngroup) {
              const VhGroupDev& g = P.g[i];
              const uint64_t d = gv[i][r] - g.lo;
              bad |= d >= g.extent;
              gid += d * g.stride;
            }
          }
          if (bad) { range_err = true; continue; }
          reinterpret_cast<uint8_t*>(lds + P.lds_present_off)[gid] = 1;
#pragma unroll
          for (int j = 0; j < VH_LANES_COLS; ++j)
            if (j < P.nmetric) vh_state_update<__HIP_MEMORY_SCOPE_WORKGROUP>(lds + P.m[j].lds_off, gid, P.m[j].sop(), mv[j][r]);
        }
      }
    }
    have = nhave; seg = nseg; unit_base = nunit_base; wave_base = nwave_base; seg_rows = nseg_rows;
  }
  if (__ballot(range_err)) { if (range_err) atomicOr(P.counters + 2, VH_ERR_RANGE); }
  if (__ballot(full_err)) { if (full_err) atomicOr(P.counters + 2, VH_ERR_HASH_FULL); }
  for (int off = 32; off > 0; off >>= 1) { npassed += __shfl_down(npassed, off); nfresh += __shfl_down(nfresh, off); }
  vh_scan_block_end(P, npassed, nfresh, 0ull, MODE == VH_MODE_DENSE_PART ? vh_part_wave_end(P, W) : 0u);
  if (MODE == VH_MODE_DENSE_PART) { vh_part_tile_finish(P, T, lane); return; }
  if (MODE == VH_MODE_HASH) { vh_lds_hash_flush(P, lds, BLOCK); return; }
  __syncthreads();
  const uint64_t xo = P.nxcd > 1 ? (uint64_t)(vh_xcc_id() % P.nxcd) * P.xcd_stride : 0;
  for (uint64_t g = threadIdx.x; g < P.G; g += BLOCK) {
    if (!reinterpret_cast<uint8_t*>(lds + P.lds_present_off)[g]) continue;
    P.present[xo + g] = 1;
    for (int j = 0; j < P.nmetric; ++j) {
      const VhMetricDev& m = P.m[j];
      const uint64_t bits = vh_sop_bytes(m.sop()) == 4 ? reinterpret_cast<uint32_t*>(lds + m.lds_off)[g]
                                                       : reinterpret_cast<uint64_t*>(lds + m.lds_off)[g];
      vh_state_update<SCOPE>(m.state, xo + g, m.sop(), bits);
    }
  }
}

// Extent tags a wave looks at per step: a full ballot when there are plenty of extents, fewer when `waves` waves would
// otherwise not all find work (the extents of a partition are spread evenly over the tag array).
__device__ __forceinline__ uint32_t vh_tag_group(uint32_t extents, uint32_t waves) {
  const uint32_t g = extents / (waves ? waves : 1u);
  return g >= 64u ? 64u : (g ? g : 1u);
}

// ------------------------------------------------- partitioned aggregation, phase 2
// grid = nfine x blocks_per_part. A block owns an LDS table for its range's 2^agg_shift groups,
// its waves walk the range's extents (64 tuples each, one coalesced 16 B/lane load for 2-word
// tuples), every tuple is an LDS-atomic update, and the block finally merges its table into the dense
// global table (one update per present group and block; plain stores when the block is the range's only one).
// One level: range f = partition f of pool 1. Two levels: range f = sub-partition f & 63 of partition f >> 6, whose
// extents are the ones tagged (f & 63) inside that partition's slice of pool 2.
#ifndef VH_P2_SLOTS
#define VH_P2_SLOTS 4      // 64-tuple slots a wave of phase 2 has in flight
#endif
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void part_agg_kernel(const VhPlanDev P, int blocks_per_part) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  int part, b;
  const bool balanced = P.part_count != nullptr && P.nlevel == 1;
  const uint32_t my_count = balanced && threadIdx.x < 64 ? vh_part_count_of(P.part_count, P.npart, (int)threadIdx.x) : 0u;
  if (!vh_part_my_share(P, blocks_per_part, my_count, part, b, blocks_per_part)) return;      // (blocks_per_part: from here on THIS partition's blocks)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = BLOCK / 64;
  const uint64_t gpp = 1ull << P.agg_shift;
  const uint64_t g0 = (uint64_t)part << P.agg_shift;
  const uint64_t ng = g0 >= P.G ? 0 : (P.G - g0 < gpp ? P.G - g0 : gpp);
  for (int j = 0; j < P.nmetric; ++j) {
    const VhMetricDev& m = P.m[j];
    if (vh_sop_bytes(m.sop()) == 4) {
      for (uint64_t g = threadIdx.x; g < ng; g += BLOCK) reinterpret_cast<uint32_t*>(lds + m.lds_off)[g] = (uint32_t)m.ident;
    } else {
      for (uint64_t g = threadIdx.x; g < ng; g += BLOCK) reinterpret_cast<uint64_t*>(lds + m.lds_off)[g] = m.ident;
    }
  }
  const bool carried = P.present_carrier >= 0;        // a SOP_ADD32P state says whether the group exists: no presence bytes in LDS
  if (!carried) for (uint64_t g = threadIdx.x; g < ng; g += BLOCK) reinterpret_cast<uint8_t*>(lds + P.lds_present_off)[g] = 0;
  __syncthreads();
  const bool two = P.nlevel == 2;
  uint32_t first = 0, total;
  if (two) {
    const uint32_t lo = P.l2[part >> 6], hi = P.l2[(part >> 6) + 1], used = P.l2[VH_L2_NEXT + (part >> 6)];
    first = lo;
    total = lo + (used < hi - lo ? used : hi - lo);
  } else {
    const unsigned long long allocated = P.counters[5];
    total = allocated < P.max_extents ? (uint32_t)allocated : P.max_extents;   // extents handed out (opened or only reserved)
  }
  const uint8_t want = (uint8_t)(two ? (part & 63) : part);
  const uint8_t* tags = two ? P.extent_part2 : P.extent_part;
  const uint16_t* missing = two ? P.extent_missing2 : P.extent_missing;
  const uint64_t* pool = two ? P.tuples2 : P.tuples;
  const uint32_t ext_tuples = (uint32_t)(two ? P.ext_tuples2 : P.ext_tuples), ext_stride = (uint32_t)(two ? P.ext_tuples2 : P.ext_stride);
  const uint32_t tw = (uint32_t)P.tw;
  // the waves of this range's blocks share the tag array `gsz` extents at a time (64, or fewer when there are not enough extents
  // to go round: phase 1 writes few, large extents when few rows survive); a tag that equals `want` is an extent to aggregate
  // SUM of a 64-bit column + SUM of a 32-bit one (SUM + COUNT: the reference's bread and butter) in two-word tuples:
  // word 0 = gid | the 32-bit value << 32, word 1 = the 64-bit value
  int j64 = -1, j32 = -1;
  if (P.tw == 2 && P.nmetric == 2)
    for (int j = 0; j < 2; ++j) {
      if (P.m[j].sop() == SOP_ADD64 && P.m[j].tword() == 1 && P.m[j].tshift() == 0) j64 = j;
      if ((P.m[j].sop() == SOP_ADD32 || P.m[j].sop() == SOP_ADD32P) && P.m[j].tword() == 0 && P.m[j].tshift() == 32) j32 = j;
    }
  const bool sum_pair = j64 >= 0 && j32 >= 0 && (P.m[j32 < 0 ? 0 : j32].sop() == SOP_ADD32P) == carried;
  // one-word tuples (VhPlanDev::gid_bits): the gid's mask, and the SUM(64-bit) + SUM(32-bit, presence carrier) pair's shifts and masks
  const uint64_t gid_mask = P.gid_bits ? (1ull << P.gid_bits) - 1ull : 0xFFFFFFFFull;
  int k64 = -1, k32 = -1;
  if (P.gid_bits && P.tw == 1 && P.nmetric == 2)
    for (int j = 0; j < 2; ++j) { if (P.m[j].sop() == SOP_ADD64) k64 = j; if (P.m[j].sop() == SOP_ADD32P) k32 = j; }
  const bool pk_pair = k64 >= 0 && k32 >= 0 && carried;
  const uint32_t pk_s64 = P.m[k64 < 0 ? 0 : k64].tshift(), pk_s32 = P.m[k32 < 0 ? 0 : k32].tshift();
  const uint64_t pk_m64 = (1ull << P.m[k64 < 0 ? 0 : k64].tbits) - 1ull, pk_m32 = (1ull << P.m[k32 < 0 ? 0 : k32].tbits) - 1ull;
  if (pk_pair) { /* sum64 / sum32 below point at the pair's LDS arrays */ }
  unsigned long long* const sum64 = reinterpret_cast<unsigned long long*>(lds + P.m[pk_pair ? k64 : j64 < 0 ? 0 : j64].lds_off);
  char* const sum32 = lds + P.m[pk_pair ? k32 : j32 < 0 ? 0 : j32].lds_off;
  const uint32_t gsz = vh_tag_group(total - first, (uint32_t)blocks_per_part * nwaves);
  // Per step a wave looks at gsz tags AND the fill of those extents (one vector load each, side by side: a dependent scalar load
  // per extent was a serial memory round trip — phase 1 leaves ~80 K extents of ~600 tuples on C3), then walks the extents that
  // are its partition's as ONE stream of 64-tuple slots, VH_P2_SLOTS slots in flight whatever extent they come from.
  for (uint32_t c0 = first + ((uint32_t)b * nwaves + wave) * gsz; c0 < total; c0 += (uint32_t)blocks_per_part * nwaves * gsz) {
   const bool in = (uint32_t)lane < gsz && c0 + lane < total;
   const uint8_t tag = in ? tags[c0 + lane] : (uint8_t)0xFF;
   const uint32_t fill = in ? ext_tuples - missing[c0 + lane] : 0u;
   uint64_t mine = __ballot(in && tag == want && fill != 0);
   uint32_t ext = 0, valid = 0, at = 0;            // the extent being walked (wave-uniform)
   while (mine || at < valid) {
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
        sbase[u] = pool + ((uint64_t)ext * ext_stride + at) * tw;
        sn[u] = valid - at < 64u ? valid - at : 64u;
        at += 64u;
      } else { sbase[u] = pool; sn[u] = 0; }
    }
    uint64_t w[VH_P2_SLOTS][1 + VH_FAST_COLS];
#pragma unroll
    for (int u = 0; u < VH_P2_SLOTS; ++u) {
#pragma unroll
      for (int x = 0; x < 1 + VH_FAST_COLS; ++x)
        w[u][x] = ((uint32_t)lane < sn[u] && (uint32_t)x < tw) ? ((VH_ABLATE & 32) ? (x == 0 ? g0 + ((at * 2654435761u + lane * 40503u + u * 977u) & (gpp - 1)) : 1ull)   // measurement build: no tuple loads
                                                                                      : __builtin_nontemporal_load(sbase[u] + (uint64_t)lane * tw + x)) : (x == 0 ? ~0ull : 0ull);
    }
    if (VH_ABLATE & 16) {        // measurement build: tuple loads only
      uint64_t acc = 0;
#pragma unroll
      for (int u = 0; u < VH_P2_SLOTS; ++u) acc += w[u][0] + w[u][1];
      if (acc == 0x123456789ABCDEFull) P.counters[7] = acc;
      continue;
    }
    if (pk_pair) {        // the same shape in ONE-word tuples: gid | the 64-bit SUM's value | the 32-bit one's, each at the width the planner gave it
#pragma unroll
      for (int u = 0; u < VH_P2_SLOTS; ++u) {
        const uint64_t w0 = w[u][0], local = (w0 & gid_mask) - g0;
        if (w0 == ~0ull || local >= ng) continue;
        __hip_atomic_fetch_add(sum64 + local, (unsigned long long)((w0 >> pk_s64) & pk_m64), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(sum32) + local, (1ull << 32) | ((w0 >> pk_s32) & pk_m32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      continue;
    }
    if (sum_pair) {       // the common shape, without the per-tuple walk over the plan's metric descriptors (550 -> ~60 instructions per 256 tuples)
#pragma unroll
      for (int u = 0; u < VH_P2_SLOTS; ++u) {
        const uint64_t w0 = w[u][0], local = (w0 & 0xFFFFFFFFull) - g0;
        if (w0 == ~0ull || local >= ng) continue;
        __hip_atomic_fetch_add(sum64 + local, (unsigned long long)w[u][1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (carried) __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(sum32) + local, (1ull << 32) | (w0 >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else {
          reinterpret_cast<uint8_t*>(lds + P.lds_present_off)[local] = 1;
          __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(sum32) + local, (uint32_t)(w0 >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      continue;
    }
#pragma unroll
    for (int u = 0; u < VH_P2_SLOTS; ++u) {
      const uint64_t local = (w[u][0] & gid_mask) - g0;
      if (w[u][0] == ~0ull || local >= ng) continue;  // an empty slot; (a corrupt tuple cannot write outside the table)
      if (!carried) reinterpret_cast<uint8_t*>(lds + P.lds_present_off)[local] = 1;
#pragma unroll
      for (int j = 0; j < VH_FAST_COLS; ++j) {
        if (j < P.nmetric) {
          const VhMetricDev& m = P.m[j];
          uint64_t v = 0;
#pragma unroll
          for (int x = 0; x < 1 + VH_FAST_COLS; ++x)
            if (m.tword() == (uint32_t)x) v = w[u][x] >> m.tshift();
          if (P.gid_bits && m.tbits) v &= (1ull << m.tbits) - 1ull;      // (one-word tuples: never negative, the planner checked the column's minimum)
          else if (vh_sop_bytes(m.sop()) == 4) {
            v &= 0xFFFFFFFFull;
            if (vh_sop_sext(m.sop())) v = (uint64_t)(int64_t)(int32_t)v;
          }
          vh_state_update<__HIP_MEMORY_SCOPE_WORKGROUP>(lds + m.lds_off, local, m.sop(), v);
        }
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
    const uint8_t here = carried ? (uint8_t)(reinterpret_cast<uint64_t*>(lds + P.m[P.present_carrier].lds_off)[g] != 0)
                                 : reinterpret_cast<uint8_t*>(lds + P.lds_present_off)[g];
    if (own && (balanced || blocks_per_part > 1)) { if (!carried) P.present[xo + g0 + g] = here; }
    else { if (!here) continue; if (!carried) P.present[g0 + g] = 1; }
    for (int j = 0; j < P.nmetric; ++j) {
      const VhMetricDev& m = P.m[j];
      if (vh_sop_bytes(m.sop()) == 4) {
        const uint32_t bits = reinterpret_cast<uint32_t*>(lds + m.lds_off)[g];
        if (own) reinterpret_cast<uint32_t*>(m.state)[xo + g0 + g] = bits;
        else vh_state_update<__HIP_MEMORY_SCOPE_AGENT>(m.state, g0 + g, m.sop(), bits);
      } else {
        const uint64_t bits = reinterpret_cast<uint64_t*>(lds + m.lds_off)[g];
        if (own) reinterpret_cast<uint64_t*>(m.state)[xo + g0 + g] = bits;
        else vh_state_update<__HIP_MEMORY_SCOPE_AGENT>(m.state, g0 + g, m.sop(), bits);
      }
    }
  }
}

// ------------------------------------------------- two-level partitioning: sizing and the second split
// After phase 1: how many tuples did each partition get? One block counts them off the extent tags and lays the
// partitions' slices of pool 2 out back to back: tuples / extent size, plus what the splitting waves can leave open
// (every wave of `waves_per_part` may hold one partly filled extent per sub-partition and an unused rest of a chunk).
// Two launches: every CU counts its share of the extent tags into l2[VH_L2_NEXT + p] (zero when the query starts), one block then
// turns the counts into slices and resets the cursors. (One block doing both took 0.39 ms for the 100 K extents of a 60 M-tuple pool.)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void part_l2_count_kernel(const VhPlanDev P) {
  __shared__ unsigned int cnt[VH_MAX_PART];
  if (threadIdx.x < VH_MAX_PART) cnt[threadIdx.x] = 0;
  __syncthreads();
  const VhPools Q = vh_pools(P);
  const uint32_t total = Q.allocated1 < Q.max1 ? (uint32_t)Q.allocated1 : Q.max1;
  for (uint32_t e = blockIdx.x * BLOCK + threadIdx.x; e < total; e += gridDim.x * BLOCK) {
    const uint8_t p = Q.tag1[e];
    if (p != 0xFF) atomicAdd(&cnt[p], (uint32_t)P.ext_tuples - Q.miss1[e]);
  }
  __syncthreads();
  if (threadIdx.x < VH_MAX_PART && cnt[threadIdx.x]) atomicAdd(Q.l2 + VH_L2_WORDS + threadIdx.x, cnt[threadIdx.x]);     // (scratch words behind the table proper)
}
// ring_blocks != 0 (part_split_ring_kernel writes the slices): every (block, sub-partition) stream of a partition gets vh_slice_levels extents by
// POSITION — its share of the partition's counted tuples, and one more —, and behind them lies the slice's shared overflow region: room for ALL the
// partition's tuples once more, so that any skew inside the partition fits (VhRingOvf; the slice's cursor starts behind the positional extents and
// phase 2 looks at every extent it says is used).
__host__ __device__ __forceinline__ uint32_t vh_slice_levels(unsigned long long count, unsigned long long streams, uint32_t et, uint32_t cap = ~0u) {
  const uint32_t k = (uint32_t)(count / streams / et) + 1u;
  return k < cap ? k : cap;
}
__host__ __device__ __forceinline__ unsigned long long vh_slice_extents(unsigned long long count, unsigned long long streams, uint32_t et, uint32_t cap = ~0u) {
  return count ? (unsigned long long)vh_slice_levels(count, streams, et, cap) * streams + count / et + streams + 2ull : 0ull;      // (+ streams: every stream's last, part-filled overflow extent)
}
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void part_l2_plan_kernel(const VhPlanDev P, int waves_per_part, int ring_blocks) {
  const VhPools Q = vh_pools(P);
  if (threadIdx.x == 0) {
    unsigned long long at = 0;
    for (int p = 0; p < P.npart; ++p) {
      const unsigned long long c = Q.l2[VH_L2_WORDS + p];
      Q.l2[p] = (uint32_t)(at < Q.max2 ? at : Q.max2);
      Q.l2[VH_L2_NEXT + p] = 0;
      if (ring_blocks) {
        const unsigned long long per = 64ull * (unsigned)ring_blocks, need = vh_slice_extents(c, per, (uint32_t)P.ext_tuples2, P.slice_levels_cap);
        Q.l2[VH_L2_NEXT + p] = c ? vh_slice_levels(c, per, (uint32_t)P.ext_tuples2, P.slice_levels_cap) * (uint32_t)per : 0u;      // (used so far: the positional extents; overflow extents are counted on top as they are taken)
        at += need;
        continue;
      }
      // (+ 1/4: an extent is closed as soon as a drain's tuples of its sub-partition do not fit, so skewed data leaves up to 63 of 256 slots unused)
      if (c) at += (c + c / 4 + (uint32_t)P.ext_tuples2 - 1) / (uint32_t)P.ext_tuples2 + (unsigned long long)waves_per_part * (64 + VH_EXT_CHUNK);
    }
    Q.l2[P.npart] = (uint32_t)(at < Q.max2 ? at : Q.max2);
    if (at > Q.max2) atomicOr(P.counters + 2, VH_ERR_PART_FULL);     // the host re-runs with a larger second pool
  }
}

// grid = npart x blocks_per_part. The block's waves walk partition p's extents of pool 1 (like phase 2 does) and append
// every tuple to sub-partition (gid >> agg_shift) & 63 of p's slice of pool 2, with the same ballot-rank append phase 1 uses.
// Tuple by tuple like the compacting form of phase 1; collecting 256-tuple tiles in LDS and writing runs (the lanes form's
// writer) was measured slower here — 64 sub-partitions leave runs of 4 tuples, one store instruction each (31.1 vs 29.8 ms for
// 1 G tuples end to end, 39 ms with fewer waves) — and so were twice the waves (more extents open at once).
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void part_split_kernel(const VhPlanDev P, int blocks_per_part) {
  const int part = blockIdx.x / blocks_per_part, b = blockIdx.x % blocks_per_part;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = BLOCK / 64;
  VhPartWave W;
  VhPartTile T;
  vh_part_tile_init(P, nullptr, T, W);
  W.base = P.l2[part]; W.limit = P.l2[part + 1]; W.cursor = P.l2 + VH_L2_NEXT + part;
  const unsigned long long allocated = P.counters[5];
  const uint32_t total = allocated < P.max_extents ? (uint32_t)allocated : P.max_extents;
  const uint32_t ext_tuples = (uint32_t)P.ext_tuples, tw = (uint32_t)P.tw;
  const uint32_t gsz = vh_tag_group(total, (uint32_t)blocks_per_part * nwaves);
  for (uint32_t c0 = ((uint32_t)b * nwaves + wave) * gsz; c0 < total; c0 += (uint32_t)blocks_per_part * nwaves * gsz) {
    uint64_t mine = __ballot((uint32_t)lane < gsz && c0 + lane < total && P.extent_part[c0 + lane] == (uint8_t)part);
    while (mine) {
      const uint32_t ext = c0 + (uint32_t)__builtin_ctzll(mine);
      mine &= mine - 1;
      const uint32_t valid = ext_tuples - P.extent_missing[ext];
      const uint64_t* base = P.tuples + (uint64_t)ext * (uint32_t)P.ext_stride * tw;
      for (uint32_t i0 = 0; i0 < valid; i0 += 128) {
        uint64_t w[2][1 + VH_FAST_COLS];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const uint32_t i = i0 + u * 64 + lane;
#pragma unroll
          for (int x = 0; x < 1 + VH_FAST_COLS; ++x)
            w[u][x] = (i < valid && (uint32_t)x < tw) ? __builtin_nontemporal_load(base + (uint64_t)i * tw + x) : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (i0 + u * 64 >= valid) break;
          const bool active = i0 + u * 64 + lane < valid;
          const uint32_t sub = (uint32_t)((w[u][0] & 0xFFFFFFFFull) >> P.agg_shift) & 63u;
          vh_part_direct_add<1 + VH_FAST_COLS, 2>(P, T, W, active, w[u], sub, lane);
        }
      }
    }
  }
  vh_part_tile_finish<2>(P, T, lane);
}

#define VH_SPLIT_TILE_TUPLES 2048      // tuples per extent of pool 2 when one- and two-word tuples are split through the ring writer (a power of two)

// ------------------------------------------------- the second split without barriers
// (Rounds 3-5 split one- and two-word tuples a block-wide tile at a time: 2 048 tuples sorted in LDS between four block barriers, runs that start and
// end anywhere in a line; 5.20 ms for 1 B tuples where this kernel takes 3.58.) A block's waves walk partition p's extents of pool 1 each on its own and append every tuple to sub-partition (gid >> agg_shift) & 63 through the
// ring writer (vh_ring_add_tb: a tuple counter and two waiting 128-byte lines per sub-partition in LDS, whole lines out, extents by position in the
// slice part_l2_plan_kernel(ring_blocks) laid out). A (block, sub-partition) stream that outgrows its positions goes on in the slice's shared overflow region.
struct VhSplitDest {
  uint32_t lo, kmax, bpp, b;
  VhRingOvf ovf;
  __device__ __forceinline__ uint64_t extent(uint32_t d, uint32_t k) const { return k < kmax ? (uint64_t)lo + ((uint64_t)k * bpp + b) * 64u + d : ~0ull; }
};
#define VH_SPLIT_RING_LDS(block) VH_RING_LDS_BYTES(64, 2, block)
template <int TB> struct VhSplitTuple;      // the tuple as it lies in the pools: two words, one, or four bytes (VhPlanDev::tuple4)
template <> struct VhSplitTuple<16> { typedef vh_u64x2 type; };
template <> struct VhSplitTuple<8> { typedef uint64_t type; };
template <> struct VhSplitTuple<4> { typedef uint32_t type; };
template <int BLOCK, int TB>
__global__ __launch_bounds__(BLOCK) void part_split_ring_kernel(const VhPlanDev P, int blocks_per_part) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  typedef typename VhSplitTuple<TB>::type Tup;
  constexpr int TW = TB == 16 ? 2 : 1, UNR = 4, NW = BLOCK / 64;
  const int part = blockIdx.x / blocks_per_part, b = blockIdx.x % blocks_per_part;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  VhRing F;
  vh_ring_init<BLOCK, 64, 2>(lds, F, wave);
  const VhPools Q = vh_pools(P);
  const uint32_t lo = Q.l2[part], cap = Q.l2[part + 1] - lo;
  // (the slice: positional extents for the counted tuples' even spread, then its shared overflow region — part_l2_plan_kernel laid it out from the same count)
  const uint32_t kpos = vh_slice_levels(Q.l2[VH_L2_WORDS + part], 64ull * (unsigned)blocks_per_part, (uint32_t)P.ext_tuples2, P.slice_levels_cap);
  const VhSplitDest D{lo, kpos * 64u * (uint32_t)blocks_per_part <= cap ? kpos : cap / (64u * (uint32_t)blocks_per_part), (uint32_t)blocks_per_part, (uint32_t)b,
                      VhRingOvf{lo, cap, Q.l2 + VH_L2_NEXT + part, nullptr, Q.miss2, Q.tag2}};
  const uint32_t et2 = (uint32_t)P.ext_tuples2, et2_shift = 31u - (uint32_t)__builtin_clz(et2);      // (a power of two: VH_SPLIT_TILE_TUPLES)
  const uint64_t gid_mask = TW == 1 ? (1ull << P.gid_bits) - 1ull : ~0ull;
  const int gshift = P.gid_shift;
  const uint32_t total = Q.allocated1 < Q.max1 ? (uint32_t)Q.allocated1 : Q.max1;
  const uint32_t ext_tuples = (uint32_t)P.ext_tuples, ext_stride1 = (uint32_t)P.ext_stride;
  char* const pool2 = reinterpret_cast<char*>(Q.t2);
  const uint32_t gsz = vh_tag_group(total, (uint32_t)blocks_per_part * NW);
  for (uint32_t c0 = ((uint32_t)b * NW + wave) * gsz; c0 < total; c0 += (uint32_t)blocks_per_part * NW * gsz) {
    const bool in = (uint32_t)lane < gsz && c0 + lane < total;
    const uint32_t fill = in && Q.tag1[c0 + lane] == (uint8_t)part ? ext_tuples - Q.miss1[c0 + lane] : 0u;
    uint64_t mine = __ballot(fill != 0);
    while (mine) {
      const int q = __builtin_ctzll(mine);
      mine &= mine - 1;
      const uint32_t ext = c0 + (uint32_t)q, valid = (uint32_t)__builtin_amdgcn_readlane((int)fill, q);
      const Tup* base = reinterpret_cast<const Tup*>(Q.t1) + (uint64_t)ext * ext_stride1;
      if constexpr (TB == 4) {
        // four-byte tuples: one 16-byte load per lane, four tuples each (a 4-byte load per lane reads a pool at 1.1 TB/s: what phase 2 found out
        // the same way). Extents start on 128-byte lines and hold a multiple of 32 tuples: the loads are aligned and stay inside the extent.
        for (uint32_t i0 = 0; i0 < valid; i0 += 256u * 2u) {
          vh_u32x4 v[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) if (i0 + u * 256u + (uint32_t)lane * 4u < valid) v[u] = __builtin_nontemporal_load(reinterpret_cast<const vh_u32x4*>(base + i0 + u * 256u) + lane);
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (i0 + u * 256u >= valid) break;                                   // (wave-uniform)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const bool ok = i0 + u * 256u + (uint32_t)lane * 4u + (uint32_t)k < valid;
              uint64_t w[1] = {k == 0 ? v[u].x : k == 1 ? v[u].y : k == 2 ? v[u].z : v[u].w};
              const uint32_t sub = ok ? ((uint32_t)((w[0] & gid_mask) >> gshift) >> P.agg_shift) & 63u : 0u;
              vh_ring_add_tb<TB, VhSplitDest, 64, 2, true>(F, pool2, et2, et2_shift, ok, w, sub, lane, D, P.counters + 2);
            }
          }
        }
        continue;
      }
      for (uint32_t i0 = 0; i0 < valid; i0 += 64u * UNR) {
        Tup t[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) if (i0 + u * 64u + lane < valid) t[u] = __builtin_nontemporal_load(base + i0 + u * 64u + lane);
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          if (i0 + u * 64u >= valid) break;                                     // (wave-uniform)
          const bool ok = i0 + u * 64u + lane < valid;
          uint64_t w[TW];
          if constexpr (TW == 1) w[0] = t[u]; else { w[0] = t[u].x; w[1] = t[u].y; }
          // (four-byte tuples carry the gid relative to the partition: its bits from agg_shift up are the sub-partition all the same)
          const uint32_t sub = ok ? ((uint32_t)((w[0] & gid_mask) >> gshift) >> P.agg_shift) & 63u : 0u;
          vh_ring_add_tb<TB, VhSplitDest, 64, 2, true>(F, pool2, et2, et2_shift, ok, w, sub, lane, D, P.counters + 2);
        }
      }
    }
  }
  vh_ring_finish_tb<TB, BLOCK, VhSplitDest, 64, 2, true>(F, pool2, et2, et2_shift, Q.miss2, Q.tag2, D, P.counters + 2);
}




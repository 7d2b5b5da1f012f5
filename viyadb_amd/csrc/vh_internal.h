// vh_internal.h — device-visible plan layout and small helpers shared by the
// kernels and the C-ABI host code. Not part of the public boundary.
#pragma once
// (hipRTC — the per-query kernels of vh_jit.hip — pre-includes the HIP runtime and is handed viya_hip.h by name)
#ifdef __HIPCC_RTC__
#include <stdint.h>
#include "viya_hip.h"
#else
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/viya_hip.h"
#endif

#define VH_MAX_LITS 65535  // literal pool (VhProgOp::lit is 16 bits)
#define VH_INLINE_PROG 24  // filter programs up to this size (and VH_INLINE_LITS literals) also travel in the kernel arguments, where the
#define VH_INLINE_LITS 32  // register-resident kernels read them: loads off the kernarg segment are hoisted out of the scan loop, loads through a pointer are not
#define VH_MAX_GROUP 16   // group-by columns
#define VH_MAX_METRIC 20  // selected metrics (+ hidden count)
#define VH_MAX_SLOTS 32   // distinct columns a query may reference (a payload projection adds one slot per gathered column)
#define VH_MAX_STACK 8    // predicate mask stack depth
#define VH_MAX_HAVING 16       // postfix nodes of a pushed-down HAVING
#define VH_MAX_HAVING_LITS 24
#define VH_MAX_XCD 8
#define VH_KEY_WORDS 8    // widest group key: 8 x u64
#define VH_BS_PAD 16      // bytes kept readable behind a segment's bitset ids: the compiled kernels fetch a row's first two ids with ONE 8-byte load
#define VH_MAX_BITSET 2   // bitset (count-distinct) metrics per query
#define VH_MAX_PRED 4     // fast path: distinct 4-byte predicate columns held in registers
#ifndef VH_FAST_COLS
#define VH_FAST_COLS 4    // fast path: group / metric columns gathered up front
#endif
#define VH_MAX_PART 64     // DENSE_PART: partitions (lane p of a wave keeps partition p's state, so at most one per lane)
#define VH_EXT_CHUNK 8    // DENSE_PART: extents a wave reserves per global allocation
#define VH_L2_NEXT 80      // VhPlanDev::l2: offset of the per-partition allocation cursors
#define VH_L2_WORDS 160

// Row geometry of one block step (see DESIGN.md "scan geometry"):
// a wave covers 1024 consecutive rows per step as 4 sub-steps of 256 rows;
// lane l owns rows {k*256 + l*4 + j | k,j in 0..3} -> 16 rows, 16 mask bits.
#define VH_WAVE 64
#define VH_ROWS_PER_LANE_VEC 4
#ifndef VH_SUBSTEPS
#define VH_SUBSTEPS 4                              // sub-steps of 256 rows a wave handles per step (4 rows per lane each)
#endif
#define VH_LANE_ROWS (4 * VH_SUBSTEPS)             // rows a lane owns per wave step
#define VH_WAVE_STEP_ROWS (256 * VH_SUBSTEPS)
#define VH_ROWMASK ((1u << VH_LANE_ROWS) - 1u)     // pass-mask bits of one lane
// per-query compiled kernels (vh_jit_body.h): a wave's survivor queue = carry-over (< 64) + every row slot of one wave step
// (a step's slots are compacted in one go, then drained)
#define VJ_QUEUE_CAP (64 + VH_WAVE_STEP_ROWS)
// ... and, where the scan of a hashed-partitioning query writes the level-A pool itself (vj_fan_add), the block's writer behind the queues:
// [pos | done | gen] counters, VJ_FAN_RING waiting 128-byte lines per digit, a list of finished lines per wave
#define VJ_FAN 256               // = HP_FAN (vh_hpart.h; the generated unit asserts it)
#define VJ_FAN_ET 4096           // = HP_ET
#define VJ_FAN_RING 2
#define VJ_FAN_LIST_BYTES 512    // per wave: 64 x (ring line | destination line << 10)
// LDS of a block's ring writer (vh_ring_add_tb, vh_kernels.h): [pos | done | gen] counters, the block's `dead` word (+ padding to 16 bytes), two
// words per digit for the overflow extents of the moment, `lines` waiting 128-byte lines per digit, a list of finished lines per wave
#define VH_RING_LDS_BYTES(fan, lines, block) ((size_t)(fan) * 4 * (1 + 2 * (lines)) + 16 + (size_t)(fan) * 16 + (size_t)(fan) * (lines) * 128 + (size_t)((block) / 64) * VJ_FAN_LIST_BYTES)
#define VJ_FAN_LDS_BYTES(block) VH_RING_LDS_BYTES(VJ_FAN, VJ_FAN_RING, block)

// table organisations of the scan kernels
enum { VH_MODE_DENSE_LDS = 1, VH_MODE_DENSE_GLOBAL = 2, VH_MODE_HASH = 3, VH_MODE_DENSE_PART = 4 };

// state-update opcodes (what one surviving row does to one metric state)
enum vh_state_op : uint8_t {
  SOP_ADD32 = 0,  // integer += in a <=32-bit type (wraps mod 2^w after truncation)
  SOP_ADD64,      // integer += in a 64-bit type
  SOP_ADDF32, SOP_ADDF64,
  SOP_MIN_I32, SOP_MAX_I32, SOP_MIN_U32, SOP_MAX_U32,
  SOP_MIN_I64, SOP_MAX_I64, SOP_MIN_U64, SOP_MAX_U64,
  SOP_MIN_F32, SOP_MAX_F32, SOP_MIN_F64, SOP_MAX_F64,
  SOP_BITSET,     // emit (group, value) pairs; handled out of line
  SOP_ADD32P      // 32-bit integer += widened to a 64-bit word that also counts rows in its upper half:
                  // state += (1 << 32) | v. Low half = the wrapped 32-bit sum, word != 0 <=> group exists,
                  // so the dense HBM table needs no separate presence store (one write transaction less per row)
};

// Every member of the structs below is a 32- or 64-bit word; the byte-sized attributes are packed into words
// and read through accessors. Reason (profiles/r01/NOTES.md, tests/test_isa_hazards.py): with byte-sized MEMBERS in a
// kernel-argument struct that is indexed dynamically, hipcc (ROCm 7.2) forms scalar loads off the address of a
// byte member (`s_load_dwordx2 sN, s[&m.sop], 0x15`); SMEM drops the low two bits of the base and the load returns
// the wrong bytes. Without sub-dword members no such base exists.
#define VH_PACKED_FIELD(name, word, shift, bits)                                                           \
  __host__ __device__ __forceinline__ uint32_t name() const { return (word >> shift) & ((1u << bits) - 1u); } \
  __host__ __device__ __forceinline__ void set_##name(uint32_t v) {                                          \
    word = (word & ~(((1u << bits) - 1u) << shift)) | ((v & ((1u << bits) - 1u)) << shift);                  \
  }

struct VhProgOp {      // 8 bytes
  uint32_t w0, w1;
  VH_PACKED_FIELD(kind, w0, 0, 8)     // vh_fkind
  VH_PACKED_FIELD(type, w0, 8, 8)     // vh_elem of the column
  VH_PACKED_FIELD(op, w0, 16, 8)      // vh_relop / IN polarity
  VH_PACKED_FIELD(count, w0, 24, 8)   // IN: literals; AND/OR: operands
  VH_PACKED_FIELD(slot, w1, 0, 8)     // referenced-column slot
  VH_PACKED_FIELD(pslot, w1, 8, 8)    // fast path: index among the distinct predicate columns
  VH_PACKED_FIELD(lit, w1, 16, 16)    // first literal
};

struct VhGroupDev {
  uint32_t w0, w1;
  uint64_t roll_units;                 // 8 x 8 bits: vh_time_unit of rollup rule k
  VH_PACKED_FIELD(slot, w0, 0, 16)
  VH_PACKED_FIELD(type, w0, 16, 8)     // vh_elem
  VH_PACKED_FIELD(gran, w0, 24, 8)     // vh_time_unit or VH_T_NONE
  VH_PACKED_FIELD(nroll, w1, 0, 8)
  VH_PACKED_FIELD(micro, w1, 8, 8)
  VH_PACKED_FIELD(key_word, w1, 16, 8)   // hash path: which u64 word of the key holds this column
  VH_PACKED_FIELD(key_shift, w1, 24, 8)  // hash path: bit offset inside that word
  __host__ __device__ __forceinline__ uint32_t roll_unit(int k) const { return (uint32_t)(roll_units >> (8 * k)) & 0xFFu; }
  __host__ __device__ __forceinline__ void set_roll_unit(int k, uint32_t v) {
    roll_units = (roll_units & ~(0xFFull << (8 * k))) | ((uint64_t)(v & 0xFFu) << (8 * k));
  }
  uint64_t roll_before[VH_MAX_ROLLUP];
  uint64_t lo;         // dense path: value - lo is the digit
  uint64_t extent;     // dense path: digit < extent
  uint64_t stride;     // dense path: gid += digit * stride
};

struct VhMetricDev {
  uint32_t w0, w1;
  VH_PACKED_FIELD(slot, w0, 0, 16)
  VH_PACKED_FIELD(type, w0, 16, 8)     // vh_elem of the source column
  VH_PACKED_FIELD(sop, w0, 24, 8)      // vh_state_op
  VH_PACKED_FIELD(tword, w1, 0, 8)     // DENSE_PART: tuple word that carries this metric's value
  VH_PACKED_FIELD(tshift, w1, 8, 8)    // DENSE_PART: bit offset inside that word (0 or 32)
  uint32_t lds_off;    // DENSE_LDS / DENSE_PART phase 2 / hash front table: byte offset of this metric's state array in LDS
  uint32_t tbits;      // hashed partitioning, packed tuples: bits the value takes in the tuple's second word (0: its state's full width)
  void* state;         // global state array (G or capacity elements, 4 or 8 B each)
  uint64_t ident;      // identity bits: 0 (SUM/AVG/COUNT), type max (MIN), cpp_min_value (MAX)
};

struct VhPlanDev {
  // ---- filter: postfix program + literal pool, uploaded next to the segment snapshot (uniform addresses: scalar loads, like
  // kernel arguments, but without their 4 KB ceiling: an IN list may hold thousands of values)
  int32_t nprog;
  int32_t prog_flat;          // 1: the program is leaves + one AND over all of them (or a single leaf), 2: ... one OR; 0: anything else (stack machine)
  int32_t nslots;
  const VhProgOp* prog;       // generic kernels (scan_agg_kernel, select_*): any length
  const uint64_t* lits;
  VhProgOp iprog[VH_INLINE_PROG];   // register-resident kernels (scan_agg_fast_kernel, scan_agg_lanes_kernel): the same program when it fits
  uint64_t ilits[VH_INLINE_LITS];
  // ---- fast path (all predicate columns 4 bytes wide, <= VH_MAX_PRED of them): their slots
  int32_t npred;
  int32_t tuple4;             // DENSE_PART, one-word tuples whose gid and values fit 32 bits together: the tuple is FOUR bytes (32 per 128-byte line;
                              // C3: gid 17 + SUM value 10 + COUNT value 2 bits) — half the tuple bytes of phase 1 and phase 2 once more. With two levels the
                              // gid field is RELATIVE to the level-1 partition (gid_bits = part_shift: 4 M groups in 8 partitions = 19 bits, not 22)
  int32_t gid_bits;           // DENSE_PART with ONE-word tuples: word 0 = gid in its low gid_bits | every metric value at m[j].tshift, m[j].tbits wide (0: two or more words, the gid in word 0's low half)
  uint8_t pred_slot[8];
  uint8_t pred_width[8];      // bytes per element the kernel reads for predicate column p: 4, or 1 / 2 when the table keeps a narrow copy (vh_table_narrow)
  // ---- columns (slot -> arena)
  const char* colbase[VH_MAX_SLOTS];
  uint64_t colstride[VH_MAX_SLOTS];  // bytes between consecutive segments
  uint32_t colpitch[VH_MAX_SLOTS];   // bytes between consecutive rows: the element size, or the record size of a payload projection
  // ---- work decomposition
  const uint32_t* seg_rows;  // per segment: rows to scan (0 = skipped)
  uint32_t nseg;
  uint32_t unit_rows;        // multiple of 4096
  uint32_t units_per_seg;
  uint32_t total_units;
  // ---- grouping
  int32_t ngroup;
  int32_t nmetric;
  int32_t key_words;         // hash path: u64 words per key
  int32_t nxcd;              // dense-global: number of private table copies
  VhGroupDev g[VH_MAX_GROUP];
  VhMetricDev m[VH_MAX_METRIC];
  uint64_t G;                // dense: number of group ids
  uint64_t xcd_stride;       // dense-global: elements between per-XCD copies
  uint8_t* present;          // dense: G (x nxcd) presence bytes
  int32_t present_carrier;   // >= 0: metric whose SOP_ADD32P state doubles as the presence flag
  int32_t pad3;
  uint32_t lds_present_off;  // DENSE_LDS: byte offset of presence words
  uint32_t lds_bytes;        // DENSE_LDS / HASH with an LDS front table: bytes of the table (queues sit behind it)
  // HASH: per-block open-addressing front table in LDS (0 slots = none). Rows whose key finds a slot there are
  // aggregated with LDS atomics; the block merges its table into the HBM table once, at the end.
  uint32_t lds_hash_slots;   // power of two
  uint32_t lds_hkeys_off;    // byte offset of the u64 key array (metric states at m[j].lds_off, one per slot)
  // ---- hash
  uint64_t* hkeys;           // capacity(+1) x key_words
  uint32_t* htags;           // wide keys: slot state words
  uint64_t hmask;            // capacity - 1
  uint32_t max_probe;
  uint32_t pad_probe;
  // single-word keys: key and metric states of a slot are ONE record of hrec_bytes (key at +0, m[j].state = table + the
  // state's offset inside the record), so an insert and its updates touch one line; 0 = separate arrays (wide keys)
  uint32_t hrec_bytes;
  uint32_t pad_hrec;
  // ---- bitset metrics (COUNT DISTINCT): per-row id sets mirrored as CSR per segment. Every id of a
  // surviving row is inserted into a device-wide open-addressing SET keyed by (group, id); the first
  // insertion of a pair bumps the group's cardinality (the metric's u64 state). Union semantics of
  // `_j |= metrics._j` + cardinality() (src/codegen/db/store.cc:153-155, src/util/bitset.h:26-67), no host pass.
  int32_t nbitset;
  int32_t bs_wide[VH_MAX_BITSET];                 // 1: uint64 ids, 0: uint32 ids
  const uint64_t* const* bs_offs[VH_MAX_BITSET];  // [nseg] -> offsets[rows + 1]
  const void* const* bs_vals[VH_MAX_BITSET];      // [nseg] -> ids
  // a bitset metric in the FILTER: the predicate sees the row's cardinality (filter.cc:216,235) = offsets[r + 1] - offsets[r];
  // VhProgOp::type() is VH_BITSET32 / VH_BITSET64 and slot() indexes this array
  const uint64_t* const* fbs_offs[VH_MAX_BITSET]; // [nseg] -> offsets[rows + 1]
  uint64_t* dset_keys[VH_MAX_BITSET];             // narrow ids: (group << 32 | id); wide ids: 2 words per slot
  uint32_t* dset_tags[VH_MAX_BITSET];             // wide ids only
  uint64_t dset_mask[VH_MAX_BITSET];
  // ---- partitioned aggregation (DENSE_PART): survivors become (gid, values) tuples, radix-partitioned
  // by gid >> part_shift into extents in HBM; a second kernel aggregates each partition in LDS. Group-id spaces of more
  // than VH_MAX_PART LDS-sized ranges take two levels: phase 1 partitions by gid >> part_shift (= agg_shift + 6) into
  // pool 1, part_split_kernel splits every partition 64 ways again into pool 2 (each partition owns a contiguous range
  // of pool-2 extents, sized on the device from what phase 1 produced), and phase 2 aggregates ranges of 1 << agg_shift.
  int32_t npart;             // partitions of phase 1, <= VH_MAX_PART
  int32_t part_shift;        // phase 1: partition = gid >> part_shift
  int32_t tw;                // 64-bit words per tuple (word 0 low half = gid)
  int32_t ext_tuples;        // tuples per extent (a tile's run of one partition never straddles extents)
  int32_t nlevel;            // 1 or 2
  int32_t shape;             // 0, or the plan shape the compacting kernel's drain is specialised for (vh_consume_fast)
  int32_t agg_shift;         // groups per LDS table of phase 2 = 1 << agg_shift (one level: == part_shift)
  int32_t nfine;             // LDS-sized ranges phase 2 aggregates (one level: == npart)
  int32_t ext_tuples2;       // pool 2: tuples per extent
  int32_t ext_stride;        // pool 1: tuples from one extent's first place to the next one's — ext_tuples, or ext_tuples + 8 (one more 128-byte
                             // line of two-word tuples): the waves of phase 1 open their extents together and fill them at the same
                             // pace, so with extents a power of two apart every wave's stores of the moment agree in the address bits that
                             // pick the HBM channel (profiles/r03/NOTES.md, "Where the tuple pool lands")
  uint64_t* tuples;          // max_extents x ext_tuples x tw words
  uint16_t* extent_missing;  // [max_extents] tuples NOT filled in an extent (0 = full)
  uint8_t* extent_part;      // [max_extents] partition an extent belongs to (0xFF: never opened). A plain store when the extent is
                             // opened; phase 2 scans the tags. (A per-partition list needed a returning atomic on npart hot counters.)
  uint64_t* tuples2;         // pool 2 (two levels only), same tuple layout
  uint16_t* extent_missing2;
  uint8_t* extent_part2;     // tag = the SUB-partition (0..63) inside the owning partition's range
  uint32_t* l2;              // [0..npart]: first pool-2 extent of partition p's range (prefix sums); [VH_L2_NEXT + p]: extents handed out of it
  uint32_t max_extents;
  // Hashed partitioning, HEAVY RANGES. A range (level-A digit a, level-B digit d: the top 16 bits of the mixed key) that holds more tuples than a
  // block should walk alone, or whose ids overflow the block's LDS set, is not aggregated by the ranges' kernel: it sets bit (a << 8 | d) here
  // (nullptr: no such escape — the attempt is void as before) and counts itself in counters[11] / its tuples' bound in counters[12]. The host then
  // runs the plain hash organisation over the rows of exactly those ranges (heavy_only below) and appends its groups to the result.
  uint32_t* heavy_mark;
  const uint32_t* heavy_only;   // the plain hash scan of a heavy pass: a survivor counts only if bit (mix(key) >> 48) is set here (nullptr: every survivor)
  uint32_t* part_count;      // one-level DENSE_PART whose phase 1 went through the ring writer: tuples per partition, counted at the scan blocks' ends — phase 2's blocks
                             // (and private table copies) are shared out by these counts (vh_part_shares); nullptr: every partition the same number of blocks
  uint32_t pos_levels;       // the ring writer's pools: extents per (block, digit) stream that lie at POSITIONS (pool 1: phase 1 of DENSE_PART) ...
  uint32_t slice_levels_cap; // (tests: an upper bound for the positional levels of the slices the device lays out — 0 sends every tuple through the overflow regions; ~0u otherwise)
  uint32_t pos_levels2;      // ... (the second pool: level A of the hashed partitioning, written by the scan); what lies behind level * streams is the
                             // pool's shared overflow region, handed out through counters[9] / counters[10] (vh_ring_add_tb)
  uint32_t ext_waves;        // pool 1, phase 1: waves of the scan launch when they take their extent chunks by position — the k-th chunk of wave w is
                             // chunk k * ext_waves + w — instead of from the shared cursor (0: shared cursor). Every wave's first drain opens
                             // extents, and 3 072 returning atomics on ONE address at the start of the kernel queue up behind each other for
                             // ~80 us whatever the table's size (profiles/r04/NOTES.md, "A fixed cost inside phase 1")
  uint32_t max_extents2;
  // ---- hashed partitioning (HASH organisation, many groups: hash_part_agg_kernel). Survivors become (mixed key, payload)
  // tuples — the mixed key is a BIJECTION of the packed 64-bit group key (vh_splitmix64 / vh_unmix64), so equal keys meet in one
  // partition and no key is ever compared through a lossy hash — radix-partitioned by the mixed key's top bits with the machinery
  // above (gid_shift = 32: the partition digits are taken from the top half of word 0), and aggregated range by range in LDS.
  int32_t hpart;             // 1: this organisation
  int32_t gid_shift;         // bits of tuple word 0 below the 32-bit partition key: 0 (dense gid) or 32 (mixed key)
  int32_t hp_passes;         // sub-ranges a block works through per range (power of two): one LDS table's worth of groups each
  int32_t hp_gslots;         // LDS group table: slots (power of two) ...
  int32_t hp_sslots;         // ... and slots of the LDS (group slot, id) set (0: no bitset metric)
  uint32_t hp_keys_off;      // LDS byte offsets: group keys [hp_gslots + 1] (u64), (group slot, id) set [hp_sslots] (u64);
  uint32_t hp_set_off;       //   metric states at m[j].lds_off [hp_gslots + 1]
  // ---- counters: [0] passed rows, [1] new groups (hash), [2] error flags,
  //                [3] reserved hash slot (key == sentinel) in use, [4] distinct (group, id) pairs,
  //                [5] extent allocation cursor (DENSE_PART, and the stream pool of the hashed partitioning)
  unsigned long long* counters;
};

// VhMetricDev::slot of the virtual row-id column (VH_COL_ROWID): value = (segment << 32) | row
#define VH_SLOT_ROWID 0xFFFFu

#define VH_ERR_RANGE 1ull      // dense digit out of range
#define VH_ERR_HASH_FULL 2ull  // probe limit hit
#define VH_ERR_PART_FULL 4ull  // tuple extents exhausted
#define VH_ERR_HPART_FULL 8ull // hashed partitioning: an LDS table of hash_part_agg_kernel overflowed (more passes, or the plain hash table)
#define VH_ERR_HP_WIDE 16ull   // hashed partitioning, packed tuples: a value or an id needed more bits than the column's recorded min / max said (the plain hash table)

static inline int vh_elem_size(int e) {
  switch (e) {
    case VH_U8: case VH_I8: return 1;
    case VH_U16: case VH_I16: return 2;
    case VH_U32: case VH_I32: case VH_F32: return 4;
    case VH_U64: case VH_I64: case VH_F64: return 8;
    default: return 0;
  }
}

// vh_jit.h — per-query compiled scan kernels: the GPU analogue of the reference's codegen + compiler cache
// (AggQueryGenerator src/codegen/query/agg_query.cc:26-75, ComparisonBuilder src/codegen/query/filter.cc:206-261,
// Compiler::Compile src/codegen/compiler.cc:97-144). Host side; the device-side frame is vh_jit_body.h.
#pragma once
#include "vh_internal.h"
#include <string>
#include <vector>

#define VJ_MAX_PRED 8       // distinct predicate columns held packed in registers
#define VJ_MAX_NV 96        // ... and the registers they may take per lane and wave step (a 1-byte column: 4, a 4-byte one: 16)
#define VJ_MAX_COLS 8       // group columns / metrics of a generated drain

struct VhJitPred { int slot, type, width; };      // width: bytes per element as streamed (a narrow copy of a u32 column: 1 or 2)
struct VhJitCol {           // one gathered value of a survivor
  int slot = 0, type = 0, pitch = 0;   // pitch: bytes between consecutive rows (element size, or the record size of a projection)
  int rec = -1, off = 0;               // rec >= 0: member of payload projection `rec`, at byte `off` of its record
  int stored = 0;                      // ... where it takes `stored` bytes (0: its element size; fewer: a compressed projection, low bytes of the value)
  int bits = 0;                        // bit-field record (4 or 8 bytes, VhPack::bits): `off` is the field's BIT offset in the record word, `stored` its bits
  int sext = 0, rowid = 0, bitset = 0;             // sign-extend to 64 bits (dense digits, signed MIN / MAX); the virtual row-id column
  // group columns
  int gran = VH_T_NONE, nroll = 0, micro = 0, key_word = 0, key_shift = 0;
  int roll_unit[VH_MAX_ROLLUP] = {};
  // metrics
  int sop = 0, tword = 0, tshift = 0;
  int tbits = 0;                       // packed tuples of the hashed partitioning: bits of the value in the tuple's second word (0: not packed)
};
// Everything the generated text depends on — and nothing else (no literals, no addresses, no row counts): the cache key.
struct VhJitShape {
  int mode = 0, block = 256, scope = 0, xcd = 0, carrier = -1, tw = 0, key_words = 1, lds_hash = 0, gid32 = 0;
  int hpart = 0, bitset_j = -1;         // HASH: hashed partitioning (vh_hpart.h); the metric that is a bitset (its ids travel in the tuples, two at a time)
  int bs_off32 = 0;                     // hashed partitioning with a bitset metric: P.bs_offs points at 32-bit offsets
  int part_ring = 0;                    // DENSE_PART: phase 1 writes its tuples through the block's ring writer (vj_part_ring_add): partitions it keeps lines for (16 / 64), 0 = tuples of three or more words, appended piece by piece (vh_part_direct_add)
  int hp_fan = 0;                       // hashed partitioning: the scan block writes the level-A pool itself (vj_fan_add; 1024-thread blocks, one per CU) — no stream pool, no level-A scatter
  int tuple4 = 0;                       // DENSE_PART: the one-word tuple is 4 bytes (VhPlanDev::tuple4)
  int gid_bits = 0;                     // DENSE_PART: one-word tuples — the gid's bits at the bottom of word 0 (0: the usual two or more words)
  int hp_agg_waves = 0;                  // ... its aggregation kernel should leave room for this many waves per SIMD (what its LDS tables allow): a register bound for the compiler (0: none)
  int hp_pack = 0, hp_pbits = 0, hp_idbits = 0;   // ... in PACKED 16-byte tuples: word 1 = payload (hp_pbits) | two ids (hp_idbits each) | ids that count << 61 | ids only << 63
  int lanes = 0;                        // DENSE_LDS, most rows pass: no compaction — a lane keeps its own 4 consecutive rows per sub-step, group and metric columns come in
                                        // with the same 16-byte vector loads as the predicates (all of a step's at once), passing rows update the LDS table directly
  int npred = 0;
  VhJitPred pred[VJ_MAX_PRED];
  // Bit-packed predicate projection (vh_table_predpack): the predicate columns are bit fields of ONE word per row, kept as `pp_nplanes` byte
  // planes of 1 or 2 bytes per row (plane q holds the word's bits from pp_pos[q] on). The kernel streams the planes instead of the columns
  // (C3: 3 bytes per row instead of 5 through narrow copies, 12 through the arenas), puts a row's word together in registers and compares
  // the fields in place. pred[k].slot / width are then unused; pp_off / pp_bits say where predicate column k lies in the word.
  int pp_nplanes = 0;
  struct Plane { int slot, width, pos; } pp_plane[4];
  int pp_off[VJ_MAX_PRED] = {}, pp_bits[VJ_MAX_PRED] = {};
  // ... or BIT-SLICED (VhPredPack::sliced): bit b of the row word is a plane of its own, one 32-bit word per 32 rows; predicate column k is
  // planes pp_off[k] .. pp_off[k] + pp_bits[k]. A lane owns 32 consecutive rows per step, loads the planes of the columns the filter reads
  // and evaluates every comparison bit-serially on 32 rows at once (vj_bits_rel): a few bitwise operations per plane instead of a compare,
  // a ballot and a rank per row — and 1 bit per row and bit of information streamed. pp_slot: the projection's slot (its pitch = bytes
  // between planes). The compacting kernels only; pred[] keeps the columns' own slots for the no-compaction form.
  int pp_sliced = 0, pp_slot = -1;
  // Streamed payload: every group / metric value of the plan is a bit field of ONE 4-byte record per row (a bit-field projection, VhPack::bits).
  // Instead of queueing a survivor's ROW and gathering its record afterwards (a random 128-byte line per survivor: at 5 % selectivity 81 % of
  // the projection's lines are fetched anyway, at the rate random lines come in), the scan streams the records with the predicate planes —
  // 16-byte loads like any 4-byte column — and queues the survivor's RECORD: the drain unpacks it out of LDS, no gather at all.
  int qpay = 0, qpay_slot = -1;          // 4: on (the record's bytes); the slot the records are streamed from
  std::vector<VhProgOp> prog;           // postfix filter; VhProgOp::pslot indexes pred[], ::lit the literal pool
  int nlits = 0;
  int ng = 0, nm = 0;
  VhJitCol g[VJ_MAX_COLS], m[VJ_MAX_COLS];
  int ablate = 0;                       // measurement builds only (VH_JIT_ABLATE): 1 = no gathers, 2 = no sink
  std::string key() const;
};

struct VhJitKernel {
  hipModule_t mod = nullptr;
  hipFunction_t fn = nullptr;
  hipFunction_t fn_agg = nullptr;   // hashed partitioning: the ranges' aggregation compiled for the same shape (`<name>_hpagg`, vh_hpart.h)
  size_t agg_lds_set = 0;
  hipFunction_t fn_pagg = nullptr;  // DENSE_PART: phase 2 compiled for the same shape (`<name>_pagg`, vj_part_agg); nullptr: the pre-built part_agg_kernel
  size_t pagg_lds_set = 0;
  size_t scan_lds_set = 0;
  std::string name;          // kernel symbol as rocprofv3 prints it
  double compile_ms = 0;     // 0: came out of the disk cache
  int vgprs = 0, sgprs = 0;
};

enum { VH_JIT_OFF = 0, VH_JIT_AUTO = 1, VH_JIT_FORCE = 2 };
int vh_jit_policy();                               // VH_JIT = off | auto (default) | force
uint64_t vh_jit_min_rows();                        // auto: rows a query must scan before a compile is worth it (VH_JIT_MIN_ROWS)
std::string vh_jit_source(const VhJitShape& s, const char* kernel_name);
// The kernel for this shape: memory cache -> disk cache -> hipRTC. nullptr + *err on failure (the caller falls back to the
// pre-built interpreting kernels). Thread-safe; a shape is compiled once.
VhJitKernel* vh_jit_get(const VhJitShape& s, std::string* err);
int vh_jit_occupancy(VhJitKernel* k, int block, size_t lds);
hipError_t vh_jit_launch(VhJitKernel* k, const VhPlanDev& P, int grid, int block, size_t lds, hipStream_t s);
hipError_t vh_jit_launch_hpagg(VhJitKernel* k, const VhPlanDev& P, const void* d_hpargs, int blocks_per_partition, int a_first, int grid, size_t lds, hipStream_t s);
hipError_t vh_jit_launch_pagg(VhJitKernel* k, const VhPlanDev& P, int blocks_per_part, size_t lds, hipStream_t s);
#ifndef VH_HP_AGG_BLOCK
#define VH_HP_AGG_BLOCK 512          // threads of the ranges' aggregation (hp_aggregate_body); -D only for measurement builds (tools/build_variant.py)
#endif
// code object for a shape without loading it (no GPU needed: build-time cache warm-up, CPU tests)
int vh_jit_compile_only(const VhJitShape& s, std::vector<char>* code, std::string* log);

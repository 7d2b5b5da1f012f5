/*
 * viya_hip.h — C ABI of the MI355X-native scan / filter / aggregate executor.
 *
 * This is the drop-in boundary for ONE path of the reference: the body of the
 * generated `viya_query_agg` function between "for (auto* s : segments_copy())"
 * and "stats.aggregated_recs = agg_map.size()"
 *   (reference: src/codegen/query/scan.cc:40-66,168-247; the JIT entry point it
 *    lives in is declared at src/query/runner.h:33-35 and emitted by
 *    src/codegen/query/agg_query.cc:26-75).
 * Built on the same scan: the select and search query functions
 * (viya_query_select / viya_query_search, scan.cc:75-166,249-299: vh_query_select and
 * VH_COL_ROWID below), a HAVING and a top-N that run on the device before the groups
 * are read back (post_agg.cc:77-83, sort.cc:24-75), and the exchange primitives for
 * multi-GPU runs (vh_result_device_buffers, vh_result_partition[_pairs]).
 * Everything here is plain C: opaque handles, POD structs, pointers and sizes.
 * No C++ types, no exceptions and no torch types cross this boundary.
 *
 * Error convention: every function returns 0 on success, a negative VH_E_* code
 * otherwise; vh_last_error() returns a thread-local message for the last
 * failure on the calling thread (reference: C++ exceptions out of the generated
 * function become HTTP 400, src/server/http/service.cc:129-133 — the host shim
 * rethrows a non-zero status as std::runtime_error).
 *
 * Ownership: the library owns device mirrors behind vh_table; the caller owns
 * every host buffer it passes in; vh_result / vh_rows objects are library-allocated
 * and must be released with vh_result_free() / vh_rows_free().
 *
 * Threading ("re-entrant per handle"; the reference runs queries of one table from several read_pool
 * threads, src/db/database.cc:28-34): every entry point may be called from any thread. A query is PLANNED and LAUNCHED
 * under a per-table lock (tens of microseconds of host work) and then runs, and is read back, on its own execution
 * context — a HIP stream, device scratch, pinned staging buffers and events taken from a grow-only pool of the table
 * (VH_MAX_EXEC, default 16) — so queries of different threads on one table overlap on the device. vh_segment_sync*,
 * vh_segment_generate, vh_table_pack/unpack take the same lock and first wait for every launched query that may still
 * read the arenas they replace. With an externally owned stream (vh_set_stream) all contexts share that stream.
 *
 * Lifetime of a vh_result: it OWNS its execution context from vh_query_launch / vh_query_agg until vh_result_free. Its
 * device-side state (what vh_result_finalize, vh_result_device_buffers and vh_result_partition[_pairs] read) and its host
 * view (vh_result_view / vh_result_copy) stay valid for exactly that long, whatever other queries run on the table
 * meanwhile. (Pointers obtained with vh_result_view additionally stay readable after vh_result_free until the
 * second-next query that happens to take the same context: enough for a single-threaded caller that frees before it
 * reads, nothing a multi-threaded one should rely on.) A vh_result must not outlive its table. Holding more live
 * results on one table than the pool may grow to blocks the next launch (and fails after 60 s).
 */
#ifndef VIYA_HIP_H_
#define VIYA_HIP_H_

#ifndef __HIPCC_RTC__ /* (hipRTC, which compiles the per-query kernels, brings its own) */
#include <stddef.h>
#endif
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VH_API __attribute__((visibility("default")))

/* ---- status codes -------------------------------------------------------- */
enum {
  VH_OK = 0,
  VH_E_INVALID = -1,     /* bad argument / unsupported plan                  */
  VH_E_DEVICE = -2,      /* HIP runtime failure                              */
  VH_E_NOMEM = -3,       /* device or host allocation failed                 */
  VH_E_UNSUPPORTED = -4, /* valid in the reference, not yet on the GPU path  */
  VH_E_RANGE = -5        /* data violated a bound the plan relied on         */
};

/* ---- element types --------------------------------------------------------
 * One per C++ storage type the reference can generate for a column
 * (src/db/column.cc:121-153 numeric types; :54-62 dict-code widths;
 *  :328-333 time/microtime; :357-359 bool; :275-286 count width). */
enum vh_elem {
  VH_U8 = 0, VH_U16 = 1, VH_U32 = 2, VH_U64 = 3,
  VH_I8 = 4, VH_I16 = 5, VH_I32 = 6, VH_I64 = 7,
  VH_F32 = 8, VH_F64 = 9,
  VH_BITSET32 = 10, /* util::Bitset<4> per row, mirrored as CSR (offsets,u32) */
  VH_BITSET64 = 11  /* util::Bitset<8> per row, mirrored as CSR (offsets,u64) */
};

/* Column kinds (reference: db::Dimension::DimType src/db/column.h:148-162 and
 * db::Metric::AggregationType :242-254). Dimension kinds matter for segment
 * skipping: only NUMERIC and TIME dimensions carry min/max stats
 * (src/codegen/db/store.cc:171-201, src/codegen/query/filter.cc:263-335). */
enum vh_kind {
  VH_DIM_STRING = 0, VH_DIM_NUMERIC = 1, VH_DIM_TIME = 2, VH_DIM_BOOLEAN = 3,
  VH_METRIC_MAX = 16, VH_METRIC_MIN = 17, VH_METRIC_SUM = 18,
  VH_METRIC_AVG = 19, VH_METRIC_COUNT = 20, VH_METRIC_BITSET = 21,
  VH_METRIC_HIDDEN_COUNT = 22 /* the uint64_t _count[] array that exists when
                                 a table has an AVG metric and no COUNT metric
                                 (src/codegen/db/store.cc:126-129,298-301)   */
};

typedef struct vh_col_desc {
  int32_t kind; /* enum vh_kind */
  int32_t elem; /* enum vh_elem */
} vh_col_desc;

/* 8-byte scalar, bit-compatible with db::AnyNum (src/db/column.h:98-121):
 * the value is stored in the low bytes in the column's own type. */
typedef union vh_anynum {
  uint8_t u8; uint16_t u16; uint32_t u32; uint64_t u64;
  int8_t i8; int16_t i16; int32_t i32; int64_t i64;
  float f32; double f64;
} vh_anynum;

/* ---- filter program --------------------------------------------------------
 * The query::Filter tree (src/query/filter.h:38-134) AFTER FilterFactory has
 * eliminated NOT and sorted children (src/query/filter.cc:36-108), flattened to
 * postfix. Literals are already decoded to the column's type by the host
 * (FilterArgsPacker/ValueDecoder, src/codegen/query/filter.cc:100-204). */
enum vh_fkind {
  VH_F_TRUE = 0, /* EmptyFilter -> "true"      (filter.cc:258-261)           */
  VH_F_REL = 1,  /* (col OP lit)               (filter.cc:206-221)           */
  VH_F_IN = 2,   /* OR of == / AND of !=       (filter.cc:223-241)           */
  VH_F_AND = 3,  /* bitwise & of `count` operands, no short circuit (:243-256)*/
  VH_F_OR = 4    /* bitwise | of `count` operands                            */
};
enum vh_relop { /* same order as query::RelOpFilter::Operator filter.h:60-67 */
  VH_OP_EQ = 0, VH_OP_NE = 1, VH_OP_LT = 2, VH_OP_LE = 3, VH_OP_GT = 4, VH_OP_GE = 5
};
typedef struct vh_filter_node {
  int32_t kind;   /* enum vh_fkind                                           */
  int32_t col;    /* table column index (REL, IN)                            */
  int32_t op;     /* enum vh_relop (REL); for IN: 1 = IN, 0 = NOT IN         */
  int32_t count;  /* IN: number of literals; AND/OR: number of operands      */
  int32_t lit;    /* REL/IN: index of first literal in vh_plan.lits          */
  int32_t reserved;
} vh_filter_node;

/* ---- group-by columns ------------------------------------------------------
 * One per AggTuple::Dimensions field, in QUERY order
 * (src/codegen/db/store.cc:42-50, src/codegen/query/agg_query.cc:50-58). */
enum vh_time_unit { /* util::TimeUnit src/util/time.h:27 */
  VH_T_YEAR = 0, VH_T_MONTH = 1, VH_T_WEEK = 2, VH_T_DAY = 3, VH_T_HOUR = 4,
  VH_T_MINUTE = 5, VH_T_SECOND = 6, VH_T_NONE = 7
};
#define VH_MAX_ROLLUP 8
typedef struct vh_group_col {
  int32_t col;          /* table column index (a dimension)                  */
  int32_t granularity;  /* query-level trunc unit or VH_T_NONE
                           (src/codegen/query/scan.cc:209-214)               */
  int32_t nrollup;      /* rollup rules, sorted as the reference sorts them
                           (`after` descending, src/db/column.cc:346-349)    */
  int32_t rollup_unit[VH_MAX_ROLLUP];  /* trunc unit of rule i              */
  uint64_t rollup_before[VH_MAX_ROLLUP]; /* rule i applies when ts < this
                           (rollup_bN_i, src/codegen/db/rollup.cc:44-95);
                           in the column's unit (seconds or microseconds)    */
  int32_t micro;        /* 1: microtime column (util::Time64)                */
  int32_t reserved;
  uint64_t cardinality; /* string dim: dict->c2v().size(); bool: 2;
                           0 = unknown (library uses per-segment min/max)    */
} vh_group_col;

/* ---- the plan --------------------------------------------------------------*/
enum vh_plan_flags {
  VH_PLAN_FORCE_HASH = 1u << 0,   /* testing: never take the dense path     */
  VH_PLAN_FORCE_GLOBAL = 1u << 1, /* testing: dense table in HBM, not LDS   */
  VH_PLAN_NO_XCD_PRIVATE = 1u << 2,/* testing: one device-scope dense table */
  VH_PLAN_NO_FAST = 1u << 3,      /* testing: always the generic scan kernel */
  VH_PLAN_NO_PART = 1u << 4,      /* testing: direct global atomics instead of
                                     radix-partitioned LDS aggregation        */
  VH_PLAN_NO_CARRIER = 1u << 5,   /* testing: separate presence bytes even when a
                                     32-bit SUM state could carry the flag    */
  VH_PLAN_FORCE_PART = 1u << 6,   /* testing: radix-partitioned aggregation
                                     regardless of the estimated selectivity  */
  VH_PLAN_NO_LANES = 1u << 7,     /* ablation: always compact survivors, even
                                     when most rows pass                      */
  VH_PLAN_FORCE_LANES = 1u << 8,  /* testing: the no-compaction "lanes" kernel
                                     whenever the plan is eligible            */
  VH_PLAN_NO_LDS_HASH = 1u << 9,  /* ablation: hash path without the per-block
                                     LDS front table                          */
  VH_PLAN_NO_HASH_RECORDS = 1u << 10,/* ablation: hash table with separate key and
                                     state arrays even when it is big (>= 4 M slots:
                                     one record per slot for single-word keys)  */
  VH_PLAN_FORCE_HASH_RECORDS = 1u << 11,/* testing: records whatever the size    */
  VH_PLAN_NO_PACK = 1u << 12,     /* ablation: gather group / metric values from the
                                     column arenas even when a payload projection
                                     (vh_table_pack) covers them               */
  VH_PLAN_FORCE_PACK = 1u << 13,  /* testing: gather from a payload projection whatever
                                     the selectivity, building one if none covers
                                     the query's columns                       */
  VH_PLAN_NO_PART2 = 1u << 14,    /* ablation: no second partition level (group-id
                                     spaces of more than 64 LDS-sized ranges stay on
                                     direct global atomics)                    */
  VH_PLAN_NO_SHAPE = 1u << 15,    /* ablation: the generic survivor drain even when the
                                     plan has a shape a specialised one exists for */
  VH_PLAN_NO_NARROW = 1u << 16,   /* ablation: predicate columns from their 4-byte arenas
                                     even when a narrow copy (vh_table_narrow) exists */
  VH_PLAN_NO_JIT = 1u << 17,      /* only the pre-built (interpreting) scan kernels, whatever VH_JIT says */
  VH_PLAN_FORCE_JIT = 1u << 18    /* testing: compile a scan kernel for this plan shape however small the table
                                     (shapes the generator does not cover — bitset metrics, the no-compaction
                                     kernels, very wide plans — still take the pre-built kernels: see
                                     vh_result_info.reserved bit 5); a compile that FAILS is an error,
                                     VH_E_UNSUPPORTED, not a silent fallback */,
  VH_PLAN_NO_HPART = 1u << 19,    /* ablation: never the hashed partitioning of the hash path (many groups: tuples keyed by
                                     a bijective mix of the group key, radix-partitioned, aggregated range by range in LDS) */
  VH_PLAN_FORCE_HPART = 1u << 20, /* testing: hashed partitioning whenever the plan is eligible, however small the table  */
  VH_PLAN_NO_HP_PACK = 1u << 21,  /* ablation: a count-distinct's tuples keep their ids in words of their own (32 bytes) even when payload,
                                     two ids and their count would fit the tuple's second word (16 bytes) */
  VH_PLAN_NO_NARROW_TUPLES = 1u << 23, /* ablation: DENSE_PART keeps two-word tuples even when gid and values would fit one */
  VH_PLAN_NO_PREDPACK = 1u << 24, /* ablation: predicate columns from their arenas / narrow copies even when a bit-packed predicate
                                     projection (vh_table_predpack) holds them */
  VH_PLAN_NO_SLICED = 1u << 27,   /* ablation: a bit-sliced predicate projection is not used (byte planes, narrow copies or the columns answer) */
  VH_PLAN_NO_QPAY = 1u << 25,     /* ablation: a survivor's values are always GATHERED by row, even where the compiled scan could stream the
                                     bit-field records of a projection beside the predicate columns and queue the survivor's record */
  VH_PLAN_FORCE_QPAY = 1u << 26,  /* testing: streamed records whenever the plan is eligible, whatever the selectivity */
  VH_PLAN_CARD32 = 1u << 22       /* the cardinality of a 32-bit-id bitset metric (count distinct) is delivered as a uint32 column instead of
                                     uint64 (it cannot exceed 2^32 - 1): vh_result_state_elem() tells what a state column holds */
};
typedef struct vh_plan {
  const vh_filter_node* filter; int32_t nfilter;   /* postfix program        */
  const vh_anynum* lits;        int32_t nlits;
  const vh_group_col* groups;   int32_t ngroups;
  const int32_t* metrics;       int32_t nmetrics;  /* table column indices in
                                   QUERY order (AggTuple::Metrics fields)    */
  /* Snapshot of SegmentBase::size() per segment taken by the caller under the
   * reference's read locks (src/db/store.h:40-44, src/db/segment.h:34-39).
   * NULL: use the row counts of the last vh_segment_sync of each segment.   */
  const uint64_t* seg_rows;     uint32_t nseg;
  uint32_t flags;
  uint64_t groups_hint;         /* expected number of groups (0 = unknown)   */
  /* Optional HAVING pushed down to the device: same node format as `filter`, literals in
   * `lits`, but `col` is a RESULT column: i < ngroups = group column i, else metric (col - ngroups) of this
   * plan. Semantics of post_agg.cc:77-83: AVG compares its raw sum, a bitset its cardinality. Only groups that
   * pass are returned; vh_result_info.ngroups still counts every group (agg_map.size()). */
  const vh_filter_node* having; int32_t nhaving; int32_t reserved2;
  /* Optional device-side top-N, for `sort` + `limit` whose FIRST sort column is numeric (the
   * reference's INTEGER / FLOAT sort types, src/db/column.h:198-200,270-272; not string, time, boolean or AVG
   * columns, whose order is an order of formatted strings): top_k = skip + limit (0 = off), top_col = RESULT
   * column as in `having`, top_desc = !ascending. The result then holds a SUPERSET of the rows the reference
   * would send — every group whose key ties with or beats the top_k-th, after HAVING — in no particular order;
   * the caller still runs the reference's comparators (sort.cc:24-75) on them, now on few rows. */
  int32_t top_col; int32_t top_desc; uint64_t top_k;
} vh_plan;

/* ---- results ---------------------------------------------------------------*/
typedef struct vh_table vh_table;
typedef struct vh_result vh_result;

enum vh_path { VH_PATH_SCALAR = 0, VH_PATH_DENSE_LDS = 1, VH_PATH_DENSE_GLOBAL = 2,
               VH_PATH_HASH = 3, VH_PATH_DENSE_PART = 4 };

typedef struct vh_result_info {
  uint64_t ngroups;          /* agg_map.size()  -> stats.aggregated_recs     */
  uint64_t scanned_recs;     /* scan.cc:44 — every segment, skipped or not   */
  uint64_t scanned_segments; /* scan.cc:51 — segments that were not skipped  */
  uint64_t passed_recs;      /* rows for which the predicate held            */
  int32_t path;              /* enum vh_path                                 */
  int32_t ngroup_cols;
  int32_t nmetrics;
  int32_t has_hidden_count;  /* AVG selected without COUNT (store.cc:126-129)*/
  float scan_kernel_ms;      /* HIP-event time of the scan/aggregate kernel  */
  float total_ms;            /* HIP-event time launch .. results in host mem */
  uint64_t algorithmic_bytes;/* B_ref of SURVEY §8(d) for this query         */
  uint32_t retries;          /* hash-table regrows                           */
  uint32_t reserved;         /* bit 0: the register-resident fast scan kernel ran; bit 1: its no-compaction "lanes" variant;
                                bit 2: LDS front table of the hash path; bit 3: payload gathered from a projection (vh_table_pack);
                                bit 4: predicate columns streamed from narrow copies (vh_table_narrow);
                                bit 5: a scan kernel compiled for this plan shape ran (vh_jit.hip);
                                bit 6: hashed partitioning of the hash path (vh_hpart.h);
                                bit 7: the projection's records are compressed (integers at the width their values need);
                                bit 8: the hashed partitioning's tuples were packed (16 bytes: payload, two ids and their count in one word);
                                bit 9: (rounds 3-5: a tuple pool placed by measurement; the search is gone, the bit is never set);
                                bit 10: DENSE_PART wrote one-word tuples (gid and values packed into 8 bytes);
                                bit 11: predicate columns streamed as byte planes of a bit-packed predicate projection (vh_table_predpack; bit 4 is set too);
                                bit 13: ... of its BIT-SLICED form (comparisons bit-serial on 32 rows per lane);
                                bit 12: the payload was STREAMED — 4-byte bit-field records beside the predicate columns, a survivor's record queued in its
                                        row's place — not gathered (bits 3 and 7 are set too) */
  uint64_t returned_groups;  /* rows vh_result_copy delivers (= ngroups without HAVING) */
} vh_result_info;

/* Device-side view of a partial (not yet finalised) result, for the caller to
 * run a collective over (RCCL all-reduce of identically indexed dense arrays,
 * SURVEY §8(e)). Only valid for dense paths. */
enum vh_reduce_op { VH_RED_SUM = 0, VH_RED_MIN = 1, VH_RED_MAX = 2 };
typedef struct vh_device_buffer {
  void* ptr; uint64_t count; int32_t elem; /* enum vh_elem */ int32_t reduce; /* enum vh_reduce_op */
} vh_device_buffer;

/* ---- synthetic data (bench / tests): SURVEY §8(d) counter-based generator --
 * value(col c, global row r) = splitmix64(seed ^ (c * 0x9E3779B97F4A7C15) ^ r),
 * reduced to the column's domain. */
enum vh_gen_mode {
  VH_GEN_UNIFORM = 0, /* add + (h % mod)   (floating columns: * scale)       */
  VH_GEN_ROWID = 1,   /* global row index r                                  */
  VH_GEN_CONST = 2,   /* add                                                 */
  /* skewed shapes (tools/skew_probe.py; integer arithmetic only, so the oracle's numpy twin is exact):                      */
  VH_GEN_ZIPF = 3,    /* add + v % mod, v = 2^b - 1 + (lo32(h) % 2^b), b = hi32(h) % (floor(log2 mod) + 1): P(v) ~ 1 / (v + 1) in
                       * octaves, the continuous cousin of Zipf(s = 1) — value 0 alone takes 1 / (floor(log2 mod) + 1) of the rows */
  VH_GEN_SORTED = 4,  /* add + r / mod: non-decreasing in load order, `mod` rows per value (a time-ordered load)                 */
  VH_GEN_HOT = 5      /* add + mod / 2 on the rows where splitmix64(seed ^ r ^ 0x407) % 1000 < param (the SAME rows in every
                       * VH_GEN_HOT column of a table: a hot composite key), add + h % mod elsewhere                             */
};
typedef struct vh_gen_spec {
  int32_t mode; int32_t param; /* VH_GEN_HOT: per mille of hot rows */ uint64_t mod; int64_t add; double scale;
} vh_gen_spec;

/* ---- entry points ----------------------------------------------------------*/

/* Bind the calling process to one GPU. SURVEY 8(b) sketched vh_init(ndev, dev_ids); the deployment unit here is one
 * PROCESS per GPU (north_star; bench.py and vh_comm_* below follow it), so a process names exactly one device.
 * The HIP current device is per thread: every entry point re-binds its calling thread to this device. */
VH_API int vh_init(int device_id);
/* Run all subsequent work of this process on an externally owned hipStream_t
 * (e.g. torch's current stream, so RCCL collectives issued by the caller are
 * ordered with the kernels). NULL is the legacy default stream; VH_OWN_STREAM
 * restores the library's private non-blocking streams (the default after vh_init:
 * one for table maintenance, one per execution context). */
#define VH_OWN_STREAM ((void*)(intptr_t)-1)
VH_API int vh_set_stream(void* hip_stream);
VH_API const char* vh_last_error(void);
VH_API const char* vh_version(void);

/* Mirror of db::Table's storage shape (src/db/table.h:53-91): column list in
 * table order (dimensions, then metrics, then the hidden count if present) and
 * segment capacity (src/db/table.cc:48). */
VH_API int vh_table_create(const vh_col_desc* cols, int32_t ncols,
                           uint64_t segment_rows, uint32_t reserve_segments,
                           vh_table** out);
VH_API void vh_table_destroy(vh_table* t);

/* Copy the first `nrows` rows of segment `seg` into HBM. (SURVEY 8(b) sketched a `version` argument: change tracking —
 * per-segment version + dirty row range — is the CALLER's, who knows what upsert touched (viyadb_amd/host/gpu_aggregate.cc:
 * SyncMirror); the library is told what to copy and stamps the segment itself for its projections.)
 * col_ptrs[c] is the host
 * address of the segment's column array (&segment->d._i[0] / &segment->m._j[0]
 * in the generated Segment class, src/codegen/db/store.cc:214-356); a NULL entry
 * leaves that column's mirror untouched. Bitset columns are not passed here
 * (see vh_segment_sync_bitset). Also refreshes the per-segment min/max stats
 * used for segment skipping. A col_ptrs entry may also be a DEVICE address (unified
 * addressing): that is how exchanged partial aggregates become a segment without a host hop. */
VH_API int vh_segment_sync(vh_table* t, uint32_t seg, uint64_t nrows,
                           const void* const* col_ptrs);
/* Dirty-range form of vh_segment_sync: upsert appends to the last segment and updates
 * metrics of existing rows in place (src/codegen/db/upsert.cc:384-411), so a caller that tracks the touched
 * row range per segment only ships that range. col_ptrs[c] is still the BASE of the segment's column array;
 * rows [row_first, row_first + nrows) are copied and the segment then has `new_size` valid rows. */
VH_API int vh_segment_sync_range(vh_table* t, uint32_t seg, uint64_t row_first, uint64_t nrows,
                                 uint64_t new_size, const void* const* col_ptrs);
/* What ONE upsert batch did to the table, in ONE call. The reference's upsert merges a micro-batch into the store row by
 * row: a row whose dimensions are new is appended to the LAST segment, a row that exists has its metrics updated IN PLACE in whatever
 * segment holds it (src/codegen/db/upsert.cc:384-411) — a batch of 100 K rows can dirty hundreds of segments by a few rows each. An item
 * is one contiguous row range of one segment; col_ptrs[c] is the BASE of the segment's column array as in vh_segment_sync (NULL: leave
 * that column alone; bitset columns are never passed). With VH_SYNC_METRICS_ONLY only the metric columns of the range are shipped
 * (dimensions of an existing row never change). After the call the segment has `new_size` rows (ranges must not leave gaps behind the
 * rows already mirrored; several items may name one segment).
 *
 * The whole batch is ONE kernel launch: every contiguous (column, range) run is pulled into its arena by one workgroup — straight out
 * of the caller's memory over PCIe when that memory was registered (vh_host_register; no host-side copy at all), through a pinned ring
 * for small runs of unregistered memory, behind a hipMemcpyAsync for big ones — which also takes the run's min / max; the per-segment
 * stats (SegmentStats, store.cc:171-201, and the value widths the derived layouts are sized from) are WIDENED by what the runs hold
 * (a range that covers every mirrored row of its segment replaces them). Under upsert a segment's true range only grows for dimensions;
 * a metric updated in place may leave stats wider than its values — never narrower. The call returns once the sources may be reused
 * (registered memory: immediately — the device reads it until the batch's event, which the next planner, sync or stats reader waits
 * for; the caller's writer may keep updating rows meanwhile exactly as the reference's readers race its writer). No per-segment host
 * synchronisation, no wait for running queries unless an arena has to move. */
enum { VH_SYNC_METRICS_ONLY = 1u, /* only metric columns (kind >= VH_METRIC_MAX) of the range changed */
       VH_SYNC_DEVICE_SRC = 2u    /* col_ptrs are DEVICE addresses (exchanged partials) */ };
typedef struct vh_sync_item {
  uint32_t seg, flags;
  uint64_t row_first, nrows, new_size;
  const void* const* col_ptrs;
} vh_sync_item;
VH_API int vh_table_sync_batch(vh_table* t, const vh_sync_item* items, uint32_t nitems);
/* Make [base, base + bytes) of the caller's memory readable by the device in place (pages pinned and mapped: hipHostRegister). A ViyaDB
 * Segment is one heap object whose column arrays never move (store.cc:203-356): registered once when it is first seen, every later sync
 * of it is zero-copy. Idempotent per base; VH_E_DEVICE when the runtime refuses (the caller carries on unregistered). */
VH_API int vh_host_register(const void* base, uint64_t bytes);
VH_API int vh_host_unregister(const void* base);
/* Counters of the batched syncs of a table since it was created: batches, runs (kernel descriptors), bytes pulled out of registered memory,
 * bytes through the pinned ring, bytes through hipMemcpyAsync. Any pointer may be NULL. */
VH_API int vh_table_sync_stats(vh_table* t, uint64_t* batches, uint64_t* runs, uint64_t* bytes_pulled, uint64_t* bytes_staged, uint64_t* bytes_dma);
/* Bitset metric column of one segment as CSR: offsets[nrows+1], values[].   */
VH_API int vh_segment_sync_bitset(vh_table* t, uint32_t seg, int32_t col,
                                  uint64_t nrows, const uint64_t* offsets,
                                  const void* values);
/* Fill segments [seg_first, seg_first+nseg) directly in HBM with synthetic
 * data; every segment gets `rows_per_seg` rows (<= segment_rows). The global
 * row index of row i of segment s is row_base + (s - seg_first) * rows_per_seg + i. */
VH_API int vh_segment_generate(vh_table* t, uint32_t seg_first, uint32_t nseg,
                               uint64_t rows_per_seg, uint64_t row_base,
                               const vh_gen_spec* specs, uint64_t seed);
/* Payload projection ("pack"): a second, row-major mirror of a FEW columns — record r of a segment holds the values
 * of row r, widest column first, padded to a power of two <= 64 B. The reference has no such thing: its Segment is
 * column arrays only (src/codegen/db/store.cc:214-356) and its loop touches `tuple_dims._i[idx]` / `tuple_metrics._j[idx]`
 * of a passing row in as many cache lines as there are columns (scan.cc:220-241). On the GPU that is the dominant
 * traffic of a selective GROUP BY (C3: four 128 B lines per survivor for 20 useful bytes), so the layout of the
 * mirror is widened: queries whose group + metric columns are all in one projection, and whose filter passes few
 * rows, gather a survivor's values from ONE record. Results are identical; vh_segment_sync* keeps projections
 * coherent (a changed segment is re-packed from HBM before the next query that uses it). Cost: rows x record bytes
 * of HBM (C3: 32 GB next to the 60 GB table). The library also builds one by itself for a column set it has seen in
 * VH_AUTO_PACK (default 3, 0 = never) selective queries when a quarter of the device stays free; vh_table_unpack
 * drops them all.
 * Compressed records: where the per-query compiled kernels run (VH_JIT != off), a projection stores every integer column at
 * the width its values need over the whole table (1, 2, 4 or 8 bytes; dimensions from their SegmentStats, metrics from a
 * min / max pass when the projection is built) — the compiled kernel widens them back, sign and all, so results do not
 * change. C3's record shrinks from 32 to 8 bytes (m0 in 4, d0 in 2, d1 and count in 1 each): 8 GB instead of 32, and 16
 * records per 128-byte line instead of 4, so fewer lines are fetched for the same survivors. A synced value that no longer
 * fits voids the projection (pack_kernel checks every value it stores); the next query that wants it rebuilds it wider.
 * The pre-built kernels do not read compressed records: a plan that ends up on them gathers from a plain projection or
 * the arenas. vh_table_pack decides by itself (compressed when the table is big enough for the compiled kernels to be used on it,
 * see VH_JIT_MIN_ROWS); vh_table_pack_ex asks for one form. The environment's VH_PACK_PLAIN=1 keeps every automatic choice plain. Analogue in the reference: the per-query g++ compile that is cached on first use
 * (src/query/runner.cc:45-64) — work done once for a query shape, outside its steady-state cost. */
VH_API int vh_table_pack(vh_table* t, const int32_t* cols, int32_t ncols);       /* = vh_table_pack_ex(.., VH_PACK_AUTO) */
enum { VH_PACK_AUTO = 0, VH_PACK_PLAIN = 1, VH_PACK_COMPRESSED = 2 };
VH_API int vh_table_pack_ex(vh_table* t, const int32_t* cols, int32_t ncols, uint32_t form);
VH_API int vh_table_unpack(vh_table* t);

/* Narrow copies of predicate columns. A column every query filters on is read in full by every query: its bytes are the
 * floor of the scan. For an unsigned 32-bit column whose values fit 8 or 16 bits (dictionary codes of small
 * dictionaries, small uint dimensions: the reference itself sizes a string dimension's codes by its cardinality,
 * src/db/column.h) the table keeps a second copy at that width and the register-resident scan kernels stream it
 * instead; values are widened in registers, so every comparison is the one the 4-byte column would get. Copies follow
 * vh_segment_sync* like projections do, are dropped when a synced value no longer fits, and are built unasked for a
 * column the VH_AUTO_NARROW-th (default 3, 0 = never) query filters on while a quarter of the device stays
 * free. Columns that do not qualify are skipped silently. vh_table_unpack drops them too. */
VH_API int vh_table_narrow(vh_table* t, const int32_t* cols, int32_t ncols);
/* Bit-packed predicate projection. A narrow copy still spends whole bytes on a column: C3's three predicate columns carry 2 + 10 + 10 =
 * 22 bits of information per row and cost 1 + 2 + 2 = 5 bytes through narrow copies (12 from the arenas). The projection keeps, for a SET of
 * predicate columns, one word per row with every column as a bit field at the bits its recorded min / max need (non-negative integers, 32
 * bits in all at most), stored as byte planes of 2 or 1 bytes per row (C3: 3 bytes per row). Only the per-query compiled scan kernels read it:
 * they stream the planes, put a row's word together in registers and compare the fields in place — the comparisons are those the columns
 * would get. It follows vh_segment_sync* / vh_table_sync_batch by row range like the other derived layouts, is dropped when a synced value
 * needs more bits than its field (the stats say so), and is built unasked for a column set the compiled kernel filters on for the
 * VH_AUTO_NARROW-th time (instead of narrow copies of those columns) while a quarter of the device stays free. A column set that would not
 * get smaller is skipped silently. vh_table_unpack drops these too. */
VH_API int vh_table_predpack(vh_table* t, const int32_t* cols, int32_t ncols);      /* = vh_table_predpack_ex(.., VH_PREDPACK_AUTO) */
/* Two forms. BYTES: the word as byte planes (above): every kernel form of the compiled scan reads it, row by row. SLICED (what AUTO builds):
 * one plane per BIT of the word, one bit per row — a predicate column of n bits costs exactly n bits per row (C3: 22 bits = 2.75 bytes) —
 * and the compiled COMPACTING scan evaluates a comparison on 32 rows of a lane at once, bit-serially over the column's planes (a handful
 * of bitwise operations per plane instead of a compare, a ballot and a rank per row): the filter's share of the scan's instructions falls
 * to a fraction. The no-compaction form (most rows pass) reads the columns themselves. */
enum { VH_PREDPACK_AUTO = 0, VH_PREDPACK_BYTES = 1, VH_PREDPACK_SLICED = 2 };
VH_API int vh_table_predpack_ex(vh_table* t, const int32_t* cols, int32_t ncols, uint32_t form);
/* Copy a mirrored column back to the host (tests). */
VH_API int vh_segment_read(vh_table* t, uint32_t seg, int32_t col, uint64_t nrows,
                           void* dst);
VH_API int vh_table_info(vh_table* t, uint32_t* nseg, uint64_t* segment_rows,
                         uint64_t* device_bytes);
/* Per-segment min/max of a column as mirrored (SegmentStats). */
VH_API int vh_segment_stats(vh_table* t, uint32_t seg, int32_t col,
                            vh_anynum* min_out, vh_anynum* max_out);

/* The hot path. vh_query_agg = vh_query_launch + vh_result_finalize. */
VH_API int vh_query_agg(vh_table* t, const vh_plan* plan, vh_result** out);
/* Launch only: partial aggregate stays in HBM (multi-GPU: reduce, then finalise). */
VH_API int vh_query_launch(vh_table* t, const vh_plan* plan, vh_result** out);
VH_API int vh_result_device_buffers(vh_result* r, vh_device_buffer* bufs,
                                    int32_t max_bufs, int32_t* nbufs);
VH_API int vh_result_finalize(vh_result* r);
/* Multi-GPU, hash path (sparse keys: partial tables are not identically indexed, so they
 * cannot be reduced in place). Regroups the emitted rows of a FINALISED result by
 * owner = mix(key columns) % nparts, in HBM: rows of owner p are
 * [part_offsets[p], part_offsets[p+1]) of every returned column buffer, i.e. one contiguous
 * send buffer per destination for an all-to-all (grouped ncclSend/ncclRecv). The owner
 * merges what it receives by re-aggregation (vh_segment_sync accepts device pointers, so the
 * received columns become a segment of a temporary table without leaving HBM). Replaces
 * the reference's cluster merge over HTTP + temp-table upsert
 * (src/cluster/query/agg_runner.cc:83-140). bufs: the key columns, then the plan's metrics
 * in plan order, then the hidden count if the plan carries one; `reduce` of a key column is
 * -1, of a metric the operator that merges two partial states. part_offsets: nparts + 1. */
VH_API int vh_result_partition(vh_result* r, uint32_t nparts, uint64_t* part_offsets,
                               vh_device_buffer* bufs, int32_t max_bufs, int32_t* nbufs);
/* Count-distinct across GPUs: a rank's partial for a bitset metric is the SET of distinct
 * (group, id) pairs it saw ("(group,value) pairs take the same route;
 * distinct-count is finalised on the owner"). Returns, for plan metric `metric`, the pairs as
 * the group's key columns + an id column (u32 / u64), regrouped by the owner of the GROUP
 * (the same function as vh_result_partition). The owner loads what it receives with
 * vh_segment_sync (key columns, device addresses) + vh_segment_sync_ids_device (one id per
 * row) and runs the aggregate query again: its count-distinct is the merged cardinality.
 * A result of the hashed partitioning (vh_result_info.reserved bit 6) kept no device-wide set: its
 * pairs are read out of its tuple pool and may repeat — the owner's set union does not mind. */
VH_API int vh_result_partition_pairs(vh_result* r, int32_t metric, uint32_t nparts,
                                     uint64_t* part_offsets, vh_device_buffer* bufs,
                                     int32_t max_bufs, int32_t* nbufs);
VH_API int vh_segment_sync_ids_device(vh_table* t, uint32_t seg, int32_t col, uint64_t nrows,
                                      const void* d_ids);
/* Host copy of `bytes` bytes of a vh_device_buffer (e.g. to put an exchanged partial on a wire that is not
 * xGMI: SURVEY 8(f)-4, viyadb_amd/host/partial_state.cc). */
VH_API int vh_device_read(void* dst, const void* device_src, uint64_t bytes);

/* ---- one query over a table sharded across GPUs (north_star: "segments shard naturally one-per-GPU ...
 * with a final RCCL reduce of per-GPU partial aggregates over xGMI") -------------------------------------------------
 * One process per GPU; every rank mirrors ITS segments in its own vh_table (same column list everywhere) and all ranks
 * call vh_query_agg_sharded with the same plan. Replaces the reference's cluster merge — worker queries over HTTP, TSV
 * text re-upserted into a temporary table on the controller (src/cluster/query/agg_runner.cc:83-140) — inside one node.
 *
 *  1. plan agreement: the table organisation must not depend on what one shard happens to hold. Each rank summarises
 *     what its planner would look at (per group column the min / max of the segments it will scan, rows to scan, the
 *     selectivity probe's counts, re-plan requests of the previous attempt); the summaries are all-gathered and every
 *     rank plans from the MERGED summary: same dense digit ranges, same dense-vs-hash decision, same partitioning,
 *     same state layout on every rank.
 *  2. every rank scans its shard (the kernels of vh_query_agg).
 *  3. a 128-byte all-reduce carries error flags and row counters: if ANY rank overflowed a table or met a value outside
 *     the agreed range, ALL ranks re-plan together and run again — never a mismatched collective.
 *  4a. dense organisations: the identically indexed partial tables are reduced to `root` in place (ncclReduce per state
 *     array: SUM / MIN / MAX in the state's own type, unsigned 32/64-bit MIN/MAX included) and root emits the groups
 *     (HAVING / top-N run there, on merged states).
 *  4b. hash organisation (sparse keys) and count-distinct: every rank regroups its groups — and, per bitset metric, its
 *     distinct (group, id) pairs — by owner = mix(key columns) % world (vh_result_partition[_pairs]), one grouped
 *     ncclSend/ncclRecv exchange moves every column straight into the column arenas of a temporary table on the owner,
 *     which merges by re-aggregation (SUM of sums and counts, MIN / MAX, set union — the reference's own algebra,
 *     agg_runner.cc:66-76) with HAVING / top-N applied to the merged groups; the owners' rows are then gathered on root.
 *
 * root >= 0: the merged groups are delivered on that rank; the other ranks get a result with no rows. root = -1 (hash
 * organisation only): results stay sharded, every rank gets the groups it owns. vh_result_info's row counters are
 * global sums on every rank; ngroups counts all groups of all owners.
 *
 * The transport is RCCL (vh_comm_init; the id comes from vh_comm_unique_id on one rank and reaches the others by whatever
 * means the host has — MPI, a file, torch.distributed's store) or a table of callbacks (vh_comm_init_custom: tests run two
 * ranks on one GPU over gloo that way; a deployment could put MPI behind it). librccl.so.1 is resolved at run time. */
typedef struct vh_comm vh_comm;
#define VH_COMM_ID_BYTES 128
typedef struct vh_comm_ops {
  void* ctx;
  /* every rank contributes `bytes` host bytes; recv = world x bytes, rank-major. Returns 0 or an error code. */
  int (*allgather_host)(void* ctx, const void* send, void* recv, uint64_t bytes);
  /* in-place on a device buffer of `count` elements of enum vh_elem `elem`, op = enum vh_reduce_op; root < 0: all-reduce,
   * else the result need only be valid on root. Enqueued on (or ordered after) `stream`. */
  int (*reduce_device)(void* ctx, void* buf, uint64_t count, int32_t elem, int32_t op, int32_t root, void* stream);
  /* all-to-all-v over `ncols` device columns at once: rows [send_off[p], send_off[p+1]) of send[c] (esize[c] bytes each) go
   * to rank p and land at rows [recv_off[q], ...) of recv[c] on the receiver, q = the sender. Offsets have world + 1 entries. */
  int (*alltoallv_device)(void* ctx, int32_t ncols, const void* const* send, void* const* recv, const uint32_t* esize,
                          const uint64_t* send_off, const uint64_t* recv_off, void* stream);
} vh_comm_ops;
VH_API int vh_comm_unique_id(void* id_out /* VH_COMM_ID_BYTES */);
VH_API int vh_comm_init(const void* id, int32_t rank, int32_t world, vh_comm** out);
VH_API int vh_comm_init_custom(const vh_comm_ops* ops, int32_t rank, int32_t world, vh_comm** out);
VH_API void vh_comm_destroy(vh_comm* c);
/* What the communicator actually is, read back from it (not from what the caller asked for): the rank count and this rank as
 * the transport reports them (RCCL: ncclCommCount / ncclCommUserRank), the HIP device the communicator is bound to
 * (ncclCommCuDevice; the current device for a callback transport) and that device's PCI bus id — so a launcher can print
 * which GPUs a scaling run used and whether its data plane was RCCL or a fallback. */
enum vh_comm_transport { VH_COMM_RCCL = 1, VH_COMM_CALLBACKS = 2 };
typedef struct vh_comm_info_t {
  int32_t transport;      /* enum vh_comm_transport */
  int32_t nranks, rank;   /* as the transport reports them */
  int32_t device;         /* HIP device ordinal inside this process */
  char pci_bus_id[32];    /* "0000:c1:00.0" */
} vh_comm_info_t;
VH_API int vh_comm_info(vh_comm* c, vh_comm_info_t* out);
VH_API int vh_query_agg_sharded(vh_table* t, const vh_plan* plan, vh_comm* comm, int32_t root, vh_result** out);

/* ---- the other two FilterBasedQuery kinds on the same scan ------------
 *
 * search (viya_query_search, src/codegen/query/scan.cc:249-299): the distinct values of one
 * dimension among the passing rows, in first-occurrence order. That is an aggregate query
 * GROUP BY the dimension with MIN over the row's storage position: put VH_COL_ROWID in
 * vh_plan.metrics; its state is a u64 (segment << 32 | row). The caller orders the groups by
 * it, matches the term against the decoded values and applies the limit
 * (viyadb_amd/host/gpu_aggregate.cc: GpuSearch).
 *
 * select (viya_query_select, scan.cc:75-166): the passing rows themselves, in storage order,
 * through the reference's skip / limit window, as one dense array per selected column (the
 * column's own element type; a bitset column yields its per-row cardinality as u64).
 * Formatting stays with the caller. `limit` 0 = none. The reference's `break` leaves the tuple
 * loop only, so after the limit is reached each later segment still contributes its first
 * passing row; that is reproduced. */
#define VH_COL_ROWID (-2)
typedef struct vh_rows vh_rows;
typedef struct vh_select_plan {
  const vh_filter_node* filter; int32_t nfilter;
  const vh_anynum* lits;        int32_t nlits;
  const int32_t* cols;          int32_t ncols;   /* table column indices, output order */
  const uint64_t* seg_rows;     uint32_t nseg;   /* size() snapshot, as in vh_plan      */
  uint32_t flags;
  uint64_t skip, limit;
} vh_select_plan;
typedef struct vh_rows_info {
  uint64_t nrows;             /* stats.output_recs                          */
  uint64_t scanned_recs, scanned_segments;
  uint64_t passed_recs;       /* rows that satisfied the filter             */
  float kernel_ms, total_ms;
} vh_rows_info;
VH_API int vh_query_select(vh_table* t, const vh_select_plan* plan, vh_rows** out);
VH_API int vh_rows_get_info(vh_rows* r, vh_rows_info* info);
/* cols[c] = host array (pinned, owned by the vh_rows) of nrows elements of column c. */
VH_API int vh_rows_view(vh_rows* r, const void** cols);
VH_API void vh_rows_free(vh_rows* r);

/* Pays a plan shape's first-use costs NOW instead of on its first queries: compiles the scan kernel for the shape (the reference does
 * the same when a query shape is first seen: Compiler::Compile, src/codegen/compiler.cc:97-144, reported as QueryStats::compile_time), builds
 * the payload projection and narrow predicate copies a selective query would get after VH_AUTO_PACK / VH_AUTO_NARROW uses, and — here only —
 * finds the derived layouts a place by measurement: which physical pages a projection and its predicate planes were given decides up to 15 % of
 * the scan that reads them (same bytes, same kernel: 1.07 or 1.23 ms per 1 B rows of C3), and nothing a process can read says which, so the
 * layouts are copied to up to VH_PREPARE_PLACE (default 8, 0 = never) other allocations — both, the planes alone, the projections alone, in turn —,
 * three queries each, about a second at most, and the fastest place is kept
 * (the others are released before returning; skipped when the copies would not leave a quarter of the device free). Runs the plan up to
 * three times before that, discards the rows; *info (may be NULL) describes the last run: vh_result_info.reserved says what a steady-state
 * query of this shape runs on (compiled kernel, projection, narrow copies). Layouts that were built without a vh_table_prepare behind them
 * (by the library after VH_AUTO_PACK / VH_AUTO_NARROW uses, or by vh_table_pack / _predpack alone) stay where hipMalloc put them. */
VH_API int vh_table_prepare(vh_table* t, const vh_plan* plan, vh_result_info* info);
/* The derived layouts moved to fresh device memory (which: 1 projections, 2 predicate planes, 0 = both) — same contents, other pages; what
 * vh_table_prepare does per candidate place, for callers that measure by themselves. */
VH_API int vh_table_relocate(vh_table* t, uint32_t which);
VH_API int vh_result_get_info(vh_result* r, vh_result_info* info);
/* Symbol(s) of the scan kernel(s) this query launched, spelled as rocprofv3 prints them ("scan_agg_fast_kernel<4, 256, 4, 3> +
 * part_agg_kernel<1024>"): what a profile of the same command must show. Valid until vh_result_free. */
VH_API const char* vh_result_kernel(vh_result* r);
/* Element type (enum vh_elem) of the state column vh_result_view / vh_result_copy deliver for metric j of the plan: the metric column's own type;
 * VH_U64 for a count distinct and for the virtual row id — VH_U32 for a count distinct over 32-bit ids when the plan carried VH_PLAN_CARD32. < 0: no such metric. */
VH_API int vh_result_state_elem(vh_result* r, int32_t metric);
/* Copy out: key_cols[i] receives ngroups elements of group column i's element
 * type; state_cols[j] receives ngroups elements of metric j's element type
 * (bitset metrics: uint64 cardinality); hidden_count (may be NULL) receives the
 * summed uint64 _count. Row k of every array belongs to the same group. */
VH_API int vh_result_copy(vh_result* r, void* const* key_cols,
                          void* const* state_cols, uint64_t* hidden_count);
/* Zero-copy variant of vh_result_copy: pointers into the table's pinned staging buffer, valid until the
 * second-next query on the same table (two buffers alternate) or vh_table_destroy. */
VH_API int vh_result_view(vh_result* r, const void** key_cols, const void** state_cols,
                          const uint64_t** hidden_count);
VH_API void vh_result_free(vh_result* r);

/* Streaming-read ceiling of this device in bytes/s (a plain 16 B/lane read
 * kernel over `bytes` of HBM) — the "achievable" figure quoted next to every
 * roofline fraction. */
VH_API int vh_measure_read_bandwidth(uint64_t bytes, int32_t iters, double* bytes_per_sec);

/* Test hook for the per-query compiled scan kernels (viyadb_amd/csrc/vh_jit.hip — the GPU analogue of the reference's
 * codegen + Compiler::Compile, src/codegen/compiler.cc:97-144): writes the HIP text for canonical plan shape `which`
 * (0..6) into `text`, compiles it with hipRTC for gfx950 — no GPU needed — and stores the code object at `hsaco_path`
 * (NULL: nowhere). VH_E_INVALID: no such shape; VH_E_UNSUPPORTED: the compile failed (`text` then starts with the log). */
VH_API int vh_jit_selftest(int32_t which, const char* hsaco_path, char* text, uint64_t text_bytes);

#ifdef __cplusplus
}
#endif
#endif /* VIYA_HIP_H_ */

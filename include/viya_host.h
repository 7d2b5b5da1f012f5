/*
 * viya_host.h — C facade over the C++ host shim (viyadb_amd/host/), for tests and bindings.
 *
 * The host shim itself is C++ and mirrors the reference's own interfaces
 * (db::Database / db::Table / input::SimpleLoader / query::AggregateQuery / query::RowOutput,
 * see viyadb_amd/host/viya_db.h and viya_query.h); this facade is the JSON-in / rows-out surface the
 * reference exposes over HTTP (POST /tables, /load, /query — src/server/http/service.cc:46-162),
 * flattened to C so that pytest can drive it through ctypes.
 *
 * Rows are returned as one malloc'ed buffer: fields separated by 0x1F, rows by 0x1E.
 */
#ifndef VIYA_HOST_H_
#define VIYA_HOST_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#define VDB_API __attribute__((visibility("default")))

typedef struct vdb vdb;

enum { VDB_OK = 0, VDB_E_INVALID_ARGUMENT = 1, /* std::invalid_argument in the reference */
       VDB_E_RUNTIME = 2 /* std::runtime_error / anything else */ };

typedef struct vdb_stats {          /* query::QueryStats (src/query/stats.h:35-58) + GPU extras */
  uint64_t scanned_segments, scanned_recs, aggregated_recs, output_recs, passed_recs;
  double compile_time, whole_time, scan_kernel_ms, device_total_ms;
  int32_t path, reserved;
} vdb_stats;

VDB_API int vdb_open(const char* config_json, int device, vdb** out);        /* db::Database(config) */
VDB_API void vdb_close(vdb* db);
VDB_API int vdb_create_table(vdb* db, const char* table_json);              /* Database::CreateTable */
/* One process per GPU of a node, each holding ITS rows of the tables: from here on aggregate queries run over all ranks'
 * rows (the same vdb_query on every rank, rows on rank 0; vh_query_agg_sharded underneath). `vh_comm_handle` comes from
 * vh_comm_init / vh_comm_init_custom (include/viya_hip.h) and stays the caller's; NULL leaves the node. Dictionary codes of
 * string dimensions must agree between the ranks (one ingest order, or numeric / time dimensions). */
VDB_API int vdb_join_node(vdb* db, void* vh_comm_handle);
/* input::SimpleLoader::Load: rows in the facade's row encoding; now < 0 = wall clock
 * (the reference's VIYA_TEST_ROLLUP_TS test hook, src/codegen/db/rollup.cc:47-49) */
VDB_API int vdb_load(vdb* db, const char* table, const char* rows, size_t rows_len, int64_t now);
VDB_API int vdb_query(vdb* db, const char* query_json, int64_t now, char** rows_out, size_t* rows_len, vdb_stats* stats);
/* Cluster aggregate with binary partial states (replaces the TSV hop of
 * src/cluster/query/agg_runner.cc:83-140 — see viyadb_amd/host/partial_state.h for the wire layout).
 * Worker: the query's partial state as one malloc'ed blob (free with vdb_free). Controller: merges the
 * workers' blobs on the GPU and finishes the query (having / sort / skip / limit / formatting). */
VDB_API int vdb_query_partial(vdb* db, const char* query_json, int64_t now, char** blob_out, size_t* blob_len, vdb_stats* stats);
VDB_API int vdb_query_merge(vdb* db, const char* query_json, const char* const* blobs, const size_t* blob_lens, int32_t nblobs,
                            char** rows_out, size_t* rows_len, vdb_stats* stats);
/* The text of the generated drop-in translation unit for (table, aggregate query) — viya::shim::codegen::AggQueryText, include/viya_shim.h —
 * or, with query_json == NULL, the line the generated viya_upsert_do gains (UpsertHookText). malloc'ed: vdb_free. For tests and tooling. */
VDB_API int vdb_shim_text(const char* table_json, const char* query_json, char** text_out, size_t* text_len);
VDB_API int vdb_table_info(vdb* db, const char* table, uint64_t* nsegments, uint64_t* first_segment_size);
VDB_API void vdb_free(char* p);
VDB_API const char* vdb_last_error(void);
#ifdef __cplusplus
}
#endif
#endif

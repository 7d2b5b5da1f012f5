// viya_host_c.cc — C facade (include/viya_host.h) over the C++ host shim.
#include "../../include/viya_host.h"
#include "../../include/viya_shim.h"

#include <cstdlib>
#include <cstring>

#include "viya_query.h"

namespace vdbimpl {
thread_local std::string g_err;
template <typename F> int guard(F&& f) {
  try {
    f();
    return VDB_OK;
  } catch (const std::invalid_argument& e) {
    g_err = std::string("invalid_argument: ") + e.what();
    return VDB_E_INVALID_ARGUMENT;
  } catch (const std::exception& e) {
    g_err = e.what();
    return VDB_E_RUNTIME;
  } catch (...) {
    g_err = "unknown exception";
    return VDB_E_RUNTIME;
  }
}
std::vector<std::vector<std::string>> decode_rows(const char* p, size_t n) {
  std::vector<std::vector<std::string>> rows;
  if (!n) return rows;
  std::vector<std::string> cur;
  std::string field;
  for (size_t i = 0; i < n; ++i) {
    const char c = p[i];
    if (c == 0x1F) { cur.push_back(field); field.clear(); }
    else if (c == 0x1E) { cur.push_back(field); field.clear(); rows.push_back(cur); cur.clear(); }
    else field += c;
  }
  return rows;
}
}  // namespace vdbimpl

struct vdb {
  std::unique_ptr<viya::db::Database> db;
};

extern "C" {
const char* vdb_last_error(void) { return vdbimpl::g_err.c_str(); }
void vdb_free(char* p) { free(p); }

int vdb_open(const char* config_json, int device, vdb** out) {
  return vdbimpl::guard([&] {
    if (device >= 0) setenv("VIYA_HIP_DEVICE", std::to_string(device).c_str(), 0);
    auto* h = new vdb();
    h->db.reset(new viya::db::Database(viya::util::Config(std::string(config_json ? config_json : "{}")), device));
    *out = h;
  });
}
void vdb_close(vdb* db) { delete db; }

int vdb_shim_text(const char* table_json, const char* query_json, char** text_out, size_t* text_len) {
  return vdbimpl::guard([&] {
    // query_json == NULL: the upsert hook (viya::shim::codegen::UpsertHookText)
    const std::string t = query_json ? viya::shim::codegen::AggQueryText(table_json ? table_json : "{}", query_json) : viya::shim::codegen::UpsertHookText();
    *text_out = (char*)malloc(t.size() + 1);
    memcpy(*text_out, t.data(), t.size());
    (*text_out)[t.size()] = 0;
    if (text_len) *text_len = t.size();
  });
}

int vdb_join_node(vdb* db, void* vh_comm_handle) {
  return vdbimpl::guard([&] { db->db->JoinNode(vh_comm_handle); });
}

int vdb_create_table(vdb* db, const char* table_json) {
  return vdbimpl::guard([&] { db->db->CreateTable(viya::util::Config(std::string(table_json))); });
}

int vdb_load(vdb* db, const char* table, const char* rows, size_t rows_len, int64_t now) {
  return vdbimpl::guard([&] { db->db->Load(table, vdbimpl::decode_rows(rows, rows_len), now); });
}

int vdb_query(vdb* db, const char* query_json, int64_t now, char** rows_out, size_t* rows_len, vdb_stats* stats) {
  return vdbimpl::guard([&] {
    viya::query::MemoryRowOutput out;
    viya::query::QueryStats st = db->db->Query(viya::util::Config(std::string(query_json)), out, now);
    std::string buf;
    for (auto& r : out.rows()) {
      for (auto& f : r) { buf += f; buf += (char)0x1F; }
      buf += (char)0x1E;
    }
    if (rows_out) {
      *rows_out = (char*)malloc(buf.size() + 1);
      memcpy(*rows_out, buf.data(), buf.size());
      (*rows_out)[buf.size()] = 0;
    }
    if (rows_len) *rows_len = buf.size();
    if (stats) {
      memset(stats, 0, sizeof(*stats));
      stats->scanned_segments = st.scanned_segments; stats->scanned_recs = st.scanned_recs;
      stats->aggregated_recs = st.aggregated_recs; stats->output_recs = st.output_recs; stats->passed_recs = st.passed_recs;
      stats->compile_time = st.compile_time; stats->whole_time = st.whole_time;
      stats->scan_kernel_ms = st.scan_kernel_ms; stats->device_total_ms = st.device_total_ms; stats->path = st.path;
    }
  });
}

static void fill_stats(vdb_stats* stats, const viya::query::QueryStats& st) {
  if (!stats) return;
  memset(stats, 0, sizeof(*stats));
  stats->scanned_segments = st.scanned_segments; stats->scanned_recs = st.scanned_recs;
  stats->aggregated_recs = st.aggregated_recs; stats->output_recs = st.output_recs; stats->passed_recs = st.passed_recs;
  stats->compile_time = st.compile_time; stats->whole_time = st.whole_time;
  stats->scan_kernel_ms = st.scan_kernel_ms; stats->device_total_ms = st.device_total_ms; stats->path = st.path;
}

int vdb_query_partial(vdb* db, const char* query_json, int64_t now, char** blob_out, size_t* blob_len, vdb_stats* stats) {
  return vdbimpl::guard([&] {
    viya::query::QueryStats st;
    std::string blob = db->db->QueryPartial(viya::util::Config(std::string(query_json)), st, now);
    *blob_out = (char*)malloc(blob.size() + 1);
    memcpy(*blob_out, blob.data(), blob.size());
    *blob_len = blob.size();
    fill_stats(stats, st);
  });
}

int vdb_query_merge(vdb* db, const char* query_json, const char* const* blobs, const size_t* blob_lens, int32_t nblobs,
                    char** rows_out, size_t* rows_len, vdb_stats* stats) {
  return vdbimpl::guard([&] {
    std::vector<std::string> partials;
    for (int32_t i = 0; i < nblobs; ++i) partials.emplace_back(blobs[i], blob_lens[i]);
    viya::query::MemoryRowOutput out;
    viya::query::QueryStats st = db->db->QueryMerge(viya::util::Config(std::string(query_json)), partials, out);
    std::string buf;
    for (auto& r : out.rows()) {
      for (auto& f : r) { buf += f; buf += (char)0x1F; }
      buf += (char)0x1E;
    }
    if (rows_out) {
      *rows_out = (char*)malloc(buf.size() + 1);
      memcpy(*rows_out, buf.data(), buf.size());
      (*rows_out)[buf.size()] = 0;
    }
    if (rows_len) *rows_len = buf.size();
    fill_stats(stats, st);
  });
}

int vdb_table_info(vdb* db, const char* table, uint64_t* nsegments, uint64_t* first_segment_size) {
  return vdbimpl::guard([&] {
    viya::db::Table* t = db->db->GetTable(table);
    if (nsegments) *nsegments = t->segments().size();
    if (first_segment_size) *first_segment_size = t->segments().empty() ? 0 : t->segments()[0]->size();
  });
}
}

// ingest_bench.cc — what the mirror costs under ingest (VERDICT r04 "missing" #3): profiles/r05/ingest.json.
//
//  A. C-ABI level, C3 at full size (1000 segments x 1 M rows generated in HBM, query shape prepared): one upsert batch of 100 K rows
//     dirtying 1 / 100 / 1000 segments — metrics of existing rows (32 B / row) or whole rows (60 B / row) — through vh_table_sync_batch,
//     out of registered host memory (pulled in place) and out of ordinary memory (pinned ring / DMA): time until the call returns, until the
//     rows are resident, the first query after it (derived layouts follow) and the second.
//  B. shim level (include/viya_shim.h), 1000 small segments: what a query pays for the generated function's pass over segments_copy()
//     when nothing changed (Pin + Sync per segment, the batch assembly in Run), and the first load with and without registration.
//
// build: g++ -std=c++17 -O2 tools/ingest_bench.cc -Iinclude -Lviyadb_amd -lviya_host -lviya_hip -Wl,-rpath,$PWD/viyadb_amd -o /tmp/ingest_bench
#include <viya_hip.h>
#include <viya_shim.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <sys/mman.h>

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CHECK(x) do { if ((x) != VH_OK) { fprintf(stderr, "%s failed: %s\n", #x, vh_last_error()); exit(1); } } while (0)

static const int NCOL = 12;
static const vh_col_desc kCols[NCOL] = {{VH_DIM_NUMERIC, VH_U32}, {VH_DIM_NUMERIC, VH_U32}, {VH_DIM_NUMERIC, VH_U32}, {VH_DIM_NUMERIC, VH_U32}, {VH_DIM_NUMERIC, VH_U32},
                                        {VH_DIM_NUMERIC, VH_U32}, {VH_DIM_NUMERIC, VH_U32}, {VH_METRIC_SUM, VH_I64}, {VH_METRIC_MAX, VH_I64}, {VH_METRIC_COUNT, VH_U32},
                                        {VH_METRIC_SUM, VH_F64}, {VH_METRIC_MIN, VH_U32}};
static const int kEsize[NCOL] = {4, 4, 4, 4, 4, 4, 4, 8, 8, 4, 8, 4};
static const vh_gen_spec kGen[NCOL] = {{VH_GEN_UNIFORM, 0, 1000, 0, 1.0}, {VH_GEN_UNIFORM, 0, 100, 0, 1.0}, {VH_GEN_UNIFORM, 0, 4, 0, 1.0}, {VH_GEN_UNIFORM, 0, 1000, 0, 1.0},
                                       {VH_GEN_UNIFORM, 0, 1000, 0, 1.0}, {VH_GEN_UNIFORM, 0, 1u << 20, 0, 1.0}, {VH_GEN_ROWID, 0, 1, 0, 1.0}, {VH_GEN_UNIFORM, 0, 1001, 0, 1.0},
                                       {VH_GEN_UNIFORM, 0, 1000000, 0, 1.0}, {VH_GEN_UNIFORM, 0, 3, 1, 1.0}, {VH_GEN_UNIFORM, 0, 10000, 0, 0.01}, {VH_GEN_UNIFORM, 0, 1000000, 0, 1.0}};

struct C3Plan {
  vh_filter_node f[4]; vh_anynum lits[3]; vh_group_col g[2]; int32_t m[2]; vh_plan p;
  C3Plan() {
    memset(this, 0, sizeof(*this));
    f[0] = {VH_F_REL, 2, VH_OP_EQ, 1, 0, 0}; f[1] = {VH_F_REL, 3, VH_OP_LT, 1, 1, 0}; f[2] = {VH_F_REL, 4, VH_OP_GE, 1, 2, 0}; f[3] = {VH_F_AND, 0, 0, 3, 0, 0};
    lits[0].u64 = 1; lits[1].u64 = 447; lits[2].u64 = 553;
    for (int i = 0; i < 2; ++i) { g[i].col = i; g[i].granularity = VH_T_NONE; }
    m[0] = 7; m[1] = 9;
    p.filter = f; p.nfilter = 4; p.lits = lits; p.nlits = 3; p.groups = g; p.ngroups = 2; p.metrics = m; p.nmetrics = 2; p.groups_hint = 100000;
  }
};

static double query_ms(vh_table* t, const vh_plan* p, uint64_t* groups = nullptr) {
  const double a = now_ms();
  vh_result* r = nullptr;
  CHECK(vh_query_agg(t, p, &r));
  const double b = now_ms();
  if (groups) { vh_result_info info; CHECK(vh_result_get_info(r, &info)); *groups = info.ngroups; }
  vh_result_free(r);
  return b - a;
}

static bool g_quick = false;      // profiling runs: registered sources, 1000 dirty segments only, no shim part
static void part_a(uint32_t nseg, uint64_t seg_rows) {
  vh_table* t = nullptr;
  CHECK(vh_table_create(kCols, NCOL, seg_rows, nseg, &t));
  for (uint32_t s = 0; s < nseg; s += 100) CHECK(vh_segment_generate(t, s, std::min<uint32_t>(100, nseg - s), seg_rows, (uint64_t)s * seg_rows, kGen, 42));
  C3Plan plan;
  vh_result_info info;
  CHECK(vh_table_prepare(t, &plan.p, &info));
  std::vector<double> steady;
  uint64_t groups0 = 0;
  for (int i = 0; i < 12; ++i) steady.push_back(query_ms(t, &plan.p, &groups0));
  std::sort(steady.begin(), steady.end());
  printf(" \"table\": {\"segments\": %u, \"rows_per_segment\": %llu, \"prepared_flags\": %u, \"steady_query_ms\": %.4f, \"groups\": %llu},\n \"batches\": [\n", nseg, (unsigned long long)seg_rows, info.reserved, steady[steady.size() / 2], (unsigned long long)groups0);
  const uint64_t batch_rows = 100000;
  bool first = true;
  for (int registered = 1; registered >= 0; --registered)
    for (int whole_rows = 0; whole_rows <= 1; ++whole_rows)
      for (uint32_t dirty : {1u, 100u, 1000u}) {
        if (dirty > nseg || (g_quick && (dirty != 1000u || !registered))) continue;
        const uint64_t n = batch_rows / dirty;
        // host memory of the touched rows only: rows [0, n) of every touched segment, read back from the mirror so that the batch changes
        // nothing a query could see (col_ptrs are the BASES of the segment's column arrays: rows [0, n) start there)
        size_t per_seg = 0;
        for (int c = 0; c < NCOL; ++c) per_seg += (n * kEsize[c] + 63) / 64 * 64;
        const size_t total = per_seg * dirty;
        char* host = static_cast<char*>(mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
        std::vector<const void*> ptrs((size_t)dirty * NCOL);
        std::vector<vh_sync_item> items(dirty);
        const uint32_t stride = nseg / dirty;
        uint64_t bytes = 0;
        for (uint32_t k = 0; k < dirty; ++k) {
          const uint32_t seg = k * stride;
          char* at = host + (size_t)k * per_seg;
          for (int c = 0; c < NCOL; ++c) {
            CHECK(vh_segment_read(t, seg, c, n, at));
            ptrs[(size_t)k * NCOL + c] = at;
            if (whole_rows || kCols[c].kind >= VH_METRIC_MAX) bytes += n * kEsize[c];
            at += (n * kEsize[c] + 63) / 64 * 64;
          }
          items[k] = vh_sync_item{seg, whole_rows ? 0u : (uint32_t)VH_SYNC_METRICS_ONLY, 0, n, seg_rows, &ptrs[(size_t)k * NCOL]};
        }
        double reg_ms = 0;
        if (registered) { const double a = now_ms(); CHECK(vh_host_register(host, total)); reg_ms = now_ms() - a; }
        double call_ms = 1e9, resident_ms = 1e9, q1 = 0, q2 = 0;
        for (int rep = 0; rep < 5; ++rep) {
          vh_anynum lo, hi;
          CHECK(vh_segment_stats(t, 0, 0, &lo, &hi));           // nothing pending
          const double a = now_ms();
          CHECK(vh_table_sync_batch(t, items.data(), dirty));
          const double b = now_ms();
          CHECK(vh_segment_stats(t, 0, 0, &lo, &hi));           // waits for the batch's event and merges its stats: the rows are resident
          const double c = now_ms();
          uint64_t g1 = 0;
          const double f1 = query_ms(t, &plan.p, &g1), f2 = query_ms(t, &plan.p);
          if (g1 != groups0) { fprintf(stderr, "groups changed: %llu != %llu\n", (unsigned long long)g1, (unsigned long long)groups0); exit(1); }
          if (c - a < resident_ms) { call_ms = b - a; resident_ms = c - a; q1 = f1; q2 = f2; }
        }
        const double pcie_ms = bytes / 55e9 * 1e3;
        printf("%s  {\"rows\": %llu, \"dirty_segments\": %u, \"what\": \"%s\", \"source\": \"%s\", \"bytes\": %llu, \"call_ms\": %.4f, \"resident_ms\": %.4f, \"pcie_ms_at_55GBps\": %.4f, "
               "\"resident_over_pcie\": %.2f, \"rows_per_s\": %.3e, \"first_query_after_ms\": %.4f, \"second_query_after_ms\": %.4f, \"register_ms\": %.3f}",
               first ? "" : ",\n", (unsigned long long)(n * dirty), dirty, whole_rows ? "whole rows (60 B)" : "metrics in place (32 B)", registered ? "registered (pulled in place)" : "unregistered (ring / DMA)",
               (unsigned long long)bytes, call_ms, resident_ms, pcie_ms, resident_ms / pcie_ms, n * dirty / (resident_ms * 1e-3), q1, q2, reg_ms);
        first = false;
        fflush(stdout);
        if (registered) CHECK(vh_host_unregister(host));
        munmap(host, total);
      }
  printf("\n ],\n");
  vh_table_destroy(t);
}

// ---- B: the shim's per-query pass over 1000 segments
static const char kTable[] = R"j({"name":"c3","segment_size":4096,"dimensions":[{"name":"d0","type":"uint"},{"name":"d1","type":"uint"},{"name":"d2","type":"uint"},{"name":"d3","type":"uint"},{"name":"d4","type":"uint"},{"name":"d5","type":"uint"},{"name":"id","type":"uint"}],"metrics":[{"name":"m0","type":"long_sum"},{"name":"m1","type":"long_max"},{"name":"count","type":"count"},{"name":"m3","type":"double_sum"},{"name":"m4","type":"uint_min"}]})j";
static const char kQuery[] = R"j({"type":"aggregate","table":"c3","dimensions":["d0","d1"],"metrics":["m0","count"],"filter":{"op":"and","filters":[{"op":"eq","column":"d2","value":"1"},{"op":"lt","column":"d3","value":"447"},{"op":"ge","column":"d4","value":"553"}]}})j";
constexpr size_t CAP = 4096;
struct Seg { uint32_t d[7][CAP]; int64_t m0[CAP], m1[CAP]; uint32_t count[CAP]; double m3[CAP]; uint32_t m4[CAP]; size_t size; };
static uint64_t g_rows = 0;
static void send(void*, const std::vector<std::string>&) { ++g_rows; }

static void part_b(int pin) {
  setenv("VIYA_SHIM_PIN", pin ? "1" : "0", 1);
  const uint32_t nseg = 1000;
  std::vector<Seg*> segs(nseg);
  uint64_t x = 88172645463325252ull;
  auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  for (auto& s : segs) {
    s = new Seg();
    for (size_t r = 0; r < CAP; ++r) {
      s->d[0][r] = rnd() % 1000; s->d[1][r] = rnd() % 100; s->d[2][r] = rnd() % 4; s->d[3][r] = rnd() % 1000; s->d[4][r] = rnd() % 1000; s->d[5][r] = rnd() % (1u << 20); s->d[6][r] = (uint32_t)r;
      s->m0[r] = rnd() % 1001; s->m1[r] = rnd() % 1000000; s->count[r] = 1 + rnd() % 3; s->m3[r] = (rnd() % 10000) * 0.01; s->m4[r] = rnd() % 1000000;
    }
    s->size = CAP;
  }
  int key = 0;
  const void* table_key = pin ? (const void*)&key : (const void*)&g_rows;       // (two shadows: one per variant)
  auto call = [&](double* loop_ms, double* total_ms, viya::shim::Stats* st) {
    const double a = now_ms();
    viya::shim::Session* s = viya::shim::Open(table_key, kTable, kQuery);
    const double l0 = now_ms();
    for (uint32_t i = 0; i < nseg; ++i) {
      Seg* sg = segs[i];
      const void* cols[] = {sg->d[0], sg->d[1], sg->d[2], sg->d[3], sg->d[4], sg->d[5], sg->d[6], sg->m0, sg->m1, sg->count, sg->m3, sg->m4};
      viya::shim::Pin(s, i, sg, sizeof(Seg));
      viya::shim::Sync(s, i, sg->size, cols);
    }
    const double l1 = now_ms();
    uint64_t fa[3] = {1, 447, 553};
    viya::shim::Run(s, fa, 3, nullptr, 0, 0, 0, send, nullptr, st);
    viya::shim::Release(s);
    *loop_ms = l1 - l0; *total_ms = now_ms() - a;
  };
  double loop_ms, total_ms;
  viya::shim::Stats st{};
  call(&loop_ms, &total_ms, &st);
  const double first_total = total_ms, first_sync = st.sync_ms;
  std::vector<double> loops, syncs, totals;
  for (int i = 0; i < 30; ++i) { call(&loop_ms, &total_ms, &st); loops.push_back(loop_ms); syncs.push_back(st.sync_ms); totals.push_back(total_ms); }
  std::sort(loops.begin(), loops.end()); std::sort(syncs.begin(), syncs.end()); std::sort(totals.begin(), totals.end());
  // one in-place update in every segment, then the query
  for (uint32_t i = 0; i < nseg; ++i) { segs[i]->m0[17] += 1; viya::shim::Touch(table_key, i, 17, 18); }
  call(&loop_ms, &total_ms, &st);
  printf("  {\"pin\": %d, \"segments\": %u, \"bytes\": %zu, \"first_call_ms\": %.3f, \"first_call_sync_ms\": %.3f, \"clean_pass\": {\"pin_sync_loop_us\": %.2f, \"batch_assembly_us\": %.2f, \"whole_call_ms\": %.4f, \"scan_kernel_ms\": %.4f}, "
         "\"after_1000_touched_rows\": {\"sync_ms\": %.4f, \"whole_call_ms\": %.4f}}",
         pin, nseg, sizeof(Seg) * (size_t)nseg, first_total, first_sync, loops[loops.size() / 2] * 1e3, syncs[syncs.size() / 2] * 1e3, totals[totals.size() / 2], st.scan_kernel_ms, st.sync_ms, total_ms);
  viya::shim::Close(table_key);
  for (auto* s : segs) delete s;
}

int main(int argc, char** argv) {
  const uint32_t nseg = argc > 1 ? (uint32_t)atoi(argv[1]) : 1000;
  const uint64_t seg_rows = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1000000;
  g_quick = argc > 3 && !strcmp(argv[3], "quick");
  CHECK(vh_init(0));
  printf("{\n \"tool\": \"tools/ingest_bench.cc\",\n");
  part_a(nseg, seg_rows);
  if (g_quick) { printf(" \"shim\": []\n}\n"); return 0; }
  printf(" \"shim\": [\n");
  part_b(1);
  printf(",\n");
  part_b(0);
  printf("\n ]\n}\n");
  return 0;
}

// Does the speed class of a tuple pool follow the ORDER in which physical memory was handed out?
//
// profiles/r03/NOTES.md: a partitioning scan — read streams + record gathers + whole-line stores into a tuple pool — runs in one of two
// speeds depending on where the pool lies relative to what is being read; hipMalloc, power-of-two hipMalloc and a VMM mapping land in either
// class alike. There the READ side always came from hipMalloc. Here everything — the stream source, the record array, the pool — is
// built from physical chunks (hipMemCreate) mapped into reservations (hipMemAddressReserve / hipMemMap), the chunks created in a
// CONTROLLED order, and the same access mix is timed (place_calib.hip's kernel). Orders:
//   seg_after   all source chunks, all record chunks, then the pool's chunks           (pool physically behind everything it competes with)
//   seg_before  the pool's chunks first, then sources and records
//   interleave  one pool chunk after every k-th source chunk                           (pool physically scattered among the pages being read)
//   spaced      sources, records, a spacer of S GB that stays allocated, then the pool
// Every order is built `reps` times from scratch (everything released in between), with two chunk sizes. If the class is a function of the
// order, a deterministic layout exists; if orders scatter alike, it is not visible at this level either.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/vmm_order tools/experiments/vmm_order.hip ; run: /tmp/vmm_order [reps] [filler_gb]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; return x ^ (x >> 31); }

__global__ __launch_bounds__(256) void place_kernel(const u32x4* __restrict__ src, uint64_t n16, u32x4* __restrict__ dst, uint64_t lines, int mode,
                                                    int reads_per_store, unsigned long long* sink, const uint64_t* __restrict__ rec, uint64_t nrec) {
  uint32_t acc = 0;
  const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x, nt = (uint64_t)gridDim.x * 256;
  const uint64_t wave = g >> 6;
  const uint32_t lane = threadIdx.x & 63;
  uint64_t it = 0;
  for (uint64_t i = g; i < n16; i += nt, ++it) {
    if (mode & 1) { const u32x4 v = __builtin_nontemporal_load(src + i); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if ((mode & 4) && lane < 10) acc ^= (uint32_t)rec[mix(i * 3 + 1) % nrec];
    if ((mode & 2) && it % (uint64_t)reads_per_store == 0) {
      const uint64_t line = mix(wave * 0x9E3779B97F4A7C15ull + it * 8 + (lane >> 3)) % lines;
      u32x4 v; v.x = (uint32_t)i; v.y = lane; v.z = acc; v.w = 7;
      dst[line * 8 + (lane & 7)] = v;
    }
  }
  if (acc == 0x12345u) atomicAdd(sink, 1ull);
}

struct Region {
  void* va = nullptr; size_t bytes = 0, chunk = 0;
  std::vector<hipMemGenericAllocationHandle_t> h;
  size_t mapped = 0;
};
static hipMemAllocationProp g_prop;
static int region_reserve(Region& r, size_t bytes, size_t chunk) {
  r.bytes = bytes; r.chunk = chunk; r.mapped = 0; r.h.clear();
  CHECK(hipMemAddressReserve(&r.va, bytes, chunk, nullptr, 0));
  return 0;
}
static int region_add_chunk(Region& r) {        // the next `chunk` bytes of the region get fresh physical memory NOW
  if (r.mapped >= r.bytes) return 0;
  hipMemGenericAllocationHandle_t h;
  CHECK(hipMemCreate(&h, r.chunk, &g_prop, 0));
  CHECK(hipMemMap(static_cast<char*>(r.va) + r.mapped, r.chunk, 0, h, 0));
  r.h.push_back(h);
  r.mapped += r.chunk;
  return 0;
}
static int region_finish(Region& r) {
  hipMemAccessDesc acc{}; acc.location = g_prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  CHECK(hipMemSetAccess(r.va, r.bytes, &acc, 1));
  return 0;
}
static int region_free(Region& r) {
  if (!r.va) return 0;
  CHECK(hipMemUnmap(r.va, r.mapped));
  for (auto h : r.h) CHECK(hipMemRelease(h));
  CHECK(hipMemAddressFree(r.va, r.bytes));
  r.va = nullptr;
  return 0;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 4;
  const size_t filler_gb = argc > 2 ? (size_t)atoi(argv[2]) : 40;
  CHECK(hipSetDevice(0));
  memset(&g_prop, 0, sizeof(g_prop));
  g_prop.type = hipMemAllocationTypePinned; g_prop.location.type = hipMemLocationTypeDevice; g_prop.location.id = 0;
  size_t gran = 0; CHECK(hipMemGetAllocationGranularity(&gran, &g_prop, hipMemAllocationGranularityRecommended));
  printf("{\"granularity\": %zu}\n", gran);
  // what a real process holds before the derived layouts and the pool come: the table (plain allocations, as the library makes them)
  std::vector<void*> filler;
  for (size_t got = 0; got < filler_gb * 1000000000ull; got += 4000000256ull) { void* p = nullptr; CHECK(hipMalloc(&p, 4000000256ull)); filler.push_back(p); }
  unsigned long long* sink = nullptr; CHECK(hipMalloc((void**)&sink, 8)); CHECK(hipMemset(sink, 0, 8));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const size_t src_bytes = 8ull << 30, rec_bytes = 4ull << 30, pool_bytes = 1ull << 30;
  const uint64_t lines = pool_bytes / 128, nrec = rec_bytes / 8;
  const char* orders[] = {"seg_after", "seg_before", "interleave", "spaced"};
  for (size_t chunk : {(size_t)256 << 20, (size_t)2 << 20}) {
    if (chunk < gran) chunk = gran;
    for (int rep = 0; rep < reps; ++rep) {
      for (const char* order : orders) {
        Region src, rec, pool, spacer;
        if (region_reserve(src, src_bytes, chunk) || region_reserve(rec, rec_bytes, chunk) || region_reserve(pool, pool_bytes, chunk)) return 1;
        const std::string o = order;
        if (o == "seg_before") while (pool.mapped < pool.bytes) if (region_add_chunk(pool)) return 1;
        const size_t nsrc = src_bytes / chunk, npool = pool_bytes / chunk, every = nsrc / npool;
        for (size_t k = 0; k < nsrc; ++k) {
          if (region_add_chunk(src)) return 1;
          if (o == "interleave" && (k + 1) % every == 0) if (region_add_chunk(pool)) return 1;
        }
        while (rec.mapped < rec.bytes) if (region_add_chunk(rec)) return 1;
        if (o == "spaced") { if (region_reserve(spacer, 8ull << 30, (size_t)256 << 20)) return 1; while (spacer.mapped < spacer.bytes) if (region_add_chunk(spacer)) return 1; }
        while (pool.mapped < pool.bytes) if (region_add_chunk(pool)) return 1;
        if (region_finish(src) || region_finish(rec) || region_finish(pool)) return 1;
        CHECK(hipMemset(src.va, 1, src_bytes)); CHECK(hipMemset(rec.va, 2, rec_bytes)); CHECK(hipMemset(pool.va, 0, pool_bytes));
        float best[3] = {1e9f, 1e9f, 1e9f};
        const int modes[3] = {7, 5, 2};           // the scan's mix | the same without stores | stores alone
        for (int m = 0; m < 3; ++m)
          for (int r = 0; r < 4; ++r) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(place_kernel, dim3(256 * 8), dim3(256), 0, 0, (const u32x4*)src.va, src_bytes / 16, (u32x4*)pool.va, lines, modes[m], modes[m] == 2 ? 1 : 6, sink,
                               (const uint64_t*)rec.va, nrec);
            CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r && ms < best[m]) best[m] = ms;
          }
        printf("{\"chunk_mb\": %zu, \"rep\": %d, \"order\": \"%s\", \"full_ms\": %.3f, \"nostore_ms\": %.3f, \"store_only_ms\": %.3f, \"pool_va\": \"%p\"}\n", chunk >> 20, rep, order, best[0], best[1], best[2], pool.va);
        fflush(stdout);
        if (region_free(pool) || region_free(rec) || region_free(src) || region_free(spacer)) return 1;
      }
    }
  }
  // the same mix with everything from hipMalloc, several pools: the classes as the library meets them
  {
    u32x4* src = nullptr; uint64_t* rec = nullptr;
    CHECK(hipMalloc((void**)&src, src_bytes)); CHECK(hipMemset(src, 1, src_bytes));
    CHECK(hipMalloc((void**)&rec, rec_bytes)); CHECK(hipMemset(rec, 2, rec_bytes));
    std::vector<void*> pools;
    for (int k = 0; k < 10; ++k) {
      void* p = nullptr; CHECK(hipMalloc(&p, 3939416320ull)); pools.push_back(p);
      float best = 1e9f;
      for (int r = 0; r < 4; ++r) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(place_kernel, dim3(256 * 8), dim3(256), 0, 0, (const u32x4*)src, src_bytes / 16, (u32x4*)p, lines, 7, 6, sink, (const uint64_t*)rec, nrec);
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r && ms < best) best = ms;
      }
      printf("{\"hipmalloc_pool\": %d, \"full_ms\": %.3f, \"ptr\": \"%p\"}\n", k, best, p);
    }
  }
  return 0;
}

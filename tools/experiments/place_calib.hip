// Where a tuple pool lands: the same partitioned C3 query runs 2.05 or 2.35 ms depending on WHICH scratch buffer its
// execution context holds (tools/scratch_probe.py), reproducibly for as long as the buffer lives. This program asks what about a buffer
// decides that, without the library: a kernel that streams a 10 GB source (the scan) while every wave stores whole 128-byte lines at
// pseudo-random places of the first 1 GB of a candidate buffer (the tuple appends, 1 byte written per 6 read as on C3), timed per
// candidate, for candidates obtained three ways behind a 73 GB filler (the table + projection + narrow copies):
//   odd    hipMalloc(3 939 416 320)                          what the library's scratch asks for on C3
//   pow2   hipMalloc(4 GiB)                                  one buddy block if the driver has one
//   vmm    hipMemAddressReserve(4 GiB, aligned 4 GiB) + hipMemCreate + hipMemMap: virtual and physical side aligned alike
// build: hipcc --offload-arch=gfx950 -O3 -o tools/experiments/place_calib tools/experiments/place_calib.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; return x ^ (x >> 31); }

// mode bit 0: stream the source; bit 1: store lines. One store instruction of a wave = 8 whole lines (8 lanes x 16 B each).
__global__ __launch_bounds__(256) void place_kernel(const u32x4* __restrict__ src, uint64_t n16, u32x4* __restrict__ dst, uint64_t lines, int mode,
                                                    int reads_per_store, unsigned long long* sink, const uint64_t* __restrict__ rec, uint64_t nrec) {
  uint32_t acc = 0;
  const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x, nt = (uint64_t)gridDim.x * 256;
  const uint64_t wave = g >> 6;
  const uint32_t lane = threadIdx.x & 63;
  uint64_t it = 0;
  for (uint64_t i = g; i < n16; i += nt, ++it) {
    if (mode & 1) { const u32x4 v = __builtin_nontemporal_load(src + i); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if ((mode & 4) && lane < 10) acc ^= (uint32_t)rec[mix(i * 3 + 1) % nrec];      // bit 2: the survivors' record gathers (10 per KB streamed, 8 bytes each, anywhere in 8 GB)
    if ((mode & 2) && it % (uint64_t)reads_per_store == 0) {
      const uint64_t line = mix(wave * 0x9E3779B97F4A7C15ull + it * 8 + (lane >> 3)) % lines;
      u32x4 v; v.x = (uint32_t)i; v.y = lane; v.z = acc; v.w = 7;
      dst[line * 8 + (lane & 7)] = v;
    }
  }
  if (acc == 0x12345u) atomicAdd(sink, 1ull);
}

struct Cand { const char* how; void* p; size_t bytes; hipMemGenericAllocationHandle_t h; bool vmm; };

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 10;
  const size_t filler_gb = argc > 2 ? (size_t)atoi(argv[2]) : 73;
  CHECK(hipSetDevice(0));
  std::vector<void*> filler;
  for (size_t got = 0; got < filler_gb * 1000000000ull; got += 4000000256ull) { void* p = nullptr; CHECK(hipMalloc(&p, 4000000256ull)); filler.push_back(p); }
  const uint64_t src_bytes = 10ull << 30;
  u32x4* src = nullptr; CHECK(hipMalloc((void**)&src, src_bytes)); CHECK(hipMemset(src, 1, src_bytes));
  const uint64_t nrec = 1000000000ull; uint64_t* rec = nullptr; CHECK(hipMalloc((void**)&rec, nrec * 8 + 1536256)); CHECK(hipMemset(rec, 2, nrec * 8));
  unsigned long long* sink = nullptr; CHECK(hipMalloc((void**)&sink, 8)); CHECK(hipMemset(sink, 0, 8));
  const size_t odd = 3939416320ull, pow2 = 4ull << 30;
  std::vector<Cand> c;
  hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  size_t gran = 0; (void)hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
  for (int k = 0; k < n; ++k) {
    Cand a{"odd", nullptr, odd, {}, false}; CHECK(hipMalloc(&a.p, odd)); c.push_back(a);
    Cand b{"pow2", nullptr, pow2, {}, false}; CHECK(hipMalloc(&b.p, pow2)); c.push_back(b);
    Cand v{"vmm", nullptr, pow2, {}, true};
    hipError_t e = hipMemAddressReserve(&v.p, pow2, pow2, nullptr, 0);
    if (e == hipSuccess) e = hipMemCreate(&v.h, pow2, &prop, 0);
    if (e == hipSuccess) e = hipMemMap(v.p, pow2, 0, v.h, 0);
    hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    if (e == hipSuccess) e = hipMemSetAccess(v.p, pow2, &acc, 1);
    if (e == hipSuccess) c.push_back(v); else if (k == 0) printf("{\"vmm\": \"unavailable: %s\"}\n", hipGetErrorString(e));
  }
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const uint64_t lines = (1ull << 30) / 128;
  auto timed = [&](u32x4* dst, int mode, int rps, float* best) -> int {
    *best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      CHECK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(place_kernel, dim3(256 * 8), dim3(256), 0, 0, src, src_bytes / 16, dst, lines, mode, rps, sink, rec, nrec);
      CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
      float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (rep && ms < *best) *best = ms;
    }
    return 0;
  };
  float rd = 0; if (timed((u32x4*)c[0].p, 1, 6, &rd)) return 1;
  printf("{\"granularity\": %zu, \"read_only_ms\": %.3f}\n", gran, rd);
  for (size_t k = 0; k < c.size(); ++k) {
    float mixed = 0, wr = 0, full = 0, gonly = 0;
    if (timed((u32x4*)c[k].p, 3, 6, &mixed)) return 1;      // stream + one line store per 6 KB read and wave
    if (timed((u32x4*)c[k].p, 2, 1, &wr)) return 1;         // stores only, one per iteration (the source is not read)
    if (timed((u32x4*)c[k].p, 7, 6, &full)) return 1;       // stream + record gathers + line stores: the C3 scan's mix
    if (timed((u32x4*)c[k].p, 5, 6, &gonly)) return 1;      // stream + record gathers, no stores (the candidate is not touched)
    printf("{\"how\": \"%s\", \"k\": %zu, \"ptr\": \"%p\", \"mixed_ms\": %.3f, \"write_only_ms\": %.3f, \"full_ms\": %.3f, \"nostore_ms\": %.3f}\n", c[k].how, k / 3, c[k].p, mixed, wr, full, gonly);
    fflush(stdout);
  }
  return 0;
}

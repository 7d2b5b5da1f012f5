// How fast can gfx950 serve uncoalesced dword loads? (profiles/r01/NOTES.md: what bounds the C3 drain)
// Every lane loads ONE u32; patterns differ in how many distinct 128-byte lines a wave instruction touches.
//   build: hipcc --offload-arch=gfx950 -O3 -o gather_bw tools/experiments/gather_bw.hip ; run: ./gather_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// lane i of "global thread" g loads p[(g * stride_elems + jitter(g)) % n]; UNROLL independent loads in flight per lane
template <int UNROLL>
__global__ __launch_bounds__(256) void gather_kernel(const uint32_t* __restrict__ p, uint64_t n, uint64_t per_thread, uint32_t stride_elems,
                                                     uint32_t jitter_mask, unsigned long long* sink) {
  const uint64_t nthreads = (uint64_t)gridDim.x * 256;
  uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0;
  for (uint64_t it = 0; it < per_thread; it += UNROLL) {
    uint32_t v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint64_t k = g + (it + u) * nthreads;                         // consecutive lanes -> consecutive k
      const uint64_t h = k * 0x9E3779B97F4A7C15ull;
      const uint64_t idx = (k * stride_elems + ((h >> 40) & jitter_mask)) % n;
      v[u] = __builtin_nontemporal_load(p + idx);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
  }
  if (acc == 0x12345u) atomicAdd(sink, 1ull);
}

// non-returning 64-bit atomic adds into a table of `slots` u64 (pseudo-random slot per lane): the C3 table update in isolation
__global__ __launch_bounds__(256) void atomic_kernel(unsigned long long* table, uint64_t slots, uint64_t per_thread) {
  const uint64_t nthreads = (uint64_t)gridDim.x * 256;
  const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  for (uint64_t it = 0; it < per_thread; ++it) {
    uint64_t h = (g + it * nthreads) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    __hip_atomic_fetch_add(table + h % slots, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// 32-bit atomic adds and plain byte stores to pseudo-random slots, for comparison with the 64-bit atomics
template <int KIND>
__global__ __launch_bounds__(256) void update32_kernel(unsigned char* table, uint64_t slots, uint64_t per_thread) {
  const uint64_t nthreads = (uint64_t)gridDim.x * 256;
  const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  for (uint64_t it = 0; it < per_thread; ++it) {
    uint64_t h = (g + it * nthreads) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    const uint64_t i = h % slots;
    if (KIND == 0) __hip_atomic_fetch_add(reinterpret_cast<unsigned int*>(table) + i, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else table[i] = 1;
  }
}

// the same update with the cache-policy bits spelled out (for an atomic sc0 means "return the old value" and is refused on the
// non-returning form; sc1 and nt remain) — does any of them change where the update is executed?
template <int SC>
__global__ __launch_bounds__(256) void atomic_sc_kernel(unsigned long long* table, uint64_t slots, uint64_t per_thread) {
  const uint64_t nthreads = (uint64_t)gridDim.x * 256;
  const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const unsigned long long one = 1ull;
  for (uint64_t it = 0; it < per_thread; ++it) {
    uint64_t h = (g + it * nthreads) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    unsigned long long* addr = table + h % slots;
    if (SC == 0) asm volatile("global_atomic_add_x2 %0, %1, off" :: "v"(addr), "v"(one) : "memory");
    else if (SC == 1) asm volatile("global_atomic_add_x2 %0, %1, off nt" :: "v"(addr), "v"(one) : "memory");
    else if (SC == 2) asm volatile("global_atomic_add_x2 %0, %1, off sc1" :: "v"(addr), "v"(one) : "memory");
    else asm volatile("global_atomic_add_x2 %0, %1, off sc1 nt" :: "v"(addr), "v"(one) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int SC>
static float time_sc(unsigned long long* table, uint64_t slots, int grid, uint64_t per_thread, hipEvent_t a, hipEvent_t b) {
  atomic_sc_kernel<SC><<<grid, 256>>>(table, slots, per_thread);
  (void)hipEventRecord(a);
  for (int i = 0; i < 3; ++i) atomic_sc_kernel<SC><<<grid, 256>>>(table, slots, per_thread);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
  return ms / 3;
}

int main() {
  const uint64_t bytes = 16ull << 30, n = bytes / 4;
  uint32_t* buf; unsigned long long* sink;
  CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&sink, 8));
  CHECK(hipMemset(buf, 1, bytes)); CHECK(hipMemset(sink, 0, 8));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  struct Case { const char* name; uint32_t stride, jitter; };
  const Case cases[] = {{"coalesced dword (32 lanes per line)", 1, 0}, {"2 lanes per line", 16, 0}, {"1 lane per line, consecutive lines", 32, 0},
                        {"1 lane per line, every other line", 64, 0}, {"~5 % density (stride 20 +- jitter): C3-like", 20, 15},
                        {"1 lane per line, lines 4 KB apart", 1024, 0}};
  for (int bpc : {4, 7, 8}) {
    for (const Case& c : cases) {
      const uint64_t loads = 1ull << 30;                                  // lane loads per launch
      const int grid = cus * bpc;
      const uint64_t per_thread = loads / ((uint64_t)grid * 256) / 4 * 4;
      gather_kernel<4><<<grid, 256>>>(buf, n, per_thread, c.stride, c.jitter, sink);
      CHECK(hipEventRecord(a));
      for (int i = 0; i < 3; ++i) gather_kernel<4><<<grid, 256>>>(buf, n, per_thread, c.stride, c.jitter, sink);
      CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
      float ms; CHECK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
      const double nloads = (double)per_thread * grid * 256;
      const double lines = c.stride >= 32 ? nloads : nloads * c.stride / 32.0 * (c.jitter ? 1.0 : 1.0);
      printf("blocks/CU %d  %-52s %7.3f ms  %6.1f G lane-loads/s  ~%6.1f G lines/s  ~%5.2f TB/s of 128 B lines\n", bpc, c.name, ms,
             nloads / ms / 1e6, lines / ms / 1e6, lines * 128 / ms / 1e9);
    }
  }
  for (uint64_t slots : {100000ull, 4000000ull, 64000000ull}) {
    for (int bpc : {4, 8}) {
      const uint64_t n_at = 1ull << 28;
      const int grid = cus * bpc;
      const uint64_t per_thread = n_at / ((uint64_t)grid * 256);
      unsigned long long* table = reinterpret_cast<unsigned long long*>(buf);
      atomic_kernel<<<grid, 256>>>(table, slots, per_thread);
      CHECK(hipEventRecord(a));
      for (int i = 0; i < 3; ++i) atomic_kernel<<<grid, 256>>>(table, slots, per_thread);
      CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
      float ms; CHECK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
      printf("blocks/CU %d  atomic add u64, %9llu slots (%6.1f MB)  %7.3f ms  %6.1f G atomics/s\n", bpc, (unsigned long long)slots, slots * 8 / 1e6, ms,
             (double)per_thread * grid * 256 / ms / 1e6);
    }
  }
  for (uint64_t slots : {100000ull, 64000000ull}) {
    const uint64_t n_at = 1ull << 28;
    const int grid = cus * 8;
    const uint64_t per_thread = n_at / ((uint64_t)grid * 256);
    float ms32, ms8;
    update32_kernel<0><<<grid, 256>>>(reinterpret_cast<unsigned char*>(buf), slots, per_thread);
    CHECK(hipEventRecord(a));
    for (int i = 0; i < 3; ++i) update32_kernel<0><<<grid, 256>>>(reinterpret_cast<unsigned char*>(buf), slots, per_thread);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms32, a, b));
    update32_kernel<1><<<grid, 256>>>(reinterpret_cast<unsigned char*>(buf), slots, per_thread);
    CHECK(hipEventRecord(a));
    for (int i = 0; i < 3; ++i) update32_kernel<1><<<grid, 256>>>(reinterpret_cast<unsigned char*>(buf), slots, per_thread);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms8, a, b));
    const double n = (double)per_thread * grid * 256;
    printf("%9llu slots: u32 atomic add %6.1f G/s | plain byte store %6.1f G/s\n", (unsigned long long)slots, n / (ms32 / 3) / 1e6, n / (ms8 / 3) / 1e6);
  }
  for (uint64_t slots : {100000ull, 64000000ull}) {
    const uint64_t n_at = 1ull << 28;
    const int grid = cus * 8;
    const uint64_t per_thread = n_at / ((uint64_t)grid * 256);
    unsigned long long* table = reinterpret_cast<unsigned long long*>(buf);
    const float t0 = time_sc<0>(table, slots, grid, per_thread, a, b), t1 = time_sc<1>(table, slots, grid, per_thread, a, b),
                t2 = time_sc<2>(table, slots, grid, per_thread, a, b), t3 = time_sc<3>(table, slots, grid, per_thread, a, b);
    const double n = (double)per_thread * grid * 256;
    printf("asm atomics, %9llu slots: no bits %6.1f | nt %6.1f | sc1 %6.1f | sc1 nt %6.1f  G atomics/s\n", (unsigned long long)slots,
           n / t0 / 1e6, n / t1 / 1e6, n / t2 / 1e6, n / t3 / 1e6);
  }
  return 0;
}

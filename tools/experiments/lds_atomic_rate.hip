// How fast are LDS atomics at random addresses? part_agg_kernel (phase 2 of DENSE_PART) does two 64-bit LDS adds per tuple and runs at
// about one lane-update per clock and CU (profiles/r02/NOTES.md). This program times, per variant, 1024-thread blocks (one per CU, a
// 120 KB table) doing `iters` updates per lane at pseudo-random slots:
//   add64        __hip_atomic_fetch_add on a 64-bit slot, result unused (ds_add_u64)
//   add32        the same on a 32-bit slot (ds_add_u32)
//   add32rtn     32-bit, result used (ds_add_rtn_u32)
//   add64x2      two 64-bit adds per iteration to two tables (what phase 2 does per tuple)
//   add32carry   a 64-bit sum as 32-bit halves: returning add on the low half, the high half only on a carry or a non-zero high word
//   store64      plain 64-bit stores (the pipeline without the read-modify-write)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/experiments/lds_atomic_rate tools/experiments/lds_atomic_rate.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; return x ^ (x >> 16); }

template <int V>
__global__ __launch_bounds__(1024) void rate_kernel(int iters, uint32_t slots, unsigned long long* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  unsigned long long* t64 = reinterpret_cast<unsigned long long*>(lds);
  uint32_t* t32 = reinterpret_cast<uint32_t*>(lds);
  for (uint32_t i = threadIdx.x; i < slots * 2; i += 1024) t64[i] = 0;        // (two 64-bit tables, or one and spare)
  __syncthreads();
  uint32_t s = mix32(blockIdx.x * 1024 + threadIdx.x + 1);
  unsigned long long acc = 0;
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;            // (cheap on purpose: the loop must not be bound by its own arithmetic; slots is a power of two)
    const uint32_t g = (s >> 9) & (slots - 1u);
    const unsigned long long v = s & 0xFFFFFu;
    if (V == 0) __hip_atomic_fetch_add(t64 + g, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (V == 1) __hip_atomic_fetch_add(t32 + g, (uint32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (V == 2) acc += __hip_atomic_fetch_add(t32 + g, (uint32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (V == 3) { __hip_atomic_fetch_add(t64 + g, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                  __hip_atomic_fetch_add(t64 + slots + g, (1ull << 32) | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    if (V == 4) { const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
                  const uint32_t old = __hip_atomic_fetch_add(t32 + 2 * g, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                  const uint32_t up = hi + (old + lo < old ? 1u : 0u);
                  if (up) __hip_atomic_fetch_add(t32 + 2 * g + 1, up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    if (V == 5) t64[g] = v;
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < slots; i += 1024) acc += t64[i];
  if (acc == 0x123456789ull) atomicAdd(sink, acc);
}

template <int V> static int run(const char* name, int iters, uint32_t slots, unsigned long long* sink, int cus, double ghz) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const size_t lds = (size_t)slots * 16;      // (8192 slots: 128 KB)
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rate_kernel<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(rate_kernel<V>, dim3(cus), dim3(1024), lds, 0, iters, slots, sink);
    CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep && ms < best) best = ms;
  }
  const double updates_per_cu = 1024.0 * iters;
  printf("{\"variant\": \"%s\", \"slots\": %u, \"ms\": %.3f, \"lane_updates_per_clock_per_cu\": %.2f}\n", name, slots, best, updates_per_cu / (best * 1e-3 * ghz * 1e9));
  return 0;
}

int main() {
  CHECK(hipSetDevice(0));
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount; const double ghz = p.clockRate / 1e6;
  unsigned long long* sink = nullptr; CHECK(hipMalloc((void**)&sink, 8)); CHECK(hipMemset(sink, 0, 8));
  printf("{\"cus\": %d, \"clock_ghz\": %.3f}\n", cus, ghz);
  for (uint32_t slots : {8192u, 1024u}) {
    const int iters = 4096;
    if (run<0>("add64", iters, slots, sink, cus, ghz)) return 1;
    if (run<1>("add32", iters, slots, sink, cus, ghz)) return 1;
    if (run<2>("add32rtn", iters, slots, sink, cus, ghz)) return 1;
    if (run<3>("add64x2 (per tuple)", iters, slots, sink, cus, ghz)) return 1;
    if (run<4>("add32carry", iters, slots, sink, cus, ghz)) return 1;
    if (run<5>("store64", iters, slots, sink, cus, ghz)) return 1;
  }
  return 0;
}

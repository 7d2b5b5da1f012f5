// FETCH_SIZE / WRITE_SIZE calibration: rocprofv3's byte counters against KNOWN byte counts, in the access patterns
// the aggregate path uses. MI355X_MICROARCH.md says FETCH_SIZE reports half the bytes of a wide (16 B/lane) coalesced stream on gfx950
// and calls every other width, and WRITE_SIZE, uncalibrated. One kernel per pattern, each moving a byte count printed on stdout:
//   read4 / read8 / read16   coalesced streams of 4 / 8 / 16 B per lane (non-temporal), 8 GiB each
//   gather32                 one 32-byte record per lane at a pseudo-random 32-byte-aligned offset of a 16 GiB buffer: the survivors'
//                            payload gathers of C3 (useful bytes 32 per lane; lines touched 128 B per lane when no two lanes share one)
//   write16p / write128      16-byte stores to pseudo-random 16-byte slots (partial lines) / whole 128-byte lines by 8 adjacent lanes
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/experiments/fetch_calib tools/experiments/fetch_calib.hip
//   run:   tools/fetch_calib.sh (two rocprofv3 --pmc passes, then the table of counter / known bytes)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <typename V>
__global__ __launch_bounds__(256) void calib_read(const V* __restrict__ p, uint64_t n, unsigned long long* sink) {
  uint32_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    const V v = __builtin_nontemporal_load(p + i);
    if constexpr (sizeof(V) == 4) acc ^= v; else if constexpr (sizeof(V) == 8) acc ^= v.x ^ v.y; else acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345u) atomicAdd(sink, 1ull);
}
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; return x ^ (x >> 31); }
__global__ __launch_bounds__(256) void calib_gather32(const u32x4* __restrict__ p, uint64_t records, uint64_t per_thread, unsigned long long* sink) {
  uint32_t acc = 0;
  const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x, nt = (uint64_t)gridDim.x * 256;
  for (uint64_t it = 0; it < per_thread; ++it) {
    const uint64_t r = mix(g + it * nt + 1) % records;
    const u32x4 a = p[r * 2], b = p[r * 2 + 1];
    acc ^= a.x ^ a.w ^ b.y ^ b.z;
  }
  if (acc == 0x12345u) atomicAdd(sink, 1ull);
}
// In-order gathers, the way a scan's survivors read their payload: thread g looks at rows [g * span, (g + 1) * span) of an array of `esize`-byte
// elements and loads the ONE element whose hash says so (span = 2: every other row on average -> both 64-byte halves of every line are
// asked for, by separate requests; span = 20: one row in twenty -> most 64-byte halves are asked for once or not at all).
template <typename V>
__global__ __launch_bounds__(256) void calib_gather_inorder(const V* __restrict__ p, uint64_t rows, uint32_t span, unsigned long long* sink, unsigned long long* touched64) {
  uint32_t acc = 0;
  unsigned long long mine = 0;
  for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; (g + 1) * span <= rows; g += (uint64_t)gridDim.x * 256) {
    const uint64_t r = g * span + mix(g + 3) % span;
    const V v = p[r];
    if constexpr (sizeof(V) == 4) acc ^= v; else acc ^= v.x ^ v.y;
    ++mine;
  }
  if (acc == 0x12345u) atomicAdd(sink, 1ull);
  (void)touched64; (void)mine;
}
__global__ __launch_bounds__(256) void calib_write16(u32x4* __restrict__ p, uint64_t slots, uint64_t per_thread) {
  const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x, nt = (uint64_t)gridDim.x * 256;
  for (uint64_t it = 0; it < per_thread; ++it) {
    u32x4 v; v.x = (uint32_t)g; v.y = (uint32_t)it; v.z = 1; v.w = 2;
    p[mix(g + it * nt + 7) % slots] = v;
  }
}
__global__ __launch_bounds__(256) void calib_write128(u32x4* __restrict__ p, uint64_t lines, uint64_t per_thread) {   // 8 adjacent lanes = one line
  const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x, nt = (uint64_t)gridDim.x * 256;
  for (uint64_t it = 0; it < per_thread; ++it) {
    u32x4 v; v.x = (uint32_t)g; v.y = (uint32_t)it; v.z = 1; v.w = 2;
    p[(mix((g >> 3) + it * (nt >> 3) + 7) % lines) * 8 + (g & 7)] = v;
  }
}

int main() {
  const uint64_t bytes = 16ull << 30;
  char* buf = nullptr;
  unsigned long long* sink = nullptr;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMalloc(&sink, 8));
  CHECK(hipMemset(buf, 1, bytes));
  CHECK(hipMemset(sink, 0, 8));
  const int grid = 256 * 8;
  const uint64_t stream = 8ull << 30;
  hipLaunchKernelGGL(calib_read<uint32_t>, dim3(grid), dim3(256), 0, 0, (const uint32_t*)buf, stream / 4, sink);
  hipLaunchKernelGGL(calib_read<u32x2>, dim3(grid), dim3(256), 0, 0, (const u32x2*)buf, stream / 8, sink);
  hipLaunchKernelGGL(calib_read<u32x4>, dim3(grid), dim3(256), 0, 0, (const u32x4*)buf, stream / 16, sink);
  const uint64_t per = 128, lanes = (uint64_t)grid * 256 * per;
  hipLaunchKernelGGL(calib_gather32, dim3(grid), dim3(256), 0, 0, (const u32x4*)buf, bytes / 32, per, sink);
  hipLaunchKernelGGL(calib_write16, dim3(grid), dim3(256), 0, 0, (u32x4*)buf, bytes / 16, per);
  hipLaunchKernelGGL(calib_write128, dim3(grid), dim3(256), 0, 0, (u32x4*)buf, bytes / 128, per);
  // 4-byte elements, every other row of 8 GiB; 8-byte records, one row in twenty of 8 GiB
  hipLaunchKernelGGL(calib_gather_inorder<uint32_t>, dim3(grid), dim3(256), 0, 0, (const uint32_t*)buf, stream / 4, 2u, sink, (unsigned long long*)nullptr);
  hipLaunchKernelGGL(calib_gather_inorder<u32x2>, dim3(grid), dim3(256), 0, 0, (const u32x2*)buf, stream / 8, 20u, sink, (unsigned long long*)nullptr);
  CHECK(hipDeviceSynchronize());
  printf("{\"gather4_every2_array\": %llu, \"gather8_every20_array\": %llu, ", (unsigned long long)stream, (unsigned long long)stream);
  printf("\"read4\": %llu, \"read8\": %llu, \"read16\": %llu, \"gather32_useful\": %llu, \"gather32_lines\": %llu, \"write16p\": %llu, \"write128\": %llu}\n",
         (unsigned long long)stream, (unsigned long long)stream, (unsigned long long)stream, (unsigned long long)(lanes * 32), (unsigned long long)(lanes * 128),
         (unsigned long long)(lanes * 16), (unsigned long long)(lanes * 16));
  return 0;
}

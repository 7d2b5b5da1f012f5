// How fast do result columns reach pinned host memory? One big copy vs per-column vs chunks, by kind of pinned allocation; and a kernel
// that writes straight into mapped host memory. (C5 delivers 35 M groups x 16-20 bytes per query: the copy is most of the query.)
// build: hipcc --offload-arch=gfx950 -O3 -o d2h_bw d2h_bw.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int main() {
  const size_t bytes = (size_t)560 << 20;
  char* d = nullptr;
  CK(hipMalloc(&d, bytes));
  CK(hipMemset(d, 1, bytes));
  hipStream_t s[4];
  for (auto& x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
  struct Kind { const char* name; unsigned flags; } kinds[] = {{"default", hipHostMallocDefault}, {"coherent", hipHostMallocCoherent}, {"noncoherent", hipHostMallocNonCoherent},
                                                               {"mapped", hipHostMallocMapped}, {"numa+default", hipHostMallocNumaUser}};
  for (auto& k : kinds) {
    char* h = nullptr;
    if (hipHostMalloc((void**)&h, bytes, k.flags) != hipSuccess) { printf("%s: alloc failed\n", k.name); (void)hipGetLastError(); continue; }
    for (size_t i = 0; i < bytes; i += 4096) h[i] = 0;
    for (int chunks : {1, 4, 16, 64}) {
      for (int nstream : {1, 2}) {
        double best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
          CK(hipDeviceSynchronize());
          auto t0 = std::chrono::steady_clock::now();
          const size_t cb = bytes / chunks;
          for (int c = 0; c < chunks; ++c) CK(hipMemcpyAsync(h + c * cb, d + c * cb, cb, hipMemcpyDeviceToHost, s[c % nstream]));
          for (int q = 0; q < nstream; ++q) CK(hipStreamSynchronize(s[q]));
          best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
        printf("%-13s chunks %2d streams %d: %.2f ms  %.1f GB/s\n", k.name, chunks, nstream, best, bytes / best / 1e6);
      }
    }
    // a kernel storing into the mapped buffer (few blocks: PCIe, not the CUs, is the limit)
    void* hd = nullptr;
    if (hipHostGetDevicePointer(&hd, h, 0) == hipSuccess) {
      for (int grid : {16, 64, 256}) {
        double best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipDeviceSynchronize());
          auto t0 = std::chrono::steady_clock::now();
          hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, s[0], (const uint4*)d, (uint4*)hd, bytes / 16);
          CK(hipStreamSynchronize(s[0]));
          best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
        printf("%-13s kernel stores, %3d blocks: %.2f ms  %.1f GB/s\n", k.name, grid, best, bytes / best / 1e6);
      }
    }
    CK(hipHostFree(h));
  }
  return 0;
}

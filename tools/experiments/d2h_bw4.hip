// d2h_bw, fourth question: does the rate of a device-to-host copy depend on WHERE in device memory the source lies?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main(int argc, char** argv) {
  const size_t copy_b = (size_t)256 << 20, dev_b = (size_t)(argc > 1 ? atoi(argv[1]) : 40) << 30;
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  char* d = nullptr; char* h = nullptr;
  CK(hipMalloc(&d, dev_b));
  CK(hipMemset(d, 1, dev_b));
  CK(hipHostMalloc((void**)&h, copy_b, hipHostMallocCoherent));
  for (size_t off = 0; off + copy_b <= dev_b; off += (dev_b / 24) / 4096 * 4096) {
    double best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      CK(hipMemcpyAsync(h, d + off, copy_b, hipMemcpyDeviceToHost, s));
      CK(hipStreamSynchronize(s));
      best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    printf("offset %6.2f GB: %.2f ms  %.1f GB/s\n", off / 1073741824.0, best, copy_b / best / 1e6);
  }
  // the same bytes through a second pinned buffer, allocated later
  char* h2 = nullptr; CK(hipHostMalloc((void**)&h2, (size_t)3 << 30, hipHostMallocCoherent));
  for (size_t hoff : {(size_t)0, (size_t)1 << 30, (size_t)2 << 30}) {
    double best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      CK(hipMemcpyAsync(h2 + hoff, d, copy_b, hipMemcpyDeviceToHost, s));
      CK(hipStreamSynchronize(s));
      best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    printf("host buffer of 3 GB, offset %zu GB: %.2f ms  %.1f GB/s\n", hoff >> 30, best, copy_b / best / 1e6);
  }
  return 0;
}

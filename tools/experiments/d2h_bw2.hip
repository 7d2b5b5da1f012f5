// d2h_bw, second question: does the size of the pinned allocation (or of the device allocation the rows are read from) change the rate?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main(int argc, char** argv) {
  const size_t copy_b = (size_t)700 << 20;
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (size_t host_mb : {700, 2800, 5600}) for (size_t dev_gb : {1, 30}) for (int touch : {0, 1}) {
    char* d = nullptr; char* h = nullptr;
    CK(hipMalloc(&d, dev_gb << 30));
    CK(hipHostMalloc((void**)&h, host_mb << 20, hipHostMallocCoherent));
    if (touch) for (size_t i = 0; i < (host_mb << 20); i += 4096) h[i] = 1;
    const size_t doff = (dev_gb << 30) - copy_b - 4096;
    double best = 1e9, first = 0;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      for (int c = 0; c < 4; ++c) CK(hipMemcpyAsync(h + c * (copy_b / 4), d + doff + c * (copy_b / 4), copy_b / 4, hipMemcpyDeviceToHost, s));
      CK(hipStreamSynchronize(s));
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (rep == 0) first = ms;
      best = std::min(best, ms);
    }
    printf("host %4zu MB (touched %d) dev %2zu GB: first %.2f ms, best %.2f ms  %.1f GB/s\n", host_mb, touch, dev_gb, first, best, copy_b / best / 1e6);
    CK(hipHostFree(h)); CK(hipFree(d));
  }
  return 0;
}

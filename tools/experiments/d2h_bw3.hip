// d2h_bw, third question: which way of allocating the pinned buffer puts it next to the GPU when the calling thread lives on the OTHER socket?
// run under: taskset -c 64-127 ./d2h_bw3   (the GPU hangs off node 0: CPUs 0-63,128-191)
#include <hip/hip_runtime.h>
#include <sched.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static void to_node0(cpu_set_t* old) { sched_getaffinity(0, sizeof(*old), old); cpu_set_t s; CPU_ZERO(&s); for (int c = 0; c < 64; ++c) CPU_SET(c, &s); sched_setaffinity(0, sizeof(s), &s); }
int main() {
  const size_t bytes = (size_t)700 << 20;
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  char* d = nullptr; CK(hipMalloc(&d, bytes));
  for (int mode = 0; mode < 5; ++mode) for (int rep2 = 0; rep2 < 2; ++rep2) {
    char* h = nullptr; cpu_set_t old; bool reg = false;
    const char* what = "";
    if (mode == 0) { what = "plain hipHostMalloc (thread on the far socket)"; CK(hipHostMalloc((void**)&h, bytes, hipHostMallocCoherent)); }
    if (mode == 1) { what = "affinity to node 0 around hipHostMalloc"; to_node0(&old); CK(hipHostMalloc((void**)&h, bytes, hipHostMallocCoherent)); sched_setaffinity(0, sizeof(old), &old); }
    if (mode == 2) { what = "affinity + hipHostMallocNumaUser"; to_node0(&old); CK(hipHostMalloc((void**)&h, bytes, hipHostMallocCoherent | hipHostMallocNumaUser)); sched_setaffinity(0, sizeof(old), &old); }
    if (mode == 3) { what = "affinity + mmap + touch + hipHostRegister"; to_node0(&old); h = (char*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
                     for (size_t i = 0; i < bytes; i += 4096) h[i] = 1; CK(hipHostRegister(h, bytes, hipHostRegisterDefault)); sched_setaffinity(0, sizeof(old), &old); reg = true; }
    if (mode == 4) { what = "NumaUser, no affinity change (far socket)"; CK(hipHostMalloc((void**)&h, bytes, hipHostMallocCoherent | hipHostMallocNumaUser)); }
    double best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s));
      CK(hipStreamSynchronize(s));
      best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    printf("%-50s %.2f ms  %.1f GB/s\n", what, best, bytes / best / 1e6);
    if (reg) { CK(hipHostUnregister(h)); munmap(h, bytes); } else CK(hipHostFree(h));
  }
  return 0;
}

// Hashed partitioning of the hash path: the kernels behind the scan (vh_hpart.h) and their launch sequence.
#define VH_HPART_KERNELS
#include "vh_hpart.h"
#include "vh_launch.h"
#include <cstdlib>

// After the scan kernel has written the stream pool: level A, the slices of the last pool, level B. (The ranges' aggregation is compiled per
// plan shape next to the scan kernel: vh_jit_launch_hpagg.)
template <int U>
static void launch_hpart(const VhPlanDev& P, const VhHpArgs* d_args, int num_cu, bool level_a_done, hipStream_t s) {
  static bool once = false;
  const size_t sl = hp_scatter_lds_bytes();
  if (!once) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&hp_scatter_kernel<1024, U>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sl);
    once = true;
  }
  static const int grid_a = getenv("VH_HP_GRID_A") ? atoi(getenv("VH_HP_GRID_A")) : 0;       // measurement
  if (!level_a_done) hipLaunchKernelGGL((hp_scatter_kernel<1024, U>), dim3(grid_a > 0 ? grid_a : num_cu), dim3(1024), sl, s, d_args, 0, P.counters);      // (level_a_done: the scan kernel wrote pool a itself, vj_fan_add)
  hipLaunchKernelGGL((hp_count_kernel<256>), dim3(num_cu), dim3(256), 0, s, d_args);
  hipLaunchKernelGGL(hp_plan_kernel, dim3(1), dim3(HP_FAN), 0, s, d_args, P.counters);
  hipLaunchKernelGGL((hp_scatter_kernel<1024, U>), dim3(HP_FAN), dim3(1024), sl, s, d_args, 1, P.counters);
}
void vh_launch_hpart(const VhPlanDev& P, const VhHpArgs* d_args, int units, int num_cu, bool level_a_done, hipStream_t s) {
  if (units == 2) launch_hpart<2>(P, d_args, num_cu, level_a_done, s);
  else launch_hpart<1>(P, d_args, num_cu, level_a_done, s);
}

// Hashed partitioning of the hash path: the kernels behind the scan (vh_hpart.h) and their launch sequence.
#define VH_HPART_KERNELS
#include "vh_hpart.h"
#include "vh_launch.h"
#include <algorithm>
#include <cstdlib>

// After the scan kernel has written the stream pool: level A, the slices of the last pool, level B. (The ranges' aggregation is compiled per
// plan shape next to the scan kernel: vh_jit_launch_hpagg.)
template <int U>
static void launch_hpart(const VhPlanDev& P, const VhHpArgs* d_args, int num_cu, int scan_blocks, int nb, hipStream_t s) {
  static bool once = false;
  const size_t sl = hp_scatter_lds_bytes(), rl = VJ_FAN_LDS_BYTES(1024);
  if (!once) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&hp_scatter_kernel<1024, U>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sl);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&hp_ring_scatter_kernel<1024, U>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rl);
    once = true;
  }
  if (scan_blocks > 0) {      // the scan kernel wrote pool a itself, by position (vj_fan_add): level B by position too, without barriers
    hipLaunchKernelGGL((hp_count_kernel<256>), dim3(num_cu), dim3(256), 0, s, d_args);
    hipLaunchKernelGGL(hp_plan_kernel, dim3(1), dim3(HP_FAN), 0, s, d_args, P.counters, nb);
    hipLaunchKernelGGL((hp_ring_scatter_kernel<1024, U>), dim3(HP_FAN * nb), dim3(1024), rl, s, d_args, (uint32_t)scan_blocks, (uint32_t)nb, P.counters);
    return;
  }
  static const int grid_a = getenv("VH_HP_GRID_A") ? atoi(getenv("VH_HP_GRID_A")) : 0;       // measurement
  hipLaunchKernelGGL((hp_scatter_kernel<1024, U>), dim3(grid_a > 0 ? grid_a : num_cu), dim3(1024), sl, s, d_args, 0, P.counters);
  hipLaunchKernelGGL((hp_count_kernel<256>), dim3(num_cu), dim3(256), 0, s, d_args);
  hipLaunchKernelGGL(hp_plan_kernel, dim3(1), dim3(HP_FAN), 0, s, d_args, P.counters, 0);
  hipLaunchKernelGGL((hp_scatter_kernel<1024, U>), dim3(HP_FAN), dim3(1024), sl, s, d_args, 1, P.counters);
}
// scan_blocks: 0, or the grid of a scan kernel that wrote level A itself; ring_blocks: blocks of level B per partition then
void vh_launch_hpart(const VhPlanDev& P, const VhHpArgs* d_args, int units, int num_cu, int scan_blocks, int ring_blocks, hipStream_t s) {
  if (units == 2) launch_hpart<2>(P, d_args, num_cu, scan_blocks, ring_blocks, s);
  else launch_hpart<1>(P, d_args, num_cu, scan_blocks, ring_blocks, s);
}

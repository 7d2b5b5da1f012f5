// Hashed partitioning of the hash path: the kernels behind the scan (vh_hpart.h) and their launch sequence.
#define VH_HPART_KERNELS
#include "vh_hpart.h"
#include "vh_launch.h"
#include <algorithm>
#include <cstdlib>

// After the scan kernel has written level A: the slices of the last pool from the partitions' counts, then level B. (The ranges' aggregation is
// compiled per plan shape next to the scan kernel: vh_jit_launch_hpagg.)
template <int U>
static void launch_hpart(const VhPlanDev& P, const VhHpArgs* d_args, int num_cu, int scan_blocks, int nb, hipStream_t s) {
  static bool once = false;
  const size_t rl = VJ_FAN_LDS_BYTES(1024);
  if (!once) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&hp_ring_scatter_kernel<1024, U>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rl);
    once = true;
  }
  hipLaunchKernelGGL((hp_count_kernel<256>), dim3(num_cu), dim3(256), 0, s, d_args);
  hipLaunchKernelGGL(hp_plan_kernel, dim3(1), dim3(HP_FAN), 0, s, d_args, P.counters, nb);
  hipLaunchKernelGGL((hp_ring_scatter_kernel<1024, U>), dim3(HP_FAN * nb), dim3(1024), rl, s, d_args, (uint32_t)scan_blocks, (uint32_t)nb, P.counters);
}
// scan_blocks: the grid of the scan kernel, which wrote level A; ring_blocks: blocks of level B per partition
void vh_launch_hpart(const VhPlanDev& P, const VhHpArgs* d_args, int units, int num_cu, int scan_blocks, int ring_blocks, hipStream_t s) {
  if (units == 2) launch_hpart<2>(P, d_args, num_cu, scan_blocks, ring_blocks, s);
  else launch_hpart<1>(P, d_args, num_cu, scan_blocks, ring_blocks, s);
}

// The second pass over heavy level-A partitions, from the first pass's tuples (hp_heavy_tuples_kernel): P is the second pass's plan.
void vh_launch_heavy_tuples(const VhPlanDev& P, const VhHeavyTuples& A, int num_cu, hipStream_t s) {
  const dim3 grid((unsigned)num_cu * 8u), block(256);
  if (A.units == 2) hipLaunchKernelGGL((hp_heavy_tuples_kernel<2, false>), grid, block, 0, s, P, A);
  else if (A.pk) hipLaunchKernelGGL((hp_heavy_tuples_kernel<1, true>), grid, block, 0, s, P, A);
  else hipLaunchKernelGGL((hp_heavy_tuples_kernel<1, false>), grid, block, 0, s, P, A);
}

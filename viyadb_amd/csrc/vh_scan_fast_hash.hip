// Fast scan kernel instantiations: open-addressing hash table in HBM.
#include "vh_kernels.h"
#include "vh_launch.h"

template <int SCOPE>
static void launch_np(const VhPlanDev& P, int grid, size_t lds, hipStream_t s) {
  switch (P.npred) {
    case 0: case 1: hipLaunchKernelGGL((scan_agg_fast_kernel<VH_MODE_HASH, 256, SCOPE, 1>), dim3(grid), dim3(256), lds, s, P); break;
    case 2: hipLaunchKernelGGL((scan_agg_fast_kernel<VH_MODE_HASH, 256, SCOPE, 2>), dim3(grid), dim3(256), lds, s, P); break;
    case 3: hipLaunchKernelGGL((scan_agg_fast_kernel<VH_MODE_HASH, 256, SCOPE, 3>), dim3(grid), dim3(256), lds, s, P); break;
    default: hipLaunchKernelGGL((scan_agg_fast_kernel<VH_MODE_HASH, 256, SCOPE, 4>), dim3(grid), dim3(256), lds, s, P); break;
  }
}

void vh_launch_scan_fast_hash(const VhPlanDev& P, int grid, size_t lds, hipStream_t s) {
  launch_np<__HIP_MEMORY_SCOPE_AGENT>(P, grid, lds, s);
}

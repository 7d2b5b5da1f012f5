// "Lanes" scan kernel instantiations (no compaction): dense table in LDS (1024-thread blocks) and the hash path's
// LDS front table (256-thread blocks).
#include "vh_kernels.h"
#include "vh_launch.h"

template <int MODE, int BLOCK, int SCOPE>
static void launch_np(const VhPlanDev& P, int grid, size_t lds, hipStream_t s) {
  switch (P.npred) {
    case 0: case 1: hipLaunchKernelGGL((scan_agg_lanes_kernel<MODE, BLOCK, SCOPE, 1>), dim3(grid), dim3(BLOCK), lds, s, P); break;
    case 2: hipLaunchKernelGGL((scan_agg_lanes_kernel<MODE, BLOCK, SCOPE, 2>), dim3(grid), dim3(BLOCK), lds, s, P); break;
    case 3: hipLaunchKernelGGL((scan_agg_lanes_kernel<MODE, BLOCK, SCOPE, 3>), dim3(grid), dim3(BLOCK), lds, s, P); break;
    default: hipLaunchKernelGGL((scan_agg_lanes_kernel<MODE, BLOCK, SCOPE, 4>), dim3(grid), dim3(BLOCK), lds, s, P); break;
  }
}

void vh_launch_scan_lanes_lds(const VhPlanDev& P, int grid, size_t lds, bool xcd_private, hipStream_t s) {
  if (xcd_private) launch_np<VH_MODE_DENSE_LDS, 1024, __HIP_MEMORY_SCOPE_WORKGROUP>(P, grid, lds, s);
  else launch_np<VH_MODE_DENSE_LDS, 1024, __HIP_MEMORY_SCOPE_AGENT>(P, grid, lds, s);
}

void vh_launch_scan_lanes_hash(const VhPlanDev& P, int grid, size_t lds, hipStream_t s) {
  launch_np<VH_MODE_HASH, 256, __HIP_MEMORY_SCOPE_AGENT>(P, grid, lds, s);
}

void vh_launch_scan_lanes_part(const VhPlanDev& P, int grid, size_t lds, hipStream_t s) {
  launch_np<VH_MODE_DENSE_PART, 256, __HIP_MEMORY_SCOPE_AGENT>(P, grid, lds, s);
}

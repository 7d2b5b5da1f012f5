// Two-pass form of the compacting scan: pass 1 (predicate columns -> pass masks) and pass 2 (the compacting kernel
// over a dense HBM table, reading the masks instead of evaluating the filter).
#include "vh_kernels.h"
#include "vh_launch.h"

void vh_launch_scan_mask(const VhPlanDev& P, int grid, hipStream_t s, int* occ) {
  switch (P.npred) {
    case 0: case 1: VH_LAUNCH_OR_OCC((scan_mask_kernel<1>), 256, grid, 0, s, P, occ); break;
    case 2: VH_LAUNCH_OR_OCC((scan_mask_kernel<2>), 256, grid, 0, s, P, occ); break;
    case 3: VH_LAUNCH_OR_OCC((scan_mask_kernel<3>), 256, grid, 0, s, P, occ); break;
    default: VH_LAUNCH_OR_OCC((scan_mask_kernel<4>), 256, grid, 0, s, P, occ); break;
  }
}

void vh_launch_scan_premask_global(const VhPlanDev& P, int grid, size_t lds, bool xcd_private, hipStream_t s, int* occ) {
  if (xcd_private) VH_LAUNCH_OR_OCC((scan_agg_fast_kernel<VH_MODE_DENSE_GLOBAL, 256, __HIP_MEMORY_SCOPE_WORKGROUP, 1, true>), 256, grid, lds, s, P, occ);
  else VH_LAUNCH_OR_OCC((scan_agg_fast_kernel<VH_MODE_DENSE_GLOBAL, 256, __HIP_MEMORY_SCOPE_AGENT, 1, true>), 256, grid, lds, s, P, occ);
}

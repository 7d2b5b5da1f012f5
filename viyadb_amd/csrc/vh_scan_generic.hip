// Generic scan kernel instantiations (any element type, any number of predicate columns).
#include "vh_kernels.h"
#include "vh_launch.h"

template <int MODE, int BLOCK>
static void launch(const VhPlanDev& P, int grid, size_t lds, bool xcd_private, hipStream_t s) {
  if (xcd_private)
    hipLaunchKernelGGL((scan_agg_kernel<MODE, BLOCK, __HIP_MEMORY_SCOPE_WORKGROUP>), dim3(grid), dim3(BLOCK), lds, s, P);
  else
    hipLaunchKernelGGL((scan_agg_kernel<MODE, BLOCK, __HIP_MEMORY_SCOPE_AGENT>), dim3(grid), dim3(BLOCK), lds, s, P);
}

void vh_launch_scan_generic(int mode, const VhPlanDev& P, int grid, size_t lds, bool xcd_private, hipStream_t s) {
  if (mode == VH_MODE_DENSE_LDS) launch<VH_MODE_DENSE_LDS, 1024>(P, grid, lds, xcd_private, s);
  else if (mode == VH_MODE_DENSE_GLOBAL) launch<VH_MODE_DENSE_GLOBAL, 256>(P, grid, lds, xcd_private, s);
  else launch<VH_MODE_HASH, 256>(P, grid, lds, false, s);
}

// Fast scan kernel instantiations: open-addressing hash table in HBM.
#include "vh_kernels.h"
#include "vh_launch.h"

template <int SCOPE>
static void launch_np(const VhPlanDev& P, int grid, size_t lds, hipStream_t s, int* occ) {
  switch (P.npred) {
    case 0: case 1: VH_LAUNCH_OR_OCC((scan_agg_fast_kernel<VH_MODE_HASH, 256, SCOPE, 1>), 256, grid, lds, s, P, occ); break;
    case 2: VH_LAUNCH_OR_OCC((scan_agg_fast_kernel<VH_MODE_HASH, 256, SCOPE, 2>), 256, grid, lds, s, P, occ); break;
    case 3: VH_LAUNCH_OR_OCC((scan_agg_fast_kernel<VH_MODE_HASH, 256, SCOPE, 3>), 256, grid, lds, s, P, occ); break;
    default: VH_LAUNCH_OR_OCC((scan_agg_fast_kernel<VH_MODE_HASH, 256, SCOPE, 4>), 256, grid, lds, s, P, occ); break;
  }
}

void vh_launch_scan_fast_hash(const VhPlanDev& P, int grid, size_t lds, hipStream_t s, int* occ) {
  launch_np<__HIP_MEMORY_SCOPE_AGENT>(P, grid, lds, s, occ);
}

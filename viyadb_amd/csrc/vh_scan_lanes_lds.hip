// "Lanes" scan kernel instantiations (no compaction): dense table in LDS, the hash path's LDS front table, and
// phase 1 of the radix-partitioned aggregation. `occ` != nullptr: do not launch, report how many blocks of this
// instantiation fit one CU with `lds` bytes of dynamic LDS (the scan kernels assume a fully co-resident grid).
#include "vh_kernels.h"
#include "vh_launch.h"

template <int MODE, int BLOCK, int SCOPE, int NP>
static void launch_one(const VhPlanDev& P, int grid, size_t lds, hipStream_t s, int* occ) {
  auto k = scan_agg_lanes_kernel<MODE, BLOCK, SCOPE, NP>;
  if (occ) {
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(occ, k, BLOCK, lds) != hipSuccess) *occ = 0;
    return;
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(BLOCK), lds, s, P);
}

template <int MODE, int BLOCK, int SCOPE>
static void launch_np(const VhPlanDev& P, int grid, size_t lds, hipStream_t s, int* occ) {
  switch (P.npred) {
    case 0: case 1: launch_one<MODE, BLOCK, SCOPE, 1>(P, grid, lds, s, occ); break;
    case 2: launch_one<MODE, BLOCK, SCOPE, 2>(P, grid, lds, s, occ); break;
    case 3: launch_one<MODE, BLOCK, SCOPE, 3>(P, grid, lds, s, occ); break;
    default: launch_one<MODE, BLOCK, SCOPE, 4>(P, grid, lds, s, occ); break;
  }
}

template <int BLOCK>
static void launch_lds(const VhPlanDev& P, int grid, size_t lds, bool xcd_private, hipStream_t s, int* occ) {
  if (xcd_private) launch_np<VH_MODE_DENSE_LDS, BLOCK, __HIP_MEMORY_SCOPE_WORKGROUP>(P, grid, lds, s, occ);
  else launch_np<VH_MODE_DENSE_LDS, BLOCK, __HIP_MEMORY_SCOPE_AGENT>(P, grid, lds, s, occ);
}

void vh_launch_scan_lanes_lds(const VhPlanDev& P, int block, int grid, size_t lds, bool xcd_private, hipStream_t s, int* occ) {
  if (block == 256) launch_lds<256>(P, grid, lds, xcd_private, s, occ);
  else if (block == 512) launch_lds<512>(P, grid, lds, xcd_private, s, occ);
  else launch_lds<1024>(P, grid, lds, xcd_private, s, occ);
}

void vh_launch_scan_lanes_hash(const VhPlanDev& P, int grid, size_t lds, hipStream_t s, int* occ) {
  launch_np<VH_MODE_HASH, 256, __HIP_MEMORY_SCOPE_AGENT>(P, grid, lds, s, occ);
}

void vh_launch_scan_lanes_part(const VhPlanDev& P, int grid, size_t lds, hipStream_t s, int* occ) {
  launch_np<VH_MODE_DENSE_PART, 256, __HIP_MEMORY_SCOPE_AGENT>(P, grid, lds, s, occ);
}

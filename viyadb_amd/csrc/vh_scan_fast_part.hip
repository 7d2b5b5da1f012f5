// Fast scan kernel instantiations: radix-partitioned tuples (phase 1) + LDS aggregation (phase 2).
#include "vh_kernels.h"
#include "vh_launch.h"

void vh_launch_scan_fast_part(const VhPlanDev& P, int grid, size_t lds, hipStream_t s, int* occ) {
#define VH_SHAPE_CASES(SH)                                                                                                                         \
  switch (P.npred) {                                                                                                                             \
    case 0: case 1: VH_LAUNCH_OR_OCC((scan_agg_shape_kernel<VH_MODE_DENSE_PART, 256, __HIP_MEMORY_SCOPE_AGENT, 1, SH>), 256, grid, lds, s, P, occ); break; \
    case 2: VH_LAUNCH_OR_OCC((scan_agg_shape_kernel<VH_MODE_DENSE_PART, 256, __HIP_MEMORY_SCOPE_AGENT, 2, SH>), 256, grid, lds, s, P, occ); break;         \
    case 3: VH_LAUNCH_OR_OCC((scan_agg_shape_kernel<VH_MODE_DENSE_PART, 256, __HIP_MEMORY_SCOPE_AGENT, 3, SH>), 256, grid, lds, s, P, occ); break;         \
    default: VH_LAUNCH_OR_OCC((scan_agg_shape_kernel<VH_MODE_DENSE_PART, 256, __HIP_MEMORY_SCOPE_AGENT, 4, SH>), 256, grid, lds, s, P, occ); break;        \
  }
  if (P.shape == 1) { VH_SHAPE_CASES(1) return; }
  if (P.shape == 2) { VH_SHAPE_CASES(2) return; }
#undef VH_SHAPE_CASES
  switch (P.npred) {
    case 0: case 1: VH_LAUNCH_OR_OCC((scan_agg_fast_kernel<VH_MODE_DENSE_PART, 256, __HIP_MEMORY_SCOPE_AGENT, 1>), 256, grid, lds, s, P, occ); break;
    case 2: VH_LAUNCH_OR_OCC((scan_agg_fast_kernel<VH_MODE_DENSE_PART, 256, __HIP_MEMORY_SCOPE_AGENT, 2>), 256, grid, lds, s, P, occ); break;
    case 3: VH_LAUNCH_OR_OCC((scan_agg_fast_kernel<VH_MODE_DENSE_PART, 256, __HIP_MEMORY_SCOPE_AGENT, 3>), 256, grid, lds, s, P, occ); break;
    default: VH_LAUNCH_OR_OCC((scan_agg_fast_kernel<VH_MODE_DENSE_PART, 256, __HIP_MEMORY_SCOPE_AGENT, 4>), 256, grid, lds, s, P, occ); break;
  }
}

void vh_launch_part_agg(const VhPlanDev& P, int blocks_per_part, size_t lds, hipStream_t s) {
  if (lds > 64 * 1024)   // a workgroup may own up to 160 KB of LDS on gfx950, but dynamic LDS above 64 KB has to be asked for
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&part_agg_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((part_agg_kernel<1024>), dim3(P.nfine * blocks_per_part), dim3(1024), lds, s, P, blocks_per_part);
}

// Two levels: size the partitions' slices of pool 2 from what phase 1 wrote, then split every partition 64 ways.
void vh_launch_part_split(const VhPlanDev& P, int blocks_per_part, bool ring, hipStream_t s) {
  if (ring && (P.tw == 2 || (P.tw == 1 && P.gid_bits)) && P.ext_tuples2 == VH_SPLIT_TILE_TUPLES) {      // extents by position, no barriers (part_split_ring_kernel)
    hipLaunchKernelGGL((part_l2_count_kernel<256>), dim3(256), dim3(256), 0, s, P);
    hipLaunchKernelGGL((part_l2_plan_kernel<64>), dim3(1), dim3(64), 0, s, P, blocks_per_part, blocks_per_part);
    if (P.tw == 1 && P.tuple4) hipLaunchKernelGGL((part_split_ring_kernel<256, 4>), dim3(P.npart * blocks_per_part), dim3(256), VH_SPLIT_RING_LDS(256), s, P, blocks_per_part);
    else if (P.tw == 1) hipLaunchKernelGGL((part_split_ring_kernel<256, 8>), dim3(P.npart * blocks_per_part), dim3(256), VH_SPLIT_RING_LDS(256), s, P, blocks_per_part);
    else hipLaunchKernelGGL((part_split_ring_kernel<256, 16>), dim3(P.npart * blocks_per_part), dim3(256), VH_SPLIT_RING_LDS(256), s, P, blocks_per_part);
    return;
  }
  hipLaunchKernelGGL((part_l2_count_kernel<256>), dim3(256), dim3(256), 0, s, P);
  hipLaunchKernelGGL((part_l2_plan_kernel<64>), dim3(1), dim3(64), 0, s, P, blocks_per_part * 4, 0);
  hipLaunchKernelGGL((part_split_kernel<256>), dim3(P.npart * blocks_per_part), dim3(256), 0, s, P, blocks_per_part);
}

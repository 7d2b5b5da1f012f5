// vh_launch.h — launch entry points of the scan kernels; each lives in its own translation unit so
// the 30-odd template instantiations compile in parallel (viyadb_amd/build.py).
#pragma once
#include "vh_internal.h"

void vh_launch_scan_generic(int mode, const VhPlanDev& P, int grid, size_t lds, bool xcd_private, hipStream_t s);
void vh_launch_scan_fast_lds(const VhPlanDev& P, int grid, size_t lds, bool xcd_private, hipStream_t s);
// occ != nullptr: no launch; *occ = blocks of that instantiation that fit one CU with `lds` bytes of dynamic LDS
void vh_launch_scan_lanes_lds(const VhPlanDev& P, int block, int grid, size_t lds, bool xcd_private, hipStream_t s, int* occ = nullptr);
void vh_launch_scan_lanes_hash(const VhPlanDev& P, int grid, size_t lds, hipStream_t s, int* occ = nullptr);
void vh_launch_scan_lanes_part(const VhPlanDev& P, int grid, size_t lds, hipStream_t s, int* occ = nullptr);
void vh_launch_scan_fast_global(const VhPlanDev& P, int grid, size_t lds, bool xcd_private, hipStream_t s, int* occ = nullptr);
void vh_launch_scan_fast_hash(const VhPlanDev& P, int grid, size_t lds, hipStream_t s, int* occ = nullptr);
void vh_launch_scan_fast_part(const VhPlanDev& P, int grid, size_t lds, hipStream_t s, int* occ = nullptr);
void vh_launch_part_agg(const VhPlanDev& P, int blocks_per_part, size_t lds, hipStream_t s);
void vh_launch_part_split(const VhPlanDev& P, int blocks_per_part, bool ring, hipStream_t s);
struct VhHpArgs;
struct VhHeavyTuples;
void vh_launch_heavy_tuples(const VhPlanDev& P, const VhHeavyTuples& A, int num_cu, hipStream_t s);      // a second pass over heavy partitions, from the first pass's tuples (vh_hpart.h)
void vh_launch_hpart(const VhPlanDev& P, const VhHpArgs* d_args, int units, int num_cu, int scan_blocks, int ring_blocks, hipStream_t s);      // hashed partitioning: everything behind the scan (vh_hpart.h)

// Launch KERNEL (parenthesised template-id), or — occ != nullptr — only ask the runtime how many of its blocks fit one CU.
#define VH_LAUNCH_OR_OCC(KERNEL, BLOCK, grid, lds, s, P, occ)                                             \
  do {                                                                                                     \
    auto k_ = KERNEL;                                                                                      \
    if (occ) {                                                                                             \
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(occ, k_, BLOCK, lds) != hipSuccess) *(occ) = 0;     \
    } else {                                                                                               \
      hipLaunchKernelGGL(k_, dim3(grid), dim3(BLOCK), lds, s, P);                                          \
    }                                                                                                      \
  } while (0)

// EXPERIMENT (env VH_EXPERIMENT_C3=1), not a product path: the C3 query with everything the plan knows folded in by
// hand — three u32 comparisons ANDed, two u32 group columns on a dense table, SUM(int64) + COUNT(u32 with presence
// carrier) — to measure what per-query specialisation of the fused scan kernel would buy over the interpreting one
// (profiles/r01/NOTES.md). Same geometry, prefetch, compaction and atomics as scan_agg_fast_kernel.
#include "vh_kernels.h"
#include "vh_launch.h"

__global__ __launch_bounds__(256) void scan_c3_experiment_kernel(const VhPlanDev P) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  typedef VhScanCfg<256> C;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  uint32_t* q = reinterpret_cast<uint32_t*>(lds) + wave * C::kQueueCap;
  const uint64_t lanemask_lt = (1ull << lane) - 1ull;
  unsigned long long npassed = 0;
  const uint32_t spu = P.unit_rows / C::kStepRows;
  const uint32_t l0 = (uint32_t)P.lits[P.prog[0].lit()], l1 = (uint32_t)P.lits[P.prog[1].lit()], l2 = (uint32_t)P.lits[P.prog[2].lit()];
  const char* pc0 = P.colbase[P.pred_slot[0]]; const uint64_t ps0 = P.colstride[P.pred_slot[0]];
  const char* pc1 = P.colbase[P.pred_slot[1]]; const uint64_t ps1 = P.colstride[P.pred_slot[1]];
  const char* pc2 = P.colbase[P.pred_slot[2]]; const uint64_t ps2 = P.colstride[P.pred_slot[2]];
  const char* gc0 = P.colbase[P.g[0].slot()]; const uint64_t gs0 = P.colstride[P.g[0].slot()];
  const char* gc1 = P.colbase[P.g[1].slot()]; const uint64_t gs1 = P.colstride[P.g[1].slot()];
  const char* mc0 = P.colbase[P.m[0].slot()]; const uint64_t ms0 = P.colstride[P.m[0].slot()];
  const char* mc1 = P.colbase[P.m[1].slot()]; const uint64_t ms1 = P.colstride[P.m[1].slot()];
  const uint32_t lo0 = (uint32_t)P.g[0].lo, lo1 = (uint32_t)P.g[1].lo, ex0 = (uint32_t)P.g[0].extent, ex1 = (uint32_t)P.g[1].extent;
  const uint32_t st0 = (uint32_t)P.g[0].stride, st1 = (uint32_t)P.g[1].stride;
  unsigned long long* s0 = reinterpret_cast<unsigned long long*>(P.m[0].state);
  unsigned long long* s1 = reinterpret_cast<unsigned long long*>(P.m[1].state);

  uint32_t t = 0, seg = 0, unit_base = 0, wave_base = 0, seg_rows = 0;
  bool have;
  {
    const uint32_t unit = blockIdx.x;
    have = unit < P.total_units;
    if (have) {
      seg = unit / P.units_per_seg;
      unit_base = (unit - seg * P.units_per_seg) * P.unit_rows;
      seg_rows = P.seg_rows[seg];
      wave_base = unit_base + wave * VH_WAVE_STEP_ROWS;
    }
  }
  uint32_t v0[16], v1[16], v2[16];
  auto preload = [&](uint32_t sg, uint32_t row_l, uint32_t rows) {
    const uint32_t* c0 = reinterpret_cast<const uint32_t*>(pc0 + (uint64_t)sg * ps0);
    const uint32_t* c1 = reinterpret_cast<const uint32_t*>(pc1 + (uint64_t)sg * ps1);
    const uint32_t* c2 = reinterpret_cast<const uint32_t*>(pc2 + (uint64_t)sg * ps2);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t r = row_l + k * 256u;
      if (r < rows) { vh_load4<uint32_t>(c0 + r, &v0[k * 4]); vh_load4<uint32_t>(c1 + r, &v1[k * 4]); vh_load4<uint32_t>(c2 + r, &v2[k * 4]); }
      else { for (int j = 0; j < 4; ++j) v0[k * 4 + j] = v1[k * 4 + j] = v2[k * 4 + j] = 0; }
    }
  };
  if (have) preload(seg, wave_base + lane * 4, seg_rows);
  uint32_t cnt = 0;
  bool bad_any = false;
  auto consume = [&](uint32_t sg, uint32_t row, bool active) {
    if (!active) row = 0;
    const uint32_t a = reinterpret_cast<const uint32_t*>(gc0 + (uint64_t)sg * gs0)[row];
    const uint32_t b = reinterpret_cast<const uint32_t*>(gc1 + (uint64_t)sg * gs1)[row];
    const unsigned long long x = reinterpret_cast<const unsigned long long*>(mc0 + (uint64_t)sg * ms0)[row];
    const uint32_t c = reinterpret_cast<const uint32_t*>(mc1 + (uint64_t)sg * ms1)[row];
    const uint32_t d0 = a - lo0, d1 = b - lo1;
    const bool bad = d0 >= ex0 || d1 >= ex1;
    if (active && bad) bad_any = true;
    if (active && !bad) {
      const uint32_t gid = d0 * st0 + d1 * st1;
      __hip_atomic_fetch_add(s0 + gid, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(s1 + gid, (1ull << 32) | c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  while (have) {
    const uint32_t row_l = wave_base + lane * 4;
    uint32_t mask = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) mask |= (uint32_t)((v0[i] == l0) & (v1[i] < l1) & (v2[i] >= l2)) << i;
    if (row_l + 3 * 256u + 4u > seg_rows) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t r = row_l + k * 256u;
        const uint32_t n = r >= seg_rows ? 0u : (seg_rows - r >= 4u ? 4u : seg_rows - r);
        mask &= ~(((0xFu << n) & 0xFu) << (k * 4));
      }
    }
    npassed += __popc(mask);
    ++t;
    uint32_t nseg = seg, nunit_base = unit_base, nwave_base = 0, nseg_rows = seg_rows;
    bool nhave;
    {
      const uint32_t unit = blockIdx.x + (t / spu) * gridDim.x;
      nhave = unit < P.total_units;
      if (nhave) {
        nseg = unit / P.units_per_seg;
        nunit_base = (unit - nseg * P.units_per_seg) * P.unit_rows;
        nseg_rows = P.seg_rows[nseg];
        nwave_base = nunit_base + (t % spu) * C::kStepRows + wave * VH_WAVE_STEP_ROWS;
      }
    }
    if (nhave) preload(nseg, nwave_base + lane * 4, nseg_rows);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t mk = (mask >> (4 * k)) & 0xFu;
      if (__ballot(mk != 0) == 0) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool bit = (mk >> j) & 1u;
        const uint64_t bal = __ballot(bit);
        if (bit) q[cnt + __popcll(bal & lanemask_lt)] = row_l + k * 256u + j;
        cnt += __popcll(bal);
      }
      __builtin_amdgcn_wave_barrier();
      while (cnt >= 64) {
        cnt -= 64;
        consume(seg, q[cnt + lane], true);
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (cnt && (!nhave || nseg != seg)) {
      const bool act = lane < (int)cnt;
      consume(seg, act ? q[lane] : 0, act);
      __builtin_amdgcn_wave_barrier();
      cnt = 0;
    }
    have = nhave; seg = nseg; unit_base = nunit_base; wave_base = nwave_base; seg_rows = nseg_rows;
  }
  if (__ballot(bad_any)) { if (bad_any) atomicOr(P.counters + 2, VH_ERR_RANGE); }
  for (int off = 32; off > 0; off >>= 1) npassed += __shfl_down(npassed, off);
  if (lane == 0 && npassed) atomicAdd(P.counters + 0, npassed);
}

void vh_launch_scan_c3_experiment(const VhPlanDev& P, int grid, size_t lds, hipStream_t s) {
  hipLaunchKernelGGL(scan_c3_experiment_kernel, dim3(grid), dim3(256), lds, s, P);
}

// bl_locate.hip - the locate kernels (gfx950): where a sample record lies on the simulation's grid.
//
//   bl_locate_plain_kernel   one SAMPLE per lane, 4+ waves per SIMD: cuts, CKS->SKS, cell search on LDS tables, trilinear fractions
//   bl_locate_kernel         -> located sample (simulation_sampling.cpp:201-575). The plain kernel is the exact tier's common case (one
//                            grid or equal blocks merged, spherical Kerr-Schild, trilinear, no optional cut, no slow light); the general
//                            kernel covers meshes with refinement, inter-block interpolation, slow light, FMKS, Cartesian grids, cuts -
//                            in both tiers (the tolerant tier's angles with the exact ones as fallback: bl_sampling_fast.h).
// A translation unit of its own since round 6 (it was the first half of bl_shade.hip): the general kernel's three copies of the record
// loop made that file the slowest of the build.
#include <type_traits>

#include "bl_sampling_fast.h"


// ---- locate kernel (simulation mode): one sample record per lane. Coordinate conversion and the
// LDS table walks of the cell search; no grid reads (the coefficient kernel issues those, where they
// overlap its arithmetic instead of saturating the texture addresser here).
// kRefined: mesh with refinement; block and cell from the mesh's tables (RefinedTables, bl_sampling.h) - staged in LDS where they fit
// (up to BL_LOCATE_REFINED_LDS), searched in HBM otherwise; the record loop is compiled once for either and once more for the samples that
// wait for FindNearbyInds.
// kSlow: slow light; the time slice of every sample that passed the cuts is found first (:296-349).
// kTablesInHbm: the coordinate tables of a merged grid are too large for LDS and are searched where they lie (a
// compile-time choice: table pointers that may be either LDS or global become flat loads, each of which waits on both
// memory counters).
// (The instantiations for meshes with refinement may be launched with 1 024 lanes, one workgroup to a compute unit - the same four
// waves per SIMD - so that one copy of the tables serves sixteen waves and may take most of the unit's 160 KiB: bl_launch_locate.)
template <bool kRefined, bool kSlow, bool kSpinZero, bool kTablesInHbm = false>
__global__ void __launch_bounds__(kRefined ? 1024 : 256, kRefined ? 1 : 4) bl_locate_kernel(const BlShadeArgs P_at_entry) {
  const BlShadeArgs &P = kernel_arguments_in_place<BlShadeArgs>();   // (bl_kernel_util.h; P_at_entry is never read)
  (void)P_at_entry;
  const BlSpacetime st = P.st;
  extern __shared__ double lds_tables[];
  GridTables tab;
  // Refined meshes: the tables of the search - block boundaries, lattice, the distinct coordinate rows with their spacing, every
  // block's rows and next centre - in LDS where they fit (refined_lds_bytes), read through LDS-typed pointers; the mesh's scalars come
  // from the kernel arguments. (The search is five rounds of dependent reads: 15 - 18 ms per 1024^2 frame from LDS, 21 - 23 from HBM.)
  RefinedTables refined = refined_tables_in_hbm(P.grid);
  bool tables_in_lds = false;
  if (kRefined) {
    for (int a = 0; a < 3; a++) {
      tab.xf[a] = tab.xv[a] = nullptr;
      tab.bucket[a] = nullptr;
    }
    const BlGridDevice &g = P.grid;
    if (g.refined_lds_bytes > 0 && !P.general_locate) {   // (the measurement switch: the tables searched where they lie in HBM)
      tables_in_lds = true;
      stage_refined_tables(g, lds_tables, &refined);
      __syncthreads();
    }
  } else if (kTablesInHbm) {
    const BlGridDevice &g = P.grid;
    for (int a = 0; a < 3; a++) {
      tab.xf[a] = g.xf[a];
      tab.xv[a] = g.xv[a];
      tab.bucket[a] = g.bucket[a];
    }
  } else {
    const BlGridDevice &g = P.grid;
    double *dst = lds_tables;
    for (int a = 0; a < 3; a++) {
      tab.xf[a] = dst;
      for (int i = threadIdx.x; i <= g.n[a]; i += blockDim.x) dst[i] = g.xf[a][i];
      dst += g.n[a] + 1;
      tab.xv[a] = dst;
      for (int i = threadIdx.x; i < g.n[a]; i += blockDim.x) dst[i] = g.xv[a][i];
      dst += g.n[a];
    }
    unsigned short *bdst = reinterpret_cast<unsigned short *>(dst);
    for (int a = 0; a < 3; a++) {
      tab.bucket[a] = bdst;
      for (int i = threadIdx.x; i < g.n_bucket[a]; i += blockDim.x) bdst[i] = g.bucket[a][i];
      bdst += g.n_bucket[a];
    }
    __syncthreads();
  }
  const unsigned long long n_records = P.counters_in[BL_CNT_RECORDS];
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  unsigned long long gathers_local = 0ull;
  // tolerant tier (its located samples carry the tag where the exact tier's carry the azimuth): the angles by the tier's functions
  const double angle_band = P.tag_in_record ? P.fast_angle_band : 0.0;
  // One record: radius and cuts, the time slice, the search; true: left for later (may_defer, locate_sample_refined's defer_nearby)
  // (where: the main loop's instantiations know where the tables lie - kTableLds, kTableHbm - and have no FindNearbyInds in them: its
  // samples wait for the pass over the lists, kTableAnywhere, which has the one copy)
  auto locate_record = [&](auto where, unsigned long long at, double x1, double x2, double x3, uint32_t ray, bool may_defer) __attribute__((always_inline)) -> bool {
    constexpr int kWhere = decltype(where)::value;
    constexpr bool kNearby = !kRefined || kWhere == kTableAnywhere;
    if (ray == BL_DEAD_RAY) {
      // kSampleNone: the tolerant coefficient kernel requests corner cells from the tag alone
      if (P.tag_in_record) reinterpret_cast<double2 *>(P.located + at)[1] = make_double2(0.0, 0.0);
      else P.located_tag[at] = 0ull;
      return false;
    }
    double r2;
    const double r = bl_radial_coordinate2<kSpinZero>(st, x1, x2, x3, &r2);
    bool skip = r > P.cuts.camera_r;                                 // simulation_sampling.cpp:238-243
    if (!skip && P.cuts.any_optional) skip = optional_cuts(*P.cold, x1, x2, x3, r);
    LocatedSample loc;
    loc.f_i = loc.f_j = loc.f_k = loc.ph = 0.0;
    loc.cell = 0u;
    loc.status = kSampleCut;
    unsigned long long t_ind = 0ull;
    if (kSlow && !skip && !(P.plasma.fallback_nan && P.ray_flags[ray] != 0)) {   // NaN rays are not sampled (:211-216)
      double t_frac;
      t_ind = (unsigned long long)locate_time(P.slow, P.sample_t[at] + P.slow.snapshot_time, ray, &t_frac);
      P.slow.frac[at] = t_frac;
    }
    if (!skip)
      locate_sample_tolerant<kRefined, kSpinZero, kWhere, kNearby>(P, tab, st, x1, x2, x3, r, &loc, &gathers_local, P.anchors != nullptr ? P.anchors + at * 8 : nullptr,
                                                                   &refined, angle_band, may_defer);
    if (kRefined && loc.status == kSamplePending) return true;
    double2 *dst = reinterpret_cast<double2 *>(P.located + at);
    const unsigned long long tag = (t_ind << 40) | ((unsigned long long)loc.status << 32) | loc.cell;
    dst[0] = make_double2(loc.f_i, loc.f_j);
    if (P.tag_in_record) {   // tolerant tier: the tag rides in the azimuth's slot (the few samples the exact kernel re-does
      dst[1] = make_double2(loc.f_k, __longlong_as_double((long long)tag));   // recompute the azimuth): 32 bytes, one stream
    } else {
      dst[1] = make_double2(loc.f_k, loc.ph);
      P.located_tag[at] = tag;
    }
    return false;
  };
  // Inter-block interpolation: samples with an anchor beyond their own block wait in a list per wave (LDS) until there are 64 of them
  // (a wave's instructions reach LDS in order: what one lane has written the next instruction's lanes read)
  // (the lists lie behind the tables in the dynamic LDS: 1 KiB per wave, bl_launch_locate adds them to the launch's bytes)
  const bool collect = kRefined && P.grid.block_interp != 0 && P.plasma.simulation_interp != 0;
  unsigned long long *pending = reinterpret_cast<unsigned long long *>(lds_tables) + (kRefined ? (size_t)((P.general_locate ? 0 : P.grid.refined_lds_bytes) + 7) / 8 + (threadIdx.x >> 6) * 128 : 0);
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t n_pending = 0u;   // (wave-uniform: the loop below is left by the whole wave at once)
  auto locate_pending = [&](uint32_t first, uint32_t count) __attribute__((always_inline)) {
    __builtin_amdgcn_wave_barrier();
    if (lane < count) {
      const unsigned long long at = pending[first + lane];
      const double2 *src = reinterpret_cast<const double2 *>(P.records_hot + at * P.record_stride);
      const double2 q0 = src[0], q1 = src[1];
      locate_record(std::integral_constant<int, kTableAnywhere>{}, at, q0.x, q0.y, q1.x, (uint32_t)__double_as_longlong(q1.y), false);
    }
    __builtin_amdgcn_wave_barrier();
  };
  // Position and id of the next record are requested one iteration ahead
  unsigned long long idx = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool more = idx < n_records;
  double2 nq0 = make_double2(0.0, 0.0), nq1 = nq0;
  if (more) {
    const double2 *src = reinterpret_cast<const double2 *>(P.records_hot + (idx) * P.record_stride);
    nq0 = src[0];
    nq1 = src[1];
  }
  while (__any(more)) {
    const bool have = more;
    const unsigned long long at = idx;
    const double x1 = nq0.x, x2 = nq0.y, x3 = nq1.x;
    const uint32_t ray = (uint32_t)__double_as_longlong(nq1.y);
    idx += stride;
    more = more && idx < n_records;
    if (more) {
      const double2 *src = reinterpret_cast<const double2 *>(P.records_hot + (idx) * P.record_stride);
      nq0 = src[0];
      nq1 = src[1];
    }
    bool waits = false;
    if (have) {
      if (kRefined && tables_in_lds) waits = locate_record(std::integral_constant<int, kTableLds>{}, at, x1, x2, x3, ray, collect);
      else waits = locate_record(std::integral_constant<int, kTableHbm>{}, at, x1, x2, x3, ray, collect);
    }
    if (kRefined && collect) {
      const unsigned long long mask = __ballot(waits);
      if (mask != 0ull) {
        if (waits) pending[n_pending + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = at;
        n_pending += (uint32_t)__popcll(mask);
        if (n_pending >= 64u) {
          n_pending -= 64u;
          locate_pending(n_pending, 64u);
        }
      }
    }
  }
  if (kRefined && collect && n_pending > 0u) locate_pending(0u, n_pending);
  // S_in accounting: one atomic per wave
  for (int offset = 32; offset > 0; offset >>= 1) gathers_local += __shfl_xor(gathers_local, offset, 64);
  if ((threadIdx.x & 63) == 0 && gathers_local != 0ull) atomicAdd(&P.counters[BL_CNT_GATHERS], gathers_local);
}


// The locate kernel of that case
template <bool kSpinZero>
__global__ void __launch_bounds__(256, 4) bl_locate_plain_kernel(const BlShadeArgs P) {
  const BlSpacetime st = P.st;
  extern __shared__ double lds_tables[];
  PlainGrid pg;
  stage_grid_tables(P.grid, lds_tables, &pg);
  __syncthreads();
  const double camera_r = P.cuts.camera_r;
  const bool tag_in_record = P.tag_in_record != 0;
  const unsigned long long n_records = P.counters_in[BL_CNT_RECORDS];
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  unsigned long long gathers_local = 0ull;
  unsigned long long idx = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool more = idx < n_records;
  double2 nq0 = make_double2(1.0, 1.0), nq1 = make_double2(1.0, __longlong_as_double((long long)BL_DEAD_RAY));
  if (more) {
    const double2 *src = reinterpret_cast<const double2 *>(P.records_hot + (idx) * P.record_stride);
    nq0 = src[0];
    nq1 = src[1];
  }
  while (more) {
    const unsigned long long at = idx;
    const bool live = (uint32_t)__double_as_longlong(nq1.y) != BL_DEAD_RAY;
    const double x1 = nq0.x, x2 = nq0.y, x3 = nq1.x;
    idx += stride;
    more = idx < n_records;
    if (more) {
      const double2 *src = reinterpret_cast<const double2 *>(P.records_hot + (idx) * P.record_stride);
      nq0 = src[0];
      nq1 = src[1];
    }
    const PlainLocated loc = locate_plain_sample<kSpinZero>(st, P.grid, pg, camera_r, live, x1, x2, x3);
    gathers_local += loc.status == kSampleInterp ? 1ull : 0ull;
    const unsigned long long tag = ((unsigned long long)loc.status << 32) | loc.cell;
    double2 *dst = reinterpret_cast<double2 *>(P.located + at);
    if (tag_in_record) {
      // (a dead slot gets its tag alone in the general kernel; the fractions nobody reads are written here as zeros)
      dst[0] = make_double2(loc.f_i, loc.f_j);
      dst[1] = make_double2(loc.f_k, __longlong_as_double((long long)tag));
    } else if (live) {
      dst[0] = make_double2(loc.f_i, loc.f_j);
      dst[1] = make_double2(loc.f_k, loc.ph_unwrapped);
      P.located_tag[at] = tag;
    } else {
      P.located_tag[at] = 0ull;
    }
  }
  for (int offset = 32; offset > 0; offset >>= 1) gathers_local += __shfl_xor(gathers_local, offset, 64);
  if ((threadIdx.x & 63) == 0 && gathers_local != 0ull) atomicAdd(&P.counters[BL_CNT_GATHERS], gathers_local);
}


// =================================================================================================
// Launch wrapper (called from bl_render.hip)
// =================================================================================================
// Locate kernel (simulation mode only); lds_bytes = size of the coordinate tables it stages in LDS
extern "C" hipError_t bl_launch_locate(const BlShadeArgs *args, int grid, int lds_bytes, hipStream_t stream) {
  const bool refined = args->grid.n_blocks > 0, slow = args->slow.n > 0;
  const bool spin_zero = args->st.bh_a == 0.0;
  // the common case has a kernel of its own (bl_locate_plain_kernel): same located samples
  const bool plain = !refined && !slow && lds_bytes > 0 && !args->grid.fmks && args->plasma.simulation_interp && !args->cuts.any_optional
      && args->plasma.simulation_coord == BL_COORD_SKS && args->anchors == nullptr && !args->general_locate;
  if (plain) {
    if (spin_zero) hipLaunchKernelGGL((bl_locate_plain_kernel<true>), dim3(grid), dim3(256), lds_bytes, stream, *args);
    else hipLaunchKernelGGL((bl_locate_plain_kernel<false>), dim3(grid), dim3(256), lds_bytes, stream, *args);
    return hipGetLastError();
  }
  // (one instantiation for any spin - bit for bit the same at a = 0, bl_geometry.h "zero spin": the paths with zero spin known at
  // compile time are the common ones, bl_locate_plain_kernel above and the coefficient kernels with the locate step inside)
  if (refined) {
    // Tables that fit four times into a compute unit's LDS (36 KiB): 256-lane workgroups. Larger ones (up to BL_LOCATE_REFINED_LDS): one
    // 1 024-lane workgroup to a compute unit, one round of them. Beyond that the tables are searched where they lie in HBM
    // (refined_lds_bytes = 0). Behind the tables: the waves' lists of samples that wait for FindNearbyInds, 1 KiB each.
    const int table_bytes = args->general_locate ? 0 : args->grid.refined_lds_bytes;
    const bool four_to_a_unit = table_bytes <= 36 * 1024;
    const int lanes = four_to_a_unit ? 256 : 1024;
    const size_t bytes = (size_t)(table_bytes + 7) / 8 * 8 + (size_t)(lanes / 64) * 1024;
    // (`grid` counts 256-lane workgroups: sixteen to a compute unit when the kernel runs alone - then one large workgroup per unit - or one
    // to a unit beside the next chunk's stepper - then as many lanes as that)
    const dim3 blocks(four_to_a_unit ? grid : (grid >= 1024 ? grid / 16 : (grid >= 4 ? grid / 4 : 1)));
    const void *kernel = slow ? reinterpret_cast<const void *>(&bl_locate_kernel<true, true, false>) : reinterpret_cast<const void *>(&bl_locate_kernel<true, false, false>);
    if (bytes > 64 * 1024) {
      const hipError_t err = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BL_LOCATE_REFINED_LDS + 16 * 1024);
      if (err != hipSuccess) return err;
    }
    if (slow) hipLaunchKernelGGL((bl_locate_kernel<true, true, false>), blocks, dim3(lanes), bytes, stream, *args);
    else hipLaunchKernelGGL((bl_locate_kernel<true, false, false>), blocks, dim3(lanes), bytes, stream, *args);
  }
  else if (lds_bytes == 0)   // merged grid with tables beyond the LDS budget (not with slow light: its instantiation needs them in LDS)
    hipLaunchKernelGGL((bl_locate_kernel<false, false, false, true>), dim3(grid), dim3(256), 0, stream, *args);
  else if (slow) hipLaunchKernelGGL((bl_locate_kernel<false, true, false>), dim3(grid), dim3(256), lds_bytes, stream, *args);
  else hipLaunchKernelGGL((bl_locate_kernel<false, false, false>), dim3(grid), dim3(256), lds_bytes, stream, *args);
  return hipGetLastError();
}

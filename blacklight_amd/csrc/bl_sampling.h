// bl_sampling.h - device functions of the exact arithmetic tier between a sample record and its transfer record: cell search,
// trilinear read, frames, coefficients (simulation_sampling.cpp, simulation_coefficients.cpp, formula_coefficients.cpp) - shared
// by bl_shade.hip, bl_shade_fast.hip (whose second pass and discrete decisions are the exact tier's) and
// bl_coefficients_freq.hip.
#pragma once
#include "bl_kernel_util.h"

namespace {

// Coordinate tables of the (single-block) grid staged in LDS: faces and centres of the three axes
// plus one bucket table per axis. The reference finds a cell by a linear scan over faces
// (simulation_sampling.cpp:458-466: first c with xf[c+1] >= x); the bucket table gives a start
// index that is never beyond that result, so a short forward scan lands on the same cell.
struct GridTables {
  const double *xf[3];
  const double *xv[3];
  const unsigned short *bucket[3];
};

__device__ __forceinline__ int find_cell(const BlGridDevice &g, const GridTables &t, int axis, double x) {
  const int n = g.n[axis];
  double u = (x - g.bucket_x0[axis]) * g.bucket_inv_w[axis];
  int b = (int)u;
  b = b < 0 ? 0 : (b >= g.n_bucket[axis] ? g.n_bucket[axis] - 1 : b);
  int c = t.bucket[axis][b];
  const double *xf = t.xf[axis];
  // straight-line common case (the answer is c or c + 1), then the general forward scan
  double f1 = xf[c + 1];
  double f2 = xf[(c + 2 <= n) ? c + 2 : n];
  if (!(f1 >= x)) {
    c += 1;
    if (!(f2 >= x)) {
      c += 1;
      while (c < n - 1 && !(xf[c + 1] >= x)) c++;
    }
  }
  if (c > n - 1) c = n - 1;
  return c;
}

__device__ __forceinline__ void load_cell(const BlGridDevice &g, int k, int j, int i, float v[8]) {
  size_t idx = (((size_t)k * g.n[1] + j) * g.n[0] + i) * 8;
  const float4 *p = reinterpret_cast<const float4 *>(g.cells + idx);
  float4 a = p[0], b = p[1];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// radiation_geometry.cpp:597-658
__device__ __forceinline__ void tetrad_build(const double ucon[4], const double ucov[4], const double kcon[4],
                                             const double kcov[4], const double up_con[4],
                                             const double gcov[4][4], const double gcon[4][4],
                                             double tetrad[4][4]) {
  double omega = 0.0;
  for (int mu = 0; mu < 4; mu++) omega -= kcov[mu] * ucon[mu];
  double k_up_over_omega = 0.0;
  for (int mu = 0; mu < 4; mu++) k_up_over_omega += kcov[mu] * up_con[mu];
  const BlRecip rc_omega = bl_recip(omega);   // six quotients over omega
  k_up_over_omega = bl_div_r(k_up_over_omega, rc_omega);
  double u_up_over_omega = 0.0;
  for (int mu = 0; mu < 4; mu++) u_up_over_omega += ucov[mu] * up_con[mu];
  u_up_over_omega = bl_div_r(u_up_over_omega, rc_omega);
  for (int mu = 0; mu < 4; mu++) tetrad[0][mu] = ucon[mu];
  for (int mu = 0; mu < 4; mu++) tetrad[3][mu] = bl_div_r(kcon[mu], rc_omega) - ucon[mu];
  for (int mu = 0; mu < 4; mu++)
    tetrad[2][mu] = up_con[mu] - k_up_over_omega * tetrad[3][mu] + u_up_over_omega * kcon[mu];
  double norm = 0.0;
  for (int mu = 0; mu < 4; mu++)
    for (int nu = 0; nu < 4; nu++) norm += gcov[mu][nu] * tetrad[2][mu] * tetrad[2][nu];
  norm = bl_sqrt_g(norm);
  const BlRecip rc_norm = bl_recip(norm);
  for (int mu = 0; mu < 4; mu++) tetrad[2][mu] = bl_div_r(tetrad[2][mu], rc_norm);
  double t1[4];
  t1[0] = tetrad[0][1] * (tetrad[2][3] * tetrad[3][2] - tetrad[2][2] * tetrad[3][3])
      + tetrad[0][2] * (tetrad[2][1] * tetrad[3][3] - tetrad[2][3] * tetrad[3][1])
      + tetrad[0][3] * (tetrad[2][2] * tetrad[3][1] - tetrad[2][1] * tetrad[3][2]);
  t1[1] = tetrad[0][0] * (tetrad[2][2] * tetrad[3][3] - tetrad[2][3] * tetrad[3][2])
      + tetrad[0][2] * (tetrad[2][3] * tetrad[3][0] - tetrad[2][0] * tetrad[3][3])
      + tetrad[0][3] * (tetrad[2][0] * tetrad[3][2] - tetrad[2][2] * tetrad[3][0]);
  t1[2] = tetrad[0][0] * (tetrad[2][3] * tetrad[3][1] - tetrad[2][1] * tetrad[3][3])
      + tetrad[0][1] * (tetrad[2][0] * tetrad[3][3] - tetrad[2][3] * tetrad[3][0])
      + tetrad[0][3] * (tetrad[2][1] * tetrad[3][0] - tetrad[2][0] * tetrad[3][1]);
  t1[3] = tetrad[0][0] * (tetrad[2][1] * tetrad[3][2] - tetrad[2][2] * tetrad[3][1])
      + tetrad[0][1] * (tetrad[2][2] * tetrad[3][0] - tetrad[2][0] * tetrad[3][2])
      + tetrad[0][2] * (tetrad[2][0] * tetrad[3][1] - tetrad[2][1] * tetrad[3][0]);
  for (int mu = 0; mu < 4; mu++) {
    double acc = 0.0;
    for (int nu = 0; nu < 4; nu++) acc += gcon[mu][nu] * t1[nu];
    tetrad[1][mu] = acc;
  }
}

// Optional geometric cuts (simulation_sampling.cpp:246-292, formula_coefficients.cpp:78-116); the
// unconditional r > camera_r cut (:238-243) is applied by the caller.
__device__ __forceinline__ bool optional_cuts(const BlShadeCold &c, double x1, double x2, double x3, double r) {
  if (c.omit_near || c.omit_far) {
    double dot_product = x1 * c.cam_x[1] + x2 * c.cam_x[2] + x3 * c.cam_x[3];
    if ((c.omit_near && dot_product > 0.0) || (c.omit_far && dot_product < 0.0)) return true;
  }
  if ((c.omit_in >= 0.0 && r < c.omit_in) || (c.omit_out >= 0.0 && r > c.omit_out)) return true;
  if (c.midplane_theta > 0.0 || c.midplane_theta < 0.0) {
    double th = bl_acos(x3 / r);
    if ((c.midplane_theta > 0.0 && blm_abs(th - kPi / 2.0) > c.midplane_theta)
        || (c.midplane_theta < 0.0 && blm_abs(th - kPi / 2.0) < -c.midplane_theta))
      return true;
  }
  if ((c.midplane_z > 0.0 && blm_abs(x3) > c.midplane_z) || (c.midplane_z < 0.0 && blm_abs(x3) < -c.midplane_z))
    return true;
  if (c.plane) {
    double dot_product = (x1 - c.plane_origin[0]) * c.plane_normal[0] + (x2 - c.plane_origin[1]) * c.plane_normal[1]
        + (x3 - c.plane_origin[2]) * c.plane_normal[2];
    if (dot_product < 0.0) return true;
  }
  return false;
}

// (a, b) of the affine update for one frequency (unpolarized.cpp:92-110); delta_lambda_cgs given
__device__ __forceinline__ double2 transfer_record(double j, double alpha, double delta_lambda_cgs) {
  double2 rec;
  if (alpha > 0.0) {
    double ss = j / alpha;
    double delta_tau = alpha * delta_lambda_cgs;
    if (delta_tau <= kDeltaTauMax) {
      rec.x = bl_exp(-delta_tau);
      rec.y = ss * bl_expm1(delta_tau);
    } else {
      rec.x = BL_THICK_MARK;
      rec.y = ss;
    }
  } else {
    rec.x = 1.0;
    rec.y = j * delta_lambda_cgs;
  }
  return rec;
}

// Everything the per-frequency loop needs from the per-sample (frequency independent) work
struct SampleShade {
  bool have_coefficients;      // false: j = alpha = 0 at every frequency
  double nu_fluid_over_nu;     // -k_mu u^mu (fluid-frame frequency per unit camera frequency*factor)
  double n_e_cgs, nu_c_cgs, theta_e, sin_theta_b, kb_tt_e_cgs;   // simulation
  double cos_theta_b, sin2_theta_b, cos2_theta_b, cos_sign;       // polarized coefficients only
  double theta_e_096, kk_0, kk_1, kk_2;   // polarized, thermal: theta_e^0.96 and K_0,1,2(1 / theta_e) - the same at every frequency
  double n_n0_fluid, fu[4];                                       // formula
  bool have_cell;              // cell_values recorded (simulation_coefficients.cpp:377-387)
  double cell[BL_NUM_CELL_VALUES];
};

// Status of a located sample (BlLocated::status)
enum SampleStatus { kSampleNone = 0, kSampleCut = 1, kSampleOffGrid = 2, kSampleNearest = 3, kSampleInterp = 4,
                    kSampleFormula = 5, kSampleAdvanced = 6,
                    kSamplePending = 7 };   // (never stored: locate_sample_refined's answer to defer_nearby, bl_locate_kernel's note to itself)

__device__ __forceinline__ void unpack_cell(const float4 &lo, const float4 &hi, float v[8]) {
  v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w;
  v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
}

__device__ __forceinline__ const float4 *cell_ptr(const BlGridDevice &g, int k, int j, int i) {
  size_t idx = (((size_t)k * g.n[1] + j) * g.n[0] + i) * 8;
  return reinterpret_cast<const float4 *>(g.cells + idx);
}

// First index i in [0, n) with table[i + 1] >= x, for x in [table[0], table[n]] (the rule of the reference's
// linear scans, simulation_sampling.cpp:458-466), by bisection.
__device__ __forceinline__ int first_upper_face(const double *table, int n, double x) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (table[mid + 1] >= x) hi = mid; else lo = mid + 1;
  }
  return lo;
}

// ---- inter-block interpolation (simulation_block_interp = true)

// The tables of a mesh with refinement as the search reads them: where they lie in HBM (BlGridDevice's pointers), or the locate
// kernel's copies of them in LDS (read by ds_read through LDS-typed pointers - a pointer that may be either becomes a flat load,
// which waits on both memory counters and goes through the texture addresser). The mesh's scalars come from BlShadeArgs::grid itself.
struct RefinedTables {
  const double *edge[3], *bxf[3], *bxv[3], *xv_next[3], *row_guess[3];
  const int *lattice, *block_row[3];
  // inter-block interpolation: the MeshBlock table and the hash from (level, location) to block
  const int *levels, *locations, *hash_blocks;
  const unsigned long long *hash_keys;
  bool in_lds;   // (wave-uniform) what kTableAnywhere reads go by
};
// where a search function's tables lie: known when it is compiled, or a wave-uniform branch per read (one copy of FindNearbyInds for both)
enum { kTableHbm = 0, kTableLds = 1, kTableAnywhere = 2 };
__device__ __forceinline__ RefinedTables refined_tables_in_hbm(const BlGridDevice &g) {
  RefinedTables t;
  for (int a = 0; a < 3; a++) {
    t.edge[a] = g.edge[a];
    t.bxf[a] = g.bxf[a];
    t.bxv[a] = g.bxv[a];
    t.xv_next[a] = g.xv_next[a];
    t.row_guess[a] = g.row_guess[a];
    t.block_row[a] = g.block_row[a];
  }
  t.lattice = g.lattice;
  t.levels = g.levels;
  t.locations = g.locations;
  t.hash_blocks = g.hash_blocks;
  t.hash_keys = g.hash_keys;
  t.in_lds = false;
  return t;
}
// A workgroup's copy of the mesh's tables in LDS (`lds`: BlGridDevice::refined_lds_bytes of them), and *t pointing at it: every lane of
// the workgroup calls this and passes a barrier before the first search (bl_locate_kernel; the exact second pass behind the fused kernel)
__device__ __forceinline__ void stage_refined_tables(const BlGridDevice &g, double *lds, RefinedTables *t) {
  double *dd = lds;
  auto stage_doubles = [&](const double *src, int count) {
    double *at = dd;
    for (int i = threadIdx.x; i < count; i += blockDim.x) at[i] = src[i];
    dd += count;
    return at;
  };
#pragma unroll
  for (int a = 0; a < 3; a++) {
    t->edge[a] = stage_doubles(g.edge[a], g.n_edge[a] + 1);
    t->bxf[a] = stage_doubles(g.bxf[a], g.n_rows[a] * (g.nb[a] + 1));
    t->bxv[a] = stage_doubles(g.bxv[a], g.n_rows[a] * g.nb[a]);
    t->xv_next[a] = stage_doubles(g.xv_next[a], g.n_blocks);
    t->row_guess[a] = stage_doubles(g.row_guess[a], 3 * g.n_rows[a]);
  }
  if (g.block_interp) {   // (the hash's keys: eight bytes each, with the doubles)
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(dd);
    for (int i = threadIdx.x; i <= (int)g.hash_mask; i += blockDim.x) keys[i] = g.hash_keys[i];
    t->hash_keys = keys;
    dd += g.hash_mask + 1;
  }
  int *ii = reinterpret_cast<int *>(dd);
  auto stage_ints = [&](const int *src, int count) {
    int *at = ii;
    for (int i = threadIdx.x; i < count; i += blockDim.x) at[i] = src[i];
    ii += count;
    return at;
  };
  t->lattice = stage_ints(g.lattice, g.n_edge[0] * g.n_edge[1] * g.n_edge[2]);
#pragma unroll
  for (int a = 0; a < 3; a++) t->block_row[a] = stage_ints(g.block_row[a], g.n_blocks);
  if (g.block_interp) {
    t->levels = stage_ints(g.levels, g.n_blocks);
    t->locations = stage_ints(g.locations, 3 * g.n_blocks);
    t->hash_blocks = stage_ints(g.hash_blocks, (int)g.hash_mask + 1);
  }
  t->in_lds = true;
}

template <int kWhere, typename T>
__device__ __forceinline__ T table_read(const RefinedTables &t, const T *table, size_t i) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (kWhere == kTableLds || (kWhere == kTableAnywhere && t.in_lds))
    return (reinterpret_cast<const __attribute__((address_space(3))) T *>((uint32_t)(size_t)table))[i];
#endif
  return table[i];
}

// Block of a refinement level at a logical location, -1: none. The reference finds it by scanning all blocks
// (simulation_sampling.cpp:1246-1256 and alike); blocks do not overlap, so the key is unique.
__device__ __forceinline__ unsigned long long block_key(int level, int li, int lj, int lk) {
  return ((unsigned long long)(unsigned)level << 57) | ((unsigned long long)(unsigned)li << 38) | ((unsigned long long)(unsigned)lj << 19)
      | (unsigned long long)(unsigned)lk;
}
template <int kWhere>
__device__ __forceinline__ int find_block(const BlGridDevice &g, const RefinedTables &t, int level, int li, int lj, int lk) {
  if (level < 0 || level > g.max_level || li < 0 || lj < 0 || lk < 0 || li >= (1 << 19) || lj >= (1 << 19) || lk >= (1 << 19)) return -1;
  const unsigned long long key = block_key(level, li, lj, lk);
  unsigned int slot = (unsigned int)((key * 0x9e3779b97f4a7c15ull) >> 32) & g.hash_mask;
  while (true) {
    const unsigned long long found = table_read<kWhere>(t, t.hash_keys, slot);
    if (found == key) return table_read<kWhere>(t, t.hash_blocks, slot);
    if (found == ~0ull) return -1;
    slot = (slot + 1) & g.hash_mask;
  }
}

// FindNearbyInds (simulation_sampling.cpp:1068-1321): the cell that stands for cell (k, j, i) of block b when an
// index is one beyond the block. c[] = cell closest to the sample, s[] = the sample. Returns the cell's position
// in the [block][k][j][i] array, or -1 where the reference throws "Grid interpolation failed."
template <int kWhere>
__device__ long long find_nearby(const BlGridDevice &g, const RefinedTables &t, bool sks, int b, int k, int j, int i, const int c[3], const double s[3]) {
  const int n_i = g.nb[0], n_j = g.nb[1], n_k = g.nb[2];
  const size_t block_cells = (size_t)g.nb[2] * g.stride_plane;
  const int i_safe = max(min(i, n_i - 1), 0), j_safe = max(min(j, n_j - 1), 0), k_safe = max(min(k, n_k - 1), 0);
  if (i == i_safe && j == j_safe && k == k_safe)
    return (long long)(b * block_cells + (size_t)k * g.stride_plane + (size_t)j * g.stride_row + i);
  const int level = table_read<kWhere>(t, t.levels, b);
  const int li = table_read<kWhere>(t, t.locations, 3 * b), lj = table_read<kWhere>(t, t.locations, 3 * b + 1), lk = table_read<kWhere>(t, t.locations, 3 * b + 2);
  const bool upper_i = i > n_i / 2, upper_j = j > n_j / 2, upper_k = k > n_k / 2;
  const int n3 = g.n_3_level0 << level;
  const int fi = upper_i ? li * 2 + 1 : li * 2, fj = upper_j ? lj * 2 + 1 : lj * 2, fk = upper_k ? lk * 2 + 1 : lk * 2;
  // does the mesh continue beyond the block in each direction (:1098-1221)?
  bool x1_off_grid = i != i_safe, x2_off_grid = j != j_safe, x3_off_grid = k != k_safe;
  if (x1_off_grid) {
    const int d = i == -1 ? -1 : 1;
    if (find_block<kWhere>(g, t, level, li + d, lj, lk) >= 0
        || find_block<kWhere>(g, t, level - 1, i == -1 ? (li - 1) / 2 : (li + 1) / 2, lj / 2, lk / 2) >= 0
        || find_block<kWhere>(g, t, level + 1, i == -1 ? li * 2 - 1 : li * 2 + 2, fj, fk) >= 0)
      x1_off_grid = false;
  }
  if (x2_off_grid) {
    const int d = j == -1 ? -1 : 1;
    if (find_block<kWhere>(g, t, level, li, lj + d, lk) >= 0
        || find_block<kWhere>(g, t, level - 1, li / 2, j == -1 ? (lj - 1) / 2 : (lj + 1) / 2, lk / 2) >= 0
        || find_block<kWhere>(g, t, level + 1, fi, j == -1 ? lj * 2 - 1 : lj * 2 + 2, fk) >= 0)
      x2_off_grid = false;
  }
  if (x3_off_grid) {
    const int d = k == -1 ? -1 : 1;
    if (find_block<kWhere>(g, t, level, li, lj, lk + d) >= 0
        || find_block<kWhere>(g, t, level - 1, li / 2, lj / 2, k == -1 ? (lk - 1) / 2 : (lk + 1) / 2) >= 0
        || find_block<kWhere>(g, t, level + 1, fi, fj, k == -1 ? lk * 2 - 1 : lk * 2 + 2) >= 0)
      x3_off_grid = false;
    // across the periodic boundary in x^3 (:1181-1219)
    if (x3_off_grid && sks && k == -1 && lk == 0
        && (find_block<kWhere>(g, t, level, li, lj, n3 - 1) >= 0 || find_block<kWhere>(g, t, level - 1, li / 2, lj / 2, (g.n_3_level0 << (level - 1)) - 1) >= 0
            || find_block<kWhere>(g, t, level + 1, fi, fj, (g.n_3_level0 << (level + 1)) - 1) >= 0))
      x3_off_grid = false;
    if (x3_off_grid && sks && k == n_k && lk == n3 - 1
        && (find_block<kWhere>(g, t, level, li, lj, 0) >= 0 || find_block<kWhere>(g, t, level - 1, li / 2, lj / 2, 0) >= 0 || find_block<kWhere>(g, t, level + 1, fi, fj, 0) >= 0))
      x3_off_grid = false;
  }
  if (x1_off_grid) i = i_safe;
  if (x2_off_grid) j = j_safe;
  if (x3_off_grid) k = k_safe;
  const bool wrap_low = sks && k == -1 && lk == 0, wrap_high = sks && k == n_k && lk == n3 - 1;
  // same level (:1239-1261)
  {
    int lks = k == k_safe ? lk : k == -1 ? lk - 1 : lk + 1;
    if (wrap_low) lks = n3 - 1;
    if (wrap_high) lks = 0;
    const int b_alt = find_block<kWhere>(g, t, level, i == i_safe ? li : i == -1 ? li - 1 : li + 1, j == j_safe ? lj : j == -1 ? lj - 1 : lj + 1, lks);
    if (b_alt >= 0) {
      const int is = i == i_safe ? i : i == -1 ? n_i - 1 : 0, js = j == j_safe ? j : j == -1 ? n_j - 1 : 0, ks = k == k_safe ? k : k == -1 ? n_k - 1 : 0;
      return (long long)(b_alt * block_cells + (size_t)ks * g.stride_plane + (size_t)js * g.stride_row + is);
    }
  }
  // coarser level (:1264-1291)
  if (level - 1 >= 0) {
    int lks = k == k_safe ? lk / 2 : k == -1 ? (lk - 1) / 2 : (lk + 1) / 2;
    if (wrap_low) lks = (g.n_3_level0 << (level - 1)) - 1;
    if (wrap_high) lks = 0;
    const int b_alt = find_block<kWhere>(g, t, level - 1, i == i_safe ? li / 2 : i == -1 ? (li - 1) / 2 : (li + 1) / 2,
                                 j == j_safe ? lj / 2 : j == -1 ? (lj - 1) / 2 : (lj + 1) / 2, lks);
    if (b_alt >= 0) {
      const int is = i == i_safe ? (li % 2 * n_i + i) / 2 : i == -1 ? n_i - 1 : 0;
      const int js = j == j_safe ? (lj % 2 * n_j + j) / 2 : j == -1 ? n_j - 1 : 0;
      const int ks = k == k_safe ? (lk % 2 * n_k + k) / 2 : k == -1 ? n_k - 1 : 0;
      return (long long)(b_alt * block_cells + (size_t)ks * g.stride_plane + (size_t)js * g.stride_row + is);
    }
  }
  // finer level (:1294-1316)
  {
    int lks = lk * 2 + (k == k_safe ? 0 : k == -1 ? -1 : 1) + (upper_k ? 1 : 0);
    if (wrap_low && level + 1 <= g.max_level) lks = (g.n_3_level0 << (level + 1)) - 1;
    if (wrap_high) lks = 0;
    const int b_alt = find_block<kWhere>(g, t, level + 1, li * 2 + (i == i_safe ? 0 : i == -1 ? -1 : 1) + (upper_i ? 1 : 0),
                                 lj * 2 + (j == j_safe ? 0 : j == -1 ? -1 : 1) + (upper_j ? 1 : 0), lks);
    if (b_alt >= 0) {
      int is = i == i_safe ? (upper_i ? (i - n_i / 2) * 2 : i * 2) : i == -1 ? n_i - 2 : 0;
      int js = j == j_safe ? (upper_j ? (j - n_j / 2) * 2 : j * 2) : j == -1 ? n_j - 2 : 0;
      int ks = k == k_safe ? (upper_k ? (k - n_k / 2) * 2 : k * 2) : k == -1 ? n_k - 2 : 0;
      const double *x1v = t.bxv[0] + (size_t)table_read<kWhere>(t, t.block_row[0], b) * n_i, *x2v = t.bxv[1] + (size_t)table_read<kWhere>(t, t.block_row[1], b) * n_j,
                   *x3v = t.bxv[2] + (size_t)table_read<kWhere>(t, t.block_row[2], b) * n_k;
      ks += (k < c[2] || (k == c[2] && s[2] > table_read<kWhere>(t, x3v, c[2]))) ? 1 : 0;
      js += (j < c[1] || (j == c[1] && s[1] > table_read<kWhere>(t, x2v, c[1]))) ? 1 : 0;
      is += (i < c[0] || (i == c[0] && s[0] > table_read<kWhere>(t, x1v, c[0]))) ? 1 : 0;
      return (long long)(b_alt * block_cells + (size_t)ks * g.stride_plane + (size_t)js * g.stride_row + is);
    }
  }
  return -1;
}

// The same on a mesh with refinement: the block from the lattice of block boundaries, then the cell inside
// it from the block's own coordinate rows (global memory; this path is not the benchmark's).
// What the locate kernel finds out about one sample (stored as BlLocated + tag)
struct LocatedSample {
  double f_i, f_j, f_k, ph;
  uint32_t cell, status;
};

// Where along a table of ascending faces the search for s starts: the position the table's spacing predicts (BlGridDevice::row_guess,
// box_guess: kind, origin, cells per unit)
__device__ __forceinline__ int guessed_index(double kind, double origin, double per_unit, double s, int n) {
  const double from_origin = (kind != 0.0 ? (double)__log2f((float)s) : s) - origin;
  const int i = (int)(from_origin * per_unit);
  return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
}
// The first of n cells whose upper face is >= s (the rule of the reference's linear scans, simulation_sampling.cpp:458-466;
// faces[0] <= s <= faces[n]), walked to from i: the cell does not depend on where the walk starts
template <int kWhere>
__device__ __forceinline__ int walk_to_cell(const RefinedTables &t, const double *faces, int n, double s, int i) {
  while (i < n - 1 && table_read<kWhere>(t, faces, i + 1) < s) i++;
  while (i > 0 && table_read<kWhere>(t, faces, i) >= s) i--;
  return i;
}
// ... and whether i is that cell already, from its two faces
__device__ __forceinline__ bool is_the_cell(int i, int n, double s, double face_lo, double face_hi) {
  return (i == n - 1 || face_hi >= s) && (i == 0 || face_lo < s);
}

// g: P.grid, or the kernel's copy of it whose tables point into LDS (bl_locate_kernel)
// The search is a straight line of five rounds of table reads - the edges around the guessed box of the block lattice, the box's
// block, the block's rows, the rows' spacing, faces and centres around the guessed cell - with the reads of a round independent
// of one another; a guess its faces do not confirm (a table that is not evenly spaced, the last place of a float log2) walks.
// angle_band > 0: s2 and s3 are the tolerant tier's angles (bl_fastmath.h: below 1e-15 of the exact tier's). The search then says
// whether they decide it - every value they were compared with further away than the band: the faces and the centre of the cell
// found, the ends of the mesh - and returns false BEFORE anything is counted or written where they do not: the caller searches
// again with the exact tier's angles. (The radius is the exact tier's in both tiers.)
// defer_nearby: a sample of inter-block interpolation with an anchor beyond its own block (one in twenty with 64^3 blocks) is not
// finished here - FindNearbyInds for eight corners, a dozen hash probes each, which a wave would run for its one or two such lanes
// with the others idle - but comes back as kSamplePending, nothing counted or written: the locate kernel collects those and runs
// them through this function again 64 to a wave.
// kWhere: kTableHbm, kTableLds, kTableAnywhere. kNearby false: the instantiation of the locate kernel's main loop, which defers those samples always and has no FindNearbyInds in it.
template <int kWhere, bool kNearby>
__device__ __forceinline__ bool locate_sample_refined(const BlShadeArgs &P, const RefinedTables &t, unsigned int *anchors, double s1, double s2, double s3,
                                                      LocatedSample *out, unsigned long long *gathers, double angle_band = 0.0, bool defer_nearby = false) {
  const BlPlasmaDevice &pl = P.plasma;
  const BlGridDevice &g = P.grid;   // (scalars only below: the tables are t's)
  const double s[3] = {s1, s2, s3};
  int box[3];
  {
    bool off_grid = false, confirmed = true;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      off_grid = off_grid || s[a] < g.edge_first[a] || s[a] > g.edge_last[a];
      box[a] = guessed_index(g.box_guess[a][0], g.box_guess[a][1], g.box_guess[a][2], s[a], g.n_edge[a]);
    }
    if (off_grid) {
      if (angle_band > 0.0 && !(s[0] < g.edge_first[0] || s[0] > g.edge_last[0])) return false;   // (off the mesh by an angle)
      out->status = kSampleOffGrid;
      return true;
    }
    double lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
      lo[a] = table_read<kWhere>(t, t.edge[a], box[a]);
      hi[a] = table_read<kWhere>(t, t.edge[a], box[a] + 1);
    }
#pragma unroll
    for (int a = 0; a < 3; a++) confirmed = confirmed && is_the_cell(box[a], g.n_edge[a], s[a], lo[a], hi[a]);
    if (!confirmed) {
#pragma unroll
      for (int a = 0; a < 3; a++) box[a] = walk_to_cell<kWhere>(t, t.edge[a], g.n_edge[a], s[a], box[a]);
    }
  }
  const int b = table_read<kWhere>(t, t.lattice, ((size_t)box[2] * g.n_edge[1] + box[1]) * g.n_edge[0] + box[0]);
  if (b < 0) {
    if (angle_band > 0.0) return false;   // (a hole in the mesh: which box it is hangs on the angles)
    out->status = kSampleOffGrid;
    return true;
  }
  int c[3];
  double face_lo[3], face_hi[3];        // faces of cell c
  double centre[3][3];                  // centres of cells c - 1, c, c + 1 (the row's first / last once more at its ends)
  {
    int row[3];
#pragma unroll
    for (int a = 0; a < 3; a++) row[a] = table_read<kWhere>(t, t.block_row[a], b);
    const double *xf[3], *xv[3];
    double guess[3][3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const int n = g.nb[a];
      xf[a] = t.bxf[a] + (size_t)row[a] * (n + 1);
      xv[a] = t.bxv[a] + (size_t)row[a] * n;
#pragma unroll
      for (int q = 0; q < 3; q++) guess[a][q] = table_read<kWhere>(t, t.row_guess[a], 3 * (size_t)row[a] + q);
    }
#pragma unroll
    for (int a = 0; a < 3; a++) c[a] = guessed_index(guess[a][0], guess[a][1], guess[a][2], s[a], g.nb[a]);
    auto read_around = [&](int a) {
      const int n = g.nb[a], i = c[a];
      face_lo[a] = table_read<kWhere>(t, xf[a], i);
      face_hi[a] = table_read<kWhere>(t, xf[a], i + 1);
      centre[a][0] = table_read<kWhere>(t, xv[a], i > 0 ? i - 1 : 0);
      centre[a][1] = table_read<kWhere>(t, xv[a], i);
      centre[a][2] = table_read<kWhere>(t, xv[a], i < n - 1 ? i + 1 : n - 1);
    };
#pragma unroll
    for (int a = 0; a < 3; a++) read_around(a);
    bool confirmed = true;
#pragma unroll
    for (int a = 0; a < 3; a++) confirmed = confirmed && is_the_cell(c[a], g.nb[a], s[a], face_lo[a], face_hi[a]);
    if (!confirmed) {
#pragma unroll
      for (int a = 0; a < 3; a++) {
        c[a] = walk_to_cell<kWhere>(t, xf[a], g.nb[a], s[a], c[a]);
        read_around(a);
      }
    }
  }
  if (angle_band > 0.0) {
    double margin = __builtin_inf();
#pragma unroll
    for (int a = 1; a < 3; a++) {
      const double d_lo = s[a] - face_lo[a], d_hi = face_hi[a] - s[a], d_c = __builtin_fabs(s[a] - centre[a][1]);
      margin = margin < d_lo ? margin : d_lo;
      margin = margin < d_hi ? margin : d_hi;
      margin = margin < d_c ? margin : d_c;
    }
    if (!(margin > angle_band)) return false;
  }
  if ((defer_nearby || !kNearby) && pl.simulation_interp && g.block_interp) {
    bool inside = true;
#pragma unroll
    for (int a = 0; a < 3; a++) inside = inside && (s[a] >= centre[a][1] ? c[a] + 1 < g.nb[a] : c[a] >= 1);
    if (!inside) {
      out->status = kSamplePending;
      return true;
    }
  }
  *gathers += 1ull;
  const size_t block_base = (size_t)b * g.nb[2] * g.stride_plane;
  if (!pl.simulation_interp) {
    out->status = kSampleNearest;
    out->cell = (uint32_t)(block_base + (size_t)c[2] * g.stride_plane + (size_t)c[1] * g.stride_row + c[0]);
    return true;
  }
  if (g.block_interp) {   // inter-block interpolation (:505-546)
    int m[3], pp[3];
    double f[3];
    bool undefined = false;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const int n = g.nb[a], i = c[a];
      const double xv_i = centre[a][1];
      const bool upper = s[a] >= xv_i;
      m[a] = upper ? i : i - 1;
      pp[a] = m[a] + 1;
      // :520-522 read x1v(b, i + 1) at a block's upper edge: the next block's first centre in the reference's
      // Array (xv[a][i + 1] here as well: rows are contiguous); past the array for the last block - undefined
      const bool past_the_array = pp[a] == n && b == g.n_blocks - 1;
      if (past_the_array) undefined = true;
      const double x_m = m[a] == -1 ? 2.0 * face_lo[a] - xv_i : (upper ? xv_i : centre[a][0]);
      // BL_UNDEFINED_EDGE: the centre mirrored about the block's upper face, the rule the lower edge has (x_m above)
      const double x_p = (pp[a] == n) ? (past_the_array ? 2.0 * face_hi[a] - xv_i : 2.0 * table_read<kWhere>(t, t.xv_next[a], b) - xv_i) : (upper ? centre[a][2] : xv_i);
      f[a] = blm_div(s[a] - x_m, x_p - x_m);   // (ordinary operands: the IEEE quotient, blmath.h)
    }
    if (undefined) {
      atomicAdd(&P.counters[BL_CNT_UNDEFINED], 1ull);
      if (!P.undefined_edge) {
        out->status = kSampleCut;
        return true;
      }
    }
    // All eight anchors inside the sample's own block (all but the outermost half cell of a block: 95 % of the samples with 64^3
    // blocks): FindNearbyInds returns each index as it is (:1075-1079) and InterpolateAdvanced has InterpolateSimple's weights and
    // order (:1365-1386 / :1334-1351), so the sample is an ordinary trilinear one anchored at cell (m_k, m_j, m_i) - no anchor list
    // written here or read by the coefficient kernel.
    if (m[0] >= 0 && m[1] >= 0 && m[2] >= 0 && pp[0] < g.nb[0] && pp[1] < g.nb[1] && pp[2] < g.nb[2]) {
      out->f_i = f[0];
      out->f_j = f[1];
      out->f_k = f[2];
      out->status = kSampleInterp;
      out->cell = (uint32_t)(block_base + (size_t)m[2] * g.stride_plane + (size_t)m[1] * g.stride_row + m[0]);
      return true;
    }
    const bool sks = pl.simulation_coord == BL_COORD_SKS;
    bool failed = false;
    if (kNearby) {   // (the other instantiation has left above)
      for (int corner = 0; corner < 8; corner++) {
        const long long cell = find_nearby<kWhere>(g, t, sks, b, (corner & 4) ? pp[2] : m[2], (corner & 2) ? pp[1] : m[1], (corner & 1) ? pp[0] : m[0], c, s);
        failed = failed || cell < 0;
        anchors[corner] = (unsigned int)cell;
      }
    }
    if (failed) {
      atomicAdd(&P.counters[BL_CNT_INTERP_FAILED], 1ull);
      out->status = kSampleCut;
      return true;
    }
    out->f_i = f[0];
    out->f_j = f[1];
    out->f_k = f[2];
    out->status = kSampleAdvanced;
    out->cell = anchors[0];
    return true;
  }
  int m[3];
  double f[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {   // :485-490 with the block's own centres
    const int i = c[a];
    const bool upper = i == 0 || (i != g.nb[a] - 1 && s[a] >= centre[a][1]);
    m[a] = upper ? i : i - 1;
    const double x_m = upper ? centre[a][1] : centre[a][0], x_p = upper ? centre[a][2] : centre[a][1];
    f[a] = blm_div(s[a] - x_m, x_p - x_m);   // (ordinary operands: the IEEE quotient, blmath.h)
  }
  out->f_i = f[0];
  out->f_j = f[1];
  out->f_k = f[2];
  out->status = kSampleInterp;
  out->cell = (uint32_t)(block_base + (size_t)m[2] * g.stride_plane + (size_t)m[1] * g.stride_row + m[0]);
  return true;
}

// Locate one sample on the simulation grid: ConvertFromCKS (radiation_geometry.cpp:37-57), block test
// and cell search (simulation_sampling.cpp:352-394, :458-490), trilinear fractions (:736-760).
// Slow light: which time slice(s) a sample at coordinate time x0 reads (simulation_sampling.cpp:296-349).
// Returns t_ind; *t_frac for slow_interp. Extrapolation beyond the window is recorded per ray and as maxima.
__device__ __forceinline__ int locate_time(const BlSlowDevice &sl, double x0, uint32_t ray, double *t_frac) {
  const double *time = sl.times;
  const int chunk = sl.n;
  const double tolerance = 1.0;   // simulation_reader.hpp:99
  int t_ind = 0;
  *t_frac = 0.0;
  int kind = -1;
  double by = 0.0;
  if (x0 >= time[0]) {
    by = x0 - time[0];
    if (x0 > time[0] + tolerance) kind = 1;
    else if (x0 > time[0]) kind = 0;
  } else if (x0 <= time[chunk - 1]) {
    by = time[chunk - 1] - x0;
    if (x0 < time[chunk - 1] - tolerance) kind = 3;
    else if (x0 < time[chunk - 1]) kind = 2;
    if (sl.interp) {
      t_ind = chunk - 2;
      *t_frac = 1.0;
    } else {
      t_ind = chunk - 1;
    }
  } else {
    while (time[t_ind] > x0) t_ind++;   // first slice not later than the sample (exists: x0 > time[chunk - 1])
    if (sl.interp) {
      t_ind--;
      *t_frac = (x0 - time[t_ind]) / (time[t_ind + 1] - time[t_ind]);
    } else if (time[t_ind - 1] - x0 <= x0 - time[t_ind]) {
      t_ind--;
    }
  }
  if (kind >= 0) {
    atomicOr(&sl.ray_extrap[ray], 1u << kind);
    atomicMax(&sl.extrap_max[kind], (unsigned long long)__double_as_longlong(by));
  }
  return t_ind;
}

// Block test, cell search and fractions for a sample at (s1, s2, s3) in the simulation's coordinates. angle_band > 0 (spherical
// grids, not FMKS): s2 and s3 are the tolerant tier's angles, and the function returns false, before anything is counted or written,
// where they do not decide the search (locate_sample_refined; locate_sample_tolerant in bl_sampling_fast.h is the caller).
// kRefined: refined (a mesh with refinement: its tables, where they lie in HBM when null; kWhere: where else they may be) instead of tab.
template <bool kRefined, int kWhere = kTableHbm, bool kNearby = true>
__device__ __forceinline__ bool locate_from_coordinates(const BlShadeArgs &P, const GridTables &tab, const RefinedTables *refined, double s1, double s2, double s3,
                                                        LocatedSample *out, unsigned long long *gathers, unsigned int *anchors, double angle_band = 0.0,
                                                        bool defer_nearby = false) {
  const BlPlasmaDevice &pl = P.plasma;
  const BlGridDevice &g = P.grid;
  if (kRefined) {
    if (refined != nullptr) return locate_sample_refined<kWhere, kNearby>(P, *refined, anchors, s1, s2, s3, out, gathers, angle_band, defer_nearby);
    const RefinedTables in_hbm = refined_tables_in_hbm(g);
    return locate_sample_refined<kTableHbm, kNearby>(P, in_hbm, anchors, s1, s2, s3, out, gathers, angle_band, defer_nearby);
  }
  const int n_i = g.n[0], n_j = g.n[1], n_k = g.n[2];
  if (g.fmks) {
    // FMKS grid (simulation_sampling.cpp:190-198, :396-456): bounds in (r, theta, phi); (r, theta) -> (x^1, x^2) from the
    // reader's table by plain scaling (its x^2 read at row j + 1 in both terms, as written there); zone and fraction from
    // the lower FACE in x^1 and x^2 (no half-cell shift), the usual centre rule in x^3. The reference reads cells
    // (k_m .. k_m + 1, j_m .. j_m + 1, i_m .. i_m + 1) of an array without bounds: one beyond a row is the next row's
    // cell - reproduced (the gather steps through the same linear order) - but beyond the block it is another variable's
    // data: no value to reproduce, counted as undefined.
    if (!(s1 >= g.fmks_bounds[0] && s1 <= g.fmks_bounds[1] && s2 >= g.fmks_bounds[2] && s2 <= g.fmks_bounds[3]
          && s3 >= g.fmks_bounds[4] && s3 <= g.fmks_bounds[5])) {
      out->status = kSampleOffGrid;
      return true;
    }
    const size_t m1 = (size_t)g.sks_map_n1, m2 = (size_t)g.sks_map_n2;
    double i_ind, j_ind;
    double f_i = bl_modf((s1 - g.sks_map_r_in) / g.sks_map_dr, &i_ind);
    double f_j = bl_modf(s2 / g.sks_map_dtheta, &j_ind);
    size_t mi = (size_t)(int)i_ind, mj = (size_t)(int)j_ind;
    if (mi + 1 >= m1 || mj + 1 >= m2) {   // r = r_out or theta = pi exactly: the reference reads past the table
      atomicAdd(&P.counters[BL_CNT_UNDEFINED], 1ull);
      if (!P.undefined_edge) {
        out->status = kSampleCut;
        return true;
      }
      if (mi + 1 >= m1) { mi = m1 - 2; f_i = 1.0; }   // BL_UNDEFINED_EDGE: the table's last entry
      if (mj + 1 >= m2) { mj = m2 - 2; f_j = 1.0; }
    }
    const double fmks_x1 = (1.0 - f_i) * g.sks_map[mj * m1 + mi] + f_i * g.sks_map[mj * m1 + mi + 1];
    const double map_x2 = g.sks_map[(m2 + mj + 1) * m1 + mi];
    const double fmks_x2 = (1.0 - f_j) * map_x2 + f_j * map_x2;
    f_i = bl_modf((fmks_x1 - g.fmks_x1_0) / g.fmks_dx1, &i_ind);
    f_j = bl_modf(fmks_x2 / g.fmks_dx2, &j_ind);
    int i_m = (int)i_ind, j_m = (int)j_ind;
    const int k = find_cell(g, tab, 2, s3);
    const int k_m = (k == 0 || (k != n_k - 1 && s3 >= tab.xv[2][k])) ? k : k - 1;
    const double f_k = (s3 - tab.xv[2][k_m]) / (tab.xv[2][k_m + 1] - tab.xv[2][k_m]);
    const long long n_cells = (long long)n_k * n_j * n_i;
    long long first, last;
    if (!pl.simulation_interp) {
      first = last = ((long long)k * n_j + (f_j >= 0.5 ? j_m + 1 : j_m)) * n_i + (f_i >= 0.5 ? i_m + 1 : i_m);
    } else {
      first = ((long long)k_m * n_j + j_m) * n_i + i_m;
      last = ((long long)(k_m + 1) * n_j + (j_m + 1)) * n_i + (i_m + 1);
    }
    if (first < 0 || last >= n_cells) {
      atomicAdd(&P.counters[BL_CNT_UNDEFINED], 1ull);
      if (!P.undefined_edge) {
        out->status = kSampleCut;
        return true;
      }
      // BL_UNDEFINED_EDGE: the zone's own row / column stands for the missing one
      if (pl.simulation_interp) {
        if (j_m + 1 >= n_j) { j_m = n_j - 2; f_j = 1.0; }
        if (i_m + 1 >= n_i) { i_m = n_i - 2; f_i = 1.0; }
        first = ((long long)k_m * n_j + j_m) * n_i + i_m;
      } else {
        const int jn = min(f_j >= 0.5 ? j_m + 1 : j_m, n_j - 1), in = min(f_i >= 0.5 ? i_m + 1 : i_m, n_i - 1);
        first = ((long long)k * n_j + jn) * n_i + in;
      }
    }
    *gathers += 1ull;
    out->cell = (uint32_t)first;
    if (!pl.simulation_interp) {
      out->status = kSampleNearest;
    } else {
      out->f_i = f_i;
      out->f_j = f_j;
      out->f_k = f_k;
      out->status = kSampleInterp;
    }
    return true;
  }
  if (s1 < tab.xf[0][0] || s1 > tab.xf[0][n_i] || s2 < tab.xf[1][0] || s2 > tab.xf[1][n_j]
      || s3 < tab.xf[2][0] || s3 > tab.xf[2][n_k]) {
    if (angle_band > 0.0 && !(s1 < tab.xf[0][0] || s1 > tab.xf[0][n_i])) return false;   // (off the grid by an angle)
    out->status = kSampleOffGrid;
    return true;
  }
  int i = find_cell(g, tab, 0, s1);
  int j = find_cell(g, tab, 1, s2);
  int k = find_cell(g, tab, 2, s3);
  if (angle_band > 0.0) {   // the angles against the faces and the centre of the cell they found
    const double d_j = fmin(fmin(s2 - tab.xf[1][j], tab.xf[1][j + 1] - s2), __builtin_fabs(s2 - tab.xv[1][j]));
    const double d_k = fmin(fmin(s3 - tab.xf[2][k], tab.xf[2][k + 1] - s3), __builtin_fabs(s3 - tab.xv[2][k]));
    if (!(fmin(d_j, d_k) > angle_band)) return false;
  }
  *gathers += 1ull;
  if (!pl.simulation_interp) {   // :710-734
    out->status = kSampleNearest;
    out->cell = (uint32_t)((k * n_j + j) * n_i + i);
    return true;
  }
  // :485-490
  // the anchor rule is per block (the indices of :485-487 are block-local): several equal blocks live in one
  // merged array here, g.nb is the block size
  // (one block - the usual case, known to the whole launch - needs no remainders: three integer divisions per sample)
  const bool one_block = g.nb[0] == n_i && g.nb[1] == n_j && g.nb[2] == n_k;
  const int i_b = one_block ? i : i % g.nb[0], j_b = one_block ? j : j % g.nb[1], k_b = one_block ? k : k % g.nb[2];
  int i_m = (i_b == 0 || (i_b != g.nb[0] - 1 && s1 >= tab.xv[0][i])) ? i : i - 1;
  int j_m = (j_b == 0 || (j_b != g.nb[1] - 1 && s2 >= tab.xv[1][j])) ? j : j - 1;
  int k_m = (k_b == 0 || (k_b != g.nb[2] - 1 && s3 >= tab.xv[2][k])) ? k : k - 1;
  // (fractions of cell widths: ordinary operands for the short division; a sample exactly on a centre gives 0 / width = 0)
  out->f_i = blm_div(s1 - tab.xv[0][i_m], tab.xv[0][i_m + 1] - tab.xv[0][i_m]);
  out->f_j = blm_div(s2 - tab.xv[1][j_m], tab.xv[1][j_m + 1] - tab.xv[1][j_m]);
  out->f_k = blm_div(s3 - tab.xv[2][k_m], tab.xv[2][k_m + 1] - tab.xv[2][k_m]);
  out->status = kSampleInterp;
  out->cell = (uint32_t)((k_m * n_j + j_m) * n_i + i_m);
  return true;
}

// Locate one sample: ConvertFromCKS (radiation_geometry.cpp:37-57) with the pinned inverse trigonometric functions, then the search
template <bool kRefined, bool kSpinZero, int kWhere = kTableHbm, bool kNearby = true>
__device__ __forceinline__ void locate_sample(const BlShadeArgs &P, const GridTables &tab, const BlSpacetime &st,
                                              double x1, double x2, double x3, double r, LocatedSample *out,
                                              unsigned long long *gathers, unsigned int *anchors, const RefinedTables *refined = nullptr,
                                              bool defer_nearby = false) {
  const bool sks = P.plasma.simulation_coord == BL_COORD_SKS;
  double s1 = x1, s2 = x2, s3 = x3;
  out->ph = 0.0;
  out->f_i = out->f_j = out->f_k = 0.0;
  out->cell = 0u;
  if (sks) {
    // z / r is cos(theta) in ConvertFromCKS, in the SKS metric and in the Jacobian (same expression)
    double th = bl_acos(blm_div(x3, r));   // (|z| <= r, both of the order of the coordinates: ordinary operands)
    // zero spin: atan(0 / r) = +0 and atan2(y, x) - 0 = atan2(y, x)
    double ph = kSpinZero ? bl_atan2(x2, x1) : bl_atan2(x2, x1) - bl_atan(blm_div(st.bh_a, r));
    out->ph = ph;
    ph += ph < 0.0 ? 2.0 * kPi : 0.0;
    ph -= ph >= 2.0 * kPi ? 2.0 * kPi : 0.0;
    s1 = r;
    s2 = th;
    s3 = ph;
  }
  locate_from_coordinates<kRefined, kWhere, kNearby>(P, tab, refined, s1, s2, s3, out, gathers, anchors, 0.0, defer_nearby);
}

// SampleSimulation's nearest / trilinear read (simulation_sampling.cpp:666-1033, InterpolateSimple
// :1334-1351) for a located sample: pr = rho, pgas, uu1..3, bb1..3 as float.
__device__ __forceinline__ void sample_primitives(const BlShadeArgs &P, int status, uint32_t cell, double f_i,
                                                  double f_j, double f_k, float pr[8]) {
  const BlPlasmaDevice &pl = P.plasma;
  const BlGridDevice &g = P.grid;
  const float4 *base = reinterpret_cast<const float4 *>(g.cells) + (size_t)cell * 2;
  if (status == kSampleInterp) {
    // the 8 corner cells: 16 independent 16-byte loads in flight per lane; the two cells of an
    // i-pair are one contiguous 64-byte segment
    const size_t row = (size_t)g.stride_row * 2, plane = (size_t)g.stride_plane * 2;
    float4 lo[8], hi[8];
#pragma unroll
    for (int corner = 0; corner < 8; corner++) {
      const int dk = corner >> 2, dj = (corner >> 1) & 1, di = corner & 1;
      const float4 *p = base + dk * plane + dj * row + di * 2;
      lo[corner] = p[0];
      hi[corner] = p[1];
    }
    // InterpolateSimple sums w_c * v_c over the corners in the order mmm, mmp, mpm, mpp, pmm, pmp,
    // ppm, ppp (:1345-1349); accumulating corner by corner keeps that order.
    const double w_k[2] = {1.0 - f_k, f_k}, w_j[2] = {1.0 - f_j, f_j}, w_i[2] = {1.0 - f_i, f_i};
    double val[8];
    float first[8];
#pragma unroll
    for (int corner = 0; corner < 8; corner++) {
      const int dk = corner >> 2, dj = (corner >> 1) & 1, di = corner & 1;
      float c[8];
      unpack_cell(lo[corner], hi[corner], c);
      double w = w_k[dk] * w_j[dj] * w_i[di];
#pragma unroll
      for (int v = 0; v < 8; v++) {
        if (corner == 0) {
          val[v] = w * (double)c[v];
          first[v] = c[v];
        } else {
          val[v] += w * (double)c[v];
        }
      }
    }
    if (val[0] <= 0.0) val[0] = (double)first[0];   // :822-825
    if (val[1] <= 0.0) val[1] = (double)first[1];
#pragma unroll
    for (int v = 0; v < 8; v++) pr[v] = (float)val[v];   // :830-839
  } else if (status == kSampleNearest) {
    unpack_cell(base[0], base[1], pr);
  } else if (status == kSampleOffGrid) {
    const float fnan = __int_as_float(0x7fc00000);
    pr[0] = pl.fallback_nan ? fnan : P.cold->fallback_rho;    // :377-384, :678-706
    pr[1] = pl.fallback_nan ? fnan : P.cold->fallback_pgas;
    for (int v = 2; v < 8; v++) pr[v] = pl.fallback_nan ? fnan : 0.0f;
  } else {
    for (int v = 0; v < 8; v++) pr[v] = 0.0f;
  }
}

// Inter-block interpolation: the nine values of a sample from its eight anchor cells (InterpolateAdvanced,
// simulation_sampling.cpp:1365-1386: the weights and order of InterpolateSimple; "<= 0 -> first anchor", :936-945)
__device__ __forceinline__ void sample_primitives_advanced(const BlShadeArgs &P, const unsigned int *anchors, double f_i, double f_j,
                                                           double f_k, float pr[8], float *kappa_out) {
  const BlGridDevice &g = P.grid;
  const double w_k[2] = {1.0 - f_k, f_k}, w_j[2] = {1.0 - f_j, f_j}, w_i[2] = {1.0 - f_i, f_i};
  double val[9];
  float first[9];
  for (int corner = 0; corner < 8; corner++) {
    const unsigned int cell = anchors[corner];
    const float4 *p = reinterpret_cast<const float4 *>(g.cells) + (size_t)cell * 2;
    float c[9];
    unpack_cell(p[0], p[1], c);
    c[8] = g.kappa != nullptr ? g.kappa[cell] : 0.0f;
    const double w = w_k[corner >> 2] * w_j[(corner >> 1) & 1] * w_i[corner & 1];
    for (int v = 0; v < 9; v++) {
      if (corner == 0) {
        val[v] = w * (double)c[v];
        first[v] = c[v];
      } else {
        val[v] += w * (double)c[v];
      }
    }
  }
  if (val[0] <= 0.0) val[0] = (double)first[0];
  if (val[1] <= 0.0) val[1] = (double)first[1];
  if (val[8] <= 0.0) val[8] = (double)first[8];
  for (int v = 0; v < 8; v++) pr[v] = (float)val[v];
  *kappa_out = (float)val[8];
}

// The electron entropy of a located sample (plasma_model = code_kappa; simulation_sampling.cpp:726-727,
// :812-833): same nearest / trilinear rule as the other primitives, from its own array.
__device__ __forceinline__ float sample_kappa(const BlShadeArgs &P, int status, uint32_t cell, double f_i, double f_j,
                                              double f_k) {
  const BlGridDevice &g = P.grid;
  const float *base = g.kappa + cell;
  if (status == kSampleInterp) {
    const size_t row = (size_t)g.stride_row, plane = (size_t)g.stride_plane;
    float c[8];
#pragma unroll
    for (int corner = 0; corner < 8; corner++)
      c[corner] = base[(corner >> 2) * plane + ((corner >> 1) & 1) * row + (corner & 1)];
    const double w_k[2] = {1.0 - f_k, f_k}, w_j[2] = {1.0 - f_j, f_j}, w_i[2] = {1.0 - f_i, f_i};
    double val = 0.0;
#pragma unroll
    for (int corner = 0; corner < 8; corner++) {
      const double w = w_k[corner >> 2] * w_j[(corner >> 1) & 1] * w_i[corner & 1];
      val = corner == 0 ? w * (double)c[0] : val + w * (double)c[corner];
    }
    if (val <= 0.0) val = (double)c[0];   // :826-827
    return (float)val;
  }
  if (status == kSampleNearest) return base[0];
  if (status == kSampleOffGrid) return P.plasma.fallback_nan ? __int_as_float(0x7fc00000) : P.cold->fallback_kappa;
  return 0.0f;
}

// Slow light: the nine values of a located sample from time slice t_ind, or blended linearly in time with
// slice t_ind + 1 (simulation_sampling.cpp:710-786 nearest, :809-912 trilinear). Spatial interpolation and the
// "<= 0 -> anchor cell" rule of rho, pgas, kappa apply per slice, before the blend; values stay double until the
// final conversion to float. Not on the benchmark path: plain loops, the extended instantiation only.
__device__ __forceinline__ void sample_slice_values(const BlShadeArgs &P, const float *cells, const float *kappa,
                                                    int status, uint32_t cell, double f_i, double f_j, double f_k,
                                                    double val[9]) {
  const BlGridDevice &g = P.grid;
  const size_t row = (size_t)g.stride_row, plane = (size_t)g.stride_plane;
  if (status == kSampleNearest) {
    const float *c = cells + (size_t)cell * 8;
    for (int v = 0; v < 8; v++) val[v] = (double)c[v];
    val[8] = kappa != nullptr ? (double)kappa[cell] : 0.0;
    return;
  }
  const double w_k[2] = {1.0 - f_k, f_k}, w_j[2] = {1.0 - f_j, f_j}, w_i[2] = {1.0 - f_i, f_i};
  double first[9];
  for (int corner = 0; corner < 8; corner++) {
    const int dk = corner >> 2, dj = (corner >> 1) & 1, di = corner & 1;
    const size_t at = (size_t)cell + dk * plane + dj * row + di;
    const double w = w_k[dk] * w_j[dj] * w_i[di];
    const float *c = cells + at * 8;
    for (int v = 0; v < 9; v++) {
      const double x = v < 8 ? (double)c[v] : (kappa != nullptr ? (double)kappa[at] : 0.0);
      if (corner == 0) {
        val[v] = w * x;
        first[v] = x;
      } else {
        val[v] += w * x;
      }
    }
  }
  if (val[0] <= 0.0) val[0] = first[0];
  if (val[1] <= 0.0) val[1] = first[1];
  if (kappa != nullptr && val[8] <= 0.0) val[8] = first[8];
}

// Slow light with inter-block interpolation: the nine values of one time slice from the eight anchor cells, double,
// with the "<= 0 -> first anchor" rule per slice (simulation_sampling.cpp:960-1033)
__device__ __forceinline__ void sample_slice_values_advanced(const BlShadeArgs &P, const float *cells, const float *kappa,
                                                             const unsigned int *anchors, double f_i, double f_j, double f_k,
                                                             double val[9]) {
  const double w_k[2] = {1.0 - f_k, f_k}, w_j[2] = {1.0 - f_j, f_j}, w_i[2] = {1.0 - f_i, f_i};
  double first[9];
  for (int corner = 0; corner < 8; corner++) {
    const unsigned int cell = anchors[corner];
    const float *c = cells + (size_t)cell * 8;
    const double w = w_k[corner >> 2] * w_j[(corner >> 1) & 1] * w_i[corner & 1];
    for (int v = 0; v < 9; v++) {
      const double x = v < 8 ? (double)c[v] : (kappa != nullptr ? (double)kappa[cell] : 0.0);
      if (corner == 0) {
        val[v] = w * x;
        first[v] = x;
      } else {
        val[v] += w * x;
      }
    }
  }
  if (val[0] <= 0.0) val[0] = first[0];
  if (val[1] <= 0.0) val[1] = first[1];
  if (val[8] <= 0.0) val[8] = first[8];
}

__device__ __forceinline__ void sample_primitives_slow(const BlShadeArgs &P, int status, uint32_t cell, const unsigned int *anchors,
                                                       int t_ind, double t_frac, double f_i, double f_j, double f_k, float pr[8],
                                                       float *kappa_out) {
  const BlPlasmaDevice &pl = P.plasma;
  const BlSlowDevice &sl = P.slow;
  if (status == kSampleAdvanced) {
    const bool entropy = pl.code_kappa != 0;
    double val[9];
    sample_slice_values_advanced(P, sl.cells[t_ind], entropy ? sl.kappa[t_ind] : nullptr, anchors, f_i, f_j, f_k, val);
    if (sl.interp) {
      double next[9];
      sample_slice_values_advanced(P, sl.cells[t_ind + 1], entropy ? sl.kappa[t_ind + 1] : nullptr, anchors, f_i, f_j, f_k, next);
      for (int v = 0; v < 9; v++) val[v] = (1.0 - t_frac) * val[v] + t_frac * next[v];
    }
    for (int v = 0; v < 8; v++) pr[v] = (float)val[v];
    *kappa_out = entropy ? (float)val[8] : 0.0f;
  } else if (status == kSampleInterp || status == kSampleNearest) {
    const bool entropy = pl.code_kappa != 0;
    double val[9];
    sample_slice_values(P, sl.cells[t_ind], entropy ? sl.kappa[t_ind] : nullptr, status, cell, f_i, f_j, f_k, val);
    if (sl.interp) {
      double next[9];
      sample_slice_values(P, sl.cells[t_ind + 1], entropy ? sl.kappa[t_ind + 1] : nullptr, status, cell, f_i, f_j, f_k, next);
      for (int v = 0; v < 9; v++) val[v] = (1.0 - t_frac) * val[v] + t_frac * next[v];
    }
    for (int v = 0; v < 8; v++) pr[v] = (float)val[v];
    *kappa_out = entropy ? (float)val[8] : 0.0f;
  } else if (status == kSampleOffGrid) {
    const float fnan = __int_as_float(0x7fc00000);
    pr[0] = pl.fallback_nan ? fnan : P.cold->fallback_rho;
    pr[1] = pl.fallback_nan ? fnan : P.cold->fallback_pgas;
    for (int v = 2; v < 8; v++) pr[v] = pl.fallback_nan ? fnan : 0.0f;
    *kappa_out = pl.fallback_nan ? fnan : P.cold->fallback_kappa;
  } else {
    for (int v = 0; v < 8; v++) pr[v] = 0.0f;
    *kappa_out = 0.0f;
  }
}

// Simulation mode: the frequency-independent part of CalculateSimulationCoefficients
// (simulation_coefficients.cpp:253-455). kExtended: the instantiation that also knows plasma_model = code_kappa.
template <bool kExtended, bool kSksCurved>
__device__ __forceinline__ void sample_finish_simulation(const BlShadeArgs &P, const BlSpacetime &st,
                                                         const BlKerrSchild &ks, double cth, double ph_unwrapped,
                                                         const float pr[8], float kappa_f, const double kcov[4],
                                                         int need_coefficients, SampleShade *out, BlPolSample *pol_out) {
  const BlPlasmaDevice &pl = P.plasma;
  const double bh_a = st.bh_a, bh_m = st.bh_m;
  const double r = ks.r, r2 = ks.r2, a2 = ks.a2;
  // kSksCurved: spherical Kerr-Schild simulation in a curved spacetime known at compile time (the common case)
  const bool sks = kSksCurved || pl.simulation_coord == BL_COORD_SKS;
  const bool ray_flat = !kSksCurved && st.ray_flat;

  // ---------------- coefficients (simulation_coefficients.cpp:274-455)
  const double rho = pr[0], pgas = pr[1];
  const double uu1 = pr[2], uu2 = pr[3], uu3 = pr[4];
  const double bb1 = pr[5], bb2 = pr[6], bb3 = pr[7];
  const double rho_cgs = rho * pl.d_unit;
  const double pgas_cgs = pgas * pl.e_unit;
  // The quotients below have operands that are products of single-precision grid values (zero, or
  // 1e-45 .. 3e38 in magnitude) and unit constants: far inside the range where bl_div_g() is the IEEE
  // quotient (bl_geometry.h). Quotients that involve results of exp() keep the plain division.
  const double n_cgs = bl_div_g(rho_cgs, pl.plasma_mu * kMp);
  const double n_e_cgs = bl_div_g(n_cgs, 1.0 + 1.0 / pl.plasma_ne_ni);

  // Velocity and field in simulation coordinates (:292-330). In SKS the metric is sparse
  // (radiation_geometry.cpp:462-489, :543-571); sums that the reference runs over all 16 entries
  // are evaluated over the non-zero ones, which is exact (the dropped terms are 0 * finite = +-0).
  double ucon_sim[4], bcon_sim[4], b_sq;
  if (sks) {
    const double cth2 = cth * cth;
    const double sth2 = 1.0 - cth2;
    const double delta = r2 - 2.0 * bh_m * r + a2;
    const double sigma = r2 + a2 * cth2;
    // three quotients over sigma (2 m r / sigma is one expression in the reference, used six times)
    const BlRecip rc_sigma = bl_recip(sigma);
    const double two_mr_sigma = bl_div_r(2.0 * bh_m * r, rc_sigma);
    const double g00 = -(1.0 - two_mr_sigma);
    const double g01 = two_mr_sigma;
    const double g03 = bl_div_r(-2.0 * bh_m * bh_a * r * sth2, rc_sigma);
    const double g11 = 1.0 + two_mr_sigma;
    const double g13 = -(1.0 + two_mr_sigma) * bh_a * sth2;
    const double g22 = sigma;
    const double g33 = (r2 + a2 + bl_div_r(2.0 * bh_m * a2 * r * sth2, rc_sigma)) * sth2;
    const double gc00 = -(1.0 + two_mr_sigma);
    const double gc01 = two_mr_sigma;
    (void)delta;
    // uu0 (:297-300): gcov_sim[1][2] = gcov_sim[2][3] = 0
    const double uu0 = bl_sqrt_g(1.0 + g11 * uu1 * uu1 + 2.0 * 0.0 * uu1 * uu2 + 2.0 * g13 * uu1 * uu3
        + g22 * uu2 * uu2 + 2.0 * 0.0 * uu2 * uu3 + g33 * uu3 * uu3);
    const double lapse = bl_div_g(1.0, bl_sqrt_g(-gc00));
    const double shift1 = bl_div_g(-gc01, gc00);
    // shift2 = shift3 = -gcon_sim[0][2,3] / gcon_sim[0][0] = -0 / gc00 = +0 (gc00 <= -1, or NaN and then
    // uu0 is NaN too), so the reference's  uu_a - shift_a * uu0 / lapse  subtracts (+0 * uu0) / lapse:
    // +0 when uu0 is finite (lapse is in (0, 1]), NaN when uu0 is not. 0.0 * uu0 is exactly that.
    const BlRecip rc_lapse = bl_recip(lapse);
    ucon_sim[0] = bl_div_r(uu0, rc_lapse);
    ucon_sim[1] = uu1 - bl_div_r(shift1 * uu0, rc_lapse);
    ucon_sim[2] = uu2 - 0.0 * uu0;
    ucon_sim[3] = uu3 - 0.0 * uu0;
    // ucov_sim[mu] = sum_nu gcov_sim[mu][nu] ucon_sim[nu] (:310-313), non-zero entries only
    const double ucov1 = (g01 * ucon_sim[0] + g11 * ucon_sim[1]) + g13 * ucon_sim[3];
    const double ucov2 = g22 * ucon_sim[2];
    const double ucov3 = (g03 * ucon_sim[0] + g13 * ucon_sim[1]) + g33 * ucon_sim[3];
    bcon_sim[0] = ucov1 * bb1 + ucov2 * bb2 + ucov3 * bb3;
    const BlRecip rc_u0 = bl_recip(ucon_sim[0]);
    bcon_sim[1] = bl_div_r(bb1 + bcon_sim[0] * ucon_sim[1], rc_u0);
    bcon_sim[2] = bl_div_r(bb2 + bcon_sim[0] * ucon_sim[2], rc_u0);
    bcon_sim[3] = bl_div_r(bb3 + bcon_sim[0] * ucon_sim[3], rc_u0);
    const double bcov0 = (g00 * bcon_sim[0] + g01 * bcon_sim[1]) + g03 * bcon_sim[3];
    const double bcov1 = (g01 * bcon_sim[0] + g11 * bcon_sim[1]) + g13 * bcon_sim[3];
    const double bcov2 = g22 * bcon_sim[2];
    const double bcov3 = (g03 * bcon_sim[0] + g13 * bcon_sim[1]) + g33 * bcon_sim[3];
    b_sq = bcov0 * bcon_sim[0] + bcov1 * bcon_sim[1] + bcov2 * bcon_sim[2] + bcov3 * bcon_sim[3];
  } else {
    // Cartesian Kerr-Schild simulation: same metric as the geodesic one, never flat
    double gs_cov[4][4], gs_con[4][4];
    bl_gcov_ks(ks, gs_cov);
    bl_gcon_ks(ks, gs_con);
    const double uu0 = bl_sqrt_g(1.0 + gs_cov[1][1] * uu1 * uu1 + 2.0 * gs_cov[1][2] * uu1 * uu2
        + 2.0 * gs_cov[1][3] * uu1 * uu3 + gs_cov[2][2] * uu2 * uu2 + 2.0 * gs_cov[2][3] * uu2 * uu3
        + gs_cov[3][3] * uu3 * uu3);
    const double lapse = 1.0 / bl_sqrt_g(-gs_con[0][0]);
    const double shift1 = -gs_con[0][1] / gs_con[0][0];
    const double shift2 = -gs_con[0][2] / gs_con[0][0];
    const double shift3 = -gs_con[0][3] / gs_con[0][0];
    ucon_sim[0] = uu0 / lapse;
    ucon_sim[1] = uu1 - shift1 * uu0 / lapse;
    ucon_sim[2] = uu2 - shift2 * uu0 / lapse;
    ucon_sim[3] = uu3 - shift3 * uu0 / lapse;
    double ucov_sim[4];
    for (int mu = 0; mu < 4; mu++) {
      double acc = 0.0;
      for (int nu = 0; nu < 4; nu++) acc += gs_cov[mu][nu] * ucon_sim[nu];
      ucov_sim[mu] = acc;
    }
    bcon_sim[0] = ucov_sim[1] * bb1 + ucov_sim[2] * bb2 + ucov_sim[3] * bb3;
    bcon_sim[1] = (bb1 + bcon_sim[0] * ucon_sim[1]) / ucon_sim[0];
    bcon_sim[2] = (bb2 + bcon_sim[0] * ucon_sim[2]) / ucon_sim[0];
    bcon_sim[3] = (bb3 + bcon_sim[0] * ucon_sim[3]) / ucon_sim[0];
    b_sq = 0.0;
    for (int mu = 0; mu < 4; mu++) {
      double acc = 0.0;
      for (int nu = 0; nu < 4; nu++) acc += gs_cov[mu][nu] * bcon_sim[nu];
      b_sq += acc * bcon_sim[mu];
    }
  }
  const double bb_cgs = bl_sqrt_g(b_sq) * pl.b_unit;
  const double sigma_cut = b_sq / rho;
  const double beta_inv = bl_div_g(b_sq, 2.0 * pgas);

  // electron temperature, T_i/T_e(beta) model (:333-348)
  double theta_e = __longlong_as_double(0x7ff8000000000000ll);
  double kb_tt_e_cgs = theta_e;
  if (kExtended && pl.code_kappa) {
    // electron entropy model (:351-358)
    if (pl.plasma_thermal_frac != 0.0) {
      const double kappa = kappa_f;
      const double mu_e = pl.plasma_mu * (1.0 + 1.0 / pl.plasma_ne_ni);
      const double rho_e = rho * kMe / (mu_e * kMp);
      const double rho_kappa_e_cbrt = bl_cbrt(rho_e * kappa);
      theta_e = 1.0 / 5.0 * (blm_sqrt(1.0 + 25.0 * rho_kappa_e_cbrt * rho_kappa_e_cbrt) - 1.0);
      kb_tt_e_cgs = theta_e * kMe * kC * kC;
    }
  } else if (pl.plasma_thermal_frac != 0.0) {
    double tti_tte = bl_div_g(pl.plasma_rat_high + pl.plasma_rat_low * beta_inv * beta_inv, 1.0 + beta_inv * beta_inv);
    double kb_tt_tot_cgs = bl_div_g(pl.plasma_mu * kMp * pgas_cgs, rho_cgs);
    if (pl.plasma_use_p) {
      kb_tt_e_cgs = bl_div_g(1.0 + pl.plasma_ne_ni, tti_tte + pl.plasma_ne_ni) * kb_tt_tot_cgs;
    } else {
      kb_tt_e_cgs = (1.0 + pl.plasma_ne_ni) * kb_tt_tot_cgs / (P.cold->plasma_gamma - 1.0);
      kb_tt_e_cgs /= tti_tte / (P.cold->plasma_gamma_i - 1.0) + pl.plasma_ne_ni / (P.cold->plasma_gamma_e - 1.0);
    }
    theta_e = bl_div_g(kb_tt_e_cgs, kMe * kC * kC);
  }

  // cell cuts (:361-375); all thresholds negative = disabled is the common case
  bool cell_cut = false;
  if (pl.any_cell_cut) {
    const BlShadeCold &cc = *P.cold;
    // disabled thresholds are -inf / +inf (bl_api.hip): one compare each, same decisions as "cut >= 0 and ..."
    cell_cut = rho_cgs < cc.cut_rho_min || rho_cgs > cc.cut_rho_max || n_e_cgs < cc.cut_n_e_min || n_e_cgs > cc.cut_n_e_max
        || pgas_cgs < cc.cut_p_gas_min || pgas_cgs > cc.cut_p_gas_max || theta_e < cc.cut_theta_e_min || theta_e > cc.cut_theta_e_max
        || bb_cgs < cc.cut_b_min || bb_cgs > cc.cut_b_max || sigma_cut < cc.cut_sigma_min || sigma_cut > cc.cut_sigma_max
        || beta_inv < cc.cut_beta_inverse_min || beta_inv > cc.cut_beta_inverse_max;
  }
  const bool no_field = bb1 == 0.0 && bb2 == 0.0 && bb3 == 0.0;   // :394
  out->have_coefficients = false;
  out->have_cell = false;
  if (cell_cut) return;
  out->have_cell = true;   // :377-387 (used in auxiliary-image mode only)
  out->cell[0] = rho_cgs;
  out->cell[1] = n_e_cgs;
  out->cell[2] = pgas_cgs;
  out->cell[3] = theta_e;
  out->cell[4] = bb_cgs;
  out->cell[5] = sigma_cut;
  out->cell[6] = beta_inv;
  if (need_coefficients == 0 || no_field) return;   // :389-395

  // Transform u and b to geodesic (CKS) coordinates (:398-408). The Jacobian of
  // radiation_geometry.cpp:69-126 has row/column 0 = identity and jacobian[3][3] = 0; the products
  // with those 0 / 1 entries are dropped (exact).
  double ucon[4], bcon[4];
  if (sks) {
    const double sth = bl_sqrt_g(1.0 - cth * cth);
    double sph, cph;
    bl_sincos(ph_unwrapped, &sph, &cph);
    const double j11 = sth * cph;
    const double j12 = cth * (r * cph - bh_a * sph);
    const double j13 = sth * (-r * sph - bh_a * cph);
    const double j21 = sth * sph;
    const double j22 = cth * (r * sph + bh_a * cph);
    const double j23 = sth * (r * cph - bh_a * sph);
    const double j31 = cth;
    const double j32 = -r * sth;
    ucon[0] = ucon_sim[0];
    ucon[1] = (j11 * ucon_sim[1] + j12 * ucon_sim[2]) + j13 * ucon_sim[3];
    ucon[2] = (j21 * ucon_sim[1] + j22 * ucon_sim[2]) + j23 * ucon_sim[3];
    ucon[3] = j31 * ucon_sim[1] + j32 * ucon_sim[2];
    bcon[0] = bcon_sim[0];
    bcon[1] = (j11 * bcon_sim[1] + j12 * bcon_sim[2]) + j13 * bcon_sim[3];
    bcon[2] = (j21 * bcon_sim[1] + j22 * bcon_sim[2]) + j23 * bcon_sim[3];
    bcon[3] = j31 * bcon_sim[1] + j32 * bcon_sim[2];
  } else {
    for (int mu = 0; mu < 4; mu++) {
      ucon[mu] = ucon_sim[mu];
      bcon[mu] = bcon_sim[mu];
    }
  }
  // kcon, ucov, bcov in the geodesic metric (:411-428); the metric is rebuilt from the Kerr-Schild
  // scalars here (a dozen multiplies) rather than kept live across the gather
  double gcov[4][4], gcon[4][4];
  if (ray_flat) {
    bl_minkowski(gcov);
    bl_minkowski(gcon);
  } else {
    bl_gcov_ks(ks, gcov);
    bl_gcon_ks(ks, gcon);
  }
  double kcon[4], ucov[4], bcov[4];
  for (int mu = 0; mu < 4; mu++) {
    double ak = 0.0, au = 0.0, ab = 0.0;
    for (int nu = 0; nu < 4; nu++) {
      ak += gcon[mu][nu] * kcov[nu];
      au += gcov[mu][nu] * ucon[nu];
      ab += gcov[mu][nu] * bcon[nu];
    }
    kcon[mu] = ak;
    ucov[mu] = au;
    bcov[mu] = ab;
  }
  double tetrad[4][4];
  tetrad_build(ucon, ucov, kcon, kcov, bcon, gcov, gcon, tetrad);
  double k_tet[3] = {0.0, 0.0, 0.0}, b_tet[3] = {0.0, 0.0, 0.0};   // :434-455
  for (int mu = 0; mu < 4; mu++)
    for (int a = 0; a < 3; a++) {
      k_tet[a] += tetrad[a + 1][mu] * kcov[mu];
      b_tet[a] += tetrad[a + 1][mu] * bcov[mu];
    }
  const double k_sq_tet = k_tet[0] * k_tet[0] + k_tet[1] * k_tet[1] + k_tet[2] * k_tet[2];
  const double b_sq_tet = b_tet[0] * b_tet[0] + b_tet[1] * b_tet[1] + b_tet[2] * b_tet[2];
  const double k_b_tet = k_tet[0] * b_tet[0] + k_tet[1] * b_tet[1] + k_tet[2] * b_tet[2];
  const double cos2_theta_b = std_min(bl_div_g(k_b_tet * k_b_tet, k_sq_tet * b_sq_tet), 1.0);
  const double sin2_theta_b = 1.0 - cos2_theta_b;
  double nu_sum = 0.0;   // :461-463
  for (int mu = 0; mu < 4; mu++) nu_sum -= kcov[mu] * ucon[mu];
  out->have_coefficients = true;
  out->nu_fluid_over_nu = nu_sum;
  out->n_e_cgs = n_e_cgs;
  out->nu_c_cgs = bl_div_g(kE * bb_cgs, 2.0 * kPi * kMe * kC);
  out->theta_e = theta_e;
  out->sin_theta_b = bl_sqrt_g(sin2_theta_b);
  out->kb_tt_e_cgs = kb_tt_e_cgs;
  out->sin2_theta_b = sin2_theta_b;
  out->cos2_theta_b = cos2_theta_b;
  out->cos_sign = k_b_tet >= 0.0 ? 1.0 : -1.0;
  out->cos_theta_b = bl_sqrt_g(cos2_theta_b) * (k_b_tet >= 0.0 ? 1.0 : -1.0);   // :455
  if (kExtended && pol_out != nullptr) {
    // polarized.cpp:163-265 rebuilds k^mu and this tetrad from the same sampled values: hand them over
    for (int mu = 0; mu < 4; mu++) {
      pol_out->kcon[mu] = kcon[mu];
      pol_out->e1[mu] = tetrad[1][mu];
      pol_out->e2[mu] = tetrad[2][mu];
    }
  }
}

// exact arithmetic tier of the coefficient formulas
#define BLC_NAME(f) f
#define BLC_SQRT bl_sqrt_g
#define BLC_SQRT_M blm_sqrt
#define BLC_CBRT bl_cbrt
#define BLC_EXP bl_exp
#define BLC_EXPM1 bl_expm1
#define BLC_LOG bl_log
#define BLC_POW bl_pow
#define BLC_POWBASE_T blm_powbase
#define BLC_POW_BASE bl_pow_base
#define BLC_POW_OF bl_pow_of
#define BLC_DIV_G bl_div_g
#define BLC_SIN bl_sin
#define BLC_COS bl_cos
#define BLC_TANH bl_tanh
#include "bl_coefficients.inc"
#undef BLC_NAME
#undef BLC_SQRT
#undef BLC_SQRT_M
#undef BLC_CBRT
#undef BLC_EXP
#undef BLC_EXPM1
#undef BLC_LOG
#undef BLC_POW
#undef BLC_POWBASE_T
#undef BLC_POW_BASE
#undef BLC_POW_OF
#undef BLC_DIV_G
#undef BLC_SIN
#undef BLC_COS
#undef BLC_TANH

// Formula mode, one sample (formula_coefficients.cpp:118-161)
__device__ __forceinline__ void shade_formula(const BlShadeArgs &P, const BlSpacetime &st, double r, double x1,
                                              double x2, double x3, SampleShade *out) {
  const BlFormulaDevice &fm = P.formula;
  const double bh_a = st.bh_a, bh_m = st.bh_m;
  double rr = blm_sqrt(r * r - x3 * x3);
  double cth = x3 / r;
  double sth = blm_sqrt(1.0 - cth * cth);
  double ph = bl_atan2(x2, x1) - bl_atan(bh_a / r);
  double sph, cph;
  bl_sincos(ph, &sph, &cph);
  double delta = r * r - 2.0 * bh_m * r + bh_a * bh_a;
  double sigma = r * r + bh_a * bh_a * cth * cth;
  double gtt_bl = -(1.0 + 2.0 * bh_m * r * (r * r + bh_a * bh_a) / (delta * sigma));
  double gtph_bl = -2.0 * bh_m * bh_a * r / (delta * sigma);
  double grr_bl = delta / sigma;
  double gthth_bl = 1.0 / sigma;
  double gphph_bl = (sigma - 2.0 * bh_m * r) / (delta * sigma * sth * sth);
  double ll = fm.l0 / (1.0 + rr) * bl_pow(rr, 1.0 + fm.q);
  double u_norm = 1.0 / blm_sqrt(-gtt_bl + 2.0 * gtph_bl * ll - gphph_bl * ll * ll);
  double u_t_bl = -u_norm;
  double u_r_bl = 0.0;
  double u_th_bl = 0.0;
  double u_ph_bl = u_norm * ll;
  double ut_bl = gtt_bl * u_t_bl + gtph_bl * u_ph_bl;
  double ur_bl = grr_bl * u_r_bl;
  double uth_bl = gthth_bl * u_th_bl;
  double uph_bl = gtph_bl * u_t_bl + gphph_bl * u_ph_bl;
  double ut = ut_bl + 2.0 * bh_m * r / delta * ur_bl;
  double ur = ur_bl;
  double uth = uth_bl;
  double uph = uph_bl + bh_a / delta * ur_bl;
  out->fu[0] = ut;
  out->fu[1] = sth * cph * ur + cth * (r * cph - bh_a * sph) * uth + sth * (-r * sph - bh_a * cph) * uph;
  out->fu[2] = sth * sph * ur + cth * (r * sph + bh_a * cph) * uth + sth * (r * cph - bh_a * sph) * uph;
  out->fu[3] = cth * ur - r * sth * uth;
  out->n_n0_fluid = bl_exp(-0.5 * (r * r / (fm.r0 * fm.r0) + fm.h * fm.h * cth * cth));
  out->have_coefficients = true;
}

}  // namespace

// ---- locating a sample in the common case: one grid (or equal blocks merged into one) in spherical Kerr-Schild coordinates
// with its coordinate tables in LDS, trilinear sampling, no optional geometric cut, no slow light. The same functions of the
// same values as locate_sample() - bit-identical results - as one straight line: every lane runs the whole search (a dead
// slot, a cut or an off-grid sample on clamped inputs) and the status is selected at the end, where the general kernel nests
// a dozen divergent branches whose masks, merges and live scalars cost it more instructions than the arithmetic they skip
// (555 vector instructions per sample there, 371 here).
struct PlainGrid {   // what the search needs of the grid, fetched once per workgroup
  GridTables tab;
  int n_i, n_j, n_k, nb_i, nb_j, nb_k;
  bool one_block;
  const double *inv_w[3];   // tolerant tier: 1 / (xv[c + 1] - xv[c]) per axis (LDS), or null: the fractions are IEEE quotients
};
__device__ __forceinline__ void stage_grid_tables(const BlGridDevice &g, double *lds, PlainGrid *pg) {
  double *dst = lds;
  for (int a = 0; a < 3; a++) {
    pg->tab.xf[a] = dst;
    for (int i = threadIdx.x; i <= g.n[a]; i += blockDim.x) dst[i] = g.xf[a][i];
    dst += g.n[a] + 1;
    pg->tab.xv[a] = dst;
    for (int i = threadIdx.x; i < g.n[a]; i += blockDim.x) dst[i] = g.xv[a][i];
    dst += g.n[a];
  }
  unsigned short *bdst = reinterpret_cast<unsigned short *>(dst);
  for (int a = 0; a < 3; a++) {
    pg->tab.bucket[a] = bdst;
    for (int i = threadIdx.x; i < g.n_bucket[a]; i += blockDim.x) bdst[i] = g.bucket[a][i];
    bdst += g.n_bucket[a];
  }
  pg->n_i = g.n[0]; pg->n_j = g.n[1]; pg->n_k = g.n[2];
  pg->nb_i = g.nb[0]; pg->nb_j = g.nb[1]; pg->nb_k = g.nb[2];
  pg->one_block = g.nb[0] == g.n[0] && g.nb[1] == g.n[1] && g.nb[2] == g.n[2];
  pg->inv_w[0] = pg->inv_w[1] = pg->inv_w[2] = nullptr;
}
// ... and the reciprocal widths between cell centres behind them (call after a barrier: reads the staged centres)
__device__ __forceinline__ void stage_reciprocal_widths(const BlGridDevice &g, double *lds, PlainGrid *pg) {
  double *dst = lds;
  for (int a = 0; a < 3; a++) {
    pg->inv_w[a] = dst;
    for (int i = threadIdx.x; i + 1 < g.n[a]; i += blockDim.x) dst[i] = 1.0 / (pg->tab.xv[a][i + 1] - pg->tab.xv[a][i]);
    dst += g.n[a];
  }
}
struct PlainLocated {
  double f_i, f_j, f_k, ph_unwrapped;
  uint32_t status, cell;   // status: kSample...; | kPlainUndecided from locate_plain_sample_tolerant()
};
constexpr uint32_t kPlainUndecided = 0x100u;
// From (r, theta, unwrapped phi) to status, cell and fractions. margin (if asked for): how far theta and phi are from the nearest
// value they are compared with on the way - the faces and the centre of their cells, the ends of the azimuth's range.
// guess_mask (tolerant tier): axes whose cell is guessed as floor((x - x0) / width) instead of searched (evenly spaced faces); a guess
// the faces do not confirm comes back as margin = 0.
__device__ __forceinline__ PlainLocated locate_plain_from_angles(const BlGridDevice &g, const PlainGrid &pg, bool live, bool cut, double r, double th,
                                                                 double ph_unwrapped, double *margin, int guess_mask = 0) {
  const GridTables &tab = pg.tab;
  double ph = ph_unwrapped;
  ph += ph < 0.0 ? 2.0 * kPi : 0.0;
  const double ph_once = ph;
  ph -= ph >= 2.0 * kPi ? 2.0 * kPi : 0.0;
  const double s1 = r, s2 = th, s3 = ph;
  const bool off_grid = s1 < tab.xf[0][0] || s1 > tab.xf[0][pg.n_i] || s2 < tab.xf[1][0] || s2 > tab.xf[1][pg.n_j]
      || s3 < tab.xf[2][0] || s3 > tab.xf[2][pg.n_k];             // :352-394
  const int i = find_cell(g, tab, 0, s1);
  int j, k;
  if (guess_mask & 2) {
    j = (int)((s2 - g.cell_x0[1]) * g.cell_inv_w[1]);
    j = j < 0 ? 0 : (j > pg.n_j - 1 ? pg.n_j - 1 : j);
  } else {
    j = find_cell(g, tab, 1, s2);
  }
  if (guess_mask & 4) {
    k = (int)((s3 - g.cell_x0[2]) * g.cell_inv_w[2]);
    k = k < 0 ? 0 : (k > pg.n_k - 1 ? pg.n_k - 1 : k);
  } else {
    k = find_cell(g, tab, 2, s3);
  }
  // :485-490, per block of a merged grid (one block - the usual case - needs no remainders)
  // (a wave-uniform branch, not a select: three integer remainders are ~120 vector instructions the one-block case never needs)
  int i_b = i, j_b = j, k_b = k;
  if (__builtin_expect(!pg.one_block, 0)) {
    i_b = i % pg.nb_i;
    j_b = j % pg.nb_j;
    k_b = k % pg.nb_k;
  }
  const double xv_at_j = tab.xv[1][j], xv_at_k = tab.xv[2][k];
  const int i_m = (i_b == 0 || (i_b != pg.nb_i - 1 && s1 >= tab.xv[0][i])) ? i : i - 1;
  const int j_m = (j_b == 0 || (j_b != pg.nb_j - 1 && s2 >= xv_at_j)) ? j : j - 1;
  const int k_m = (k_b == 0 || (k_b != pg.nb_k - 1 && s3 >= xv_at_k)) ? k : k - 1;
  const double xv_i = tab.xv[0][i_m], xv_j = tab.xv[1][j_m], xv_k = tab.xv[2][k_m];
  const bool sampled = live && !cut && !off_grid;
  PlainLocated out;
  // (fractions of cell widths: ordinary operands for the short division)
  double f_i, f_j, f_k;
  if (pg.inv_w[0] != nullptr) {   // tolerant tier: one multiplication by the width's reciprocal (2e-16 of the fraction)
    f_i = (s1 - xv_i) * pg.inv_w[0][i_m];
    f_j = (s2 - xv_j) * pg.inv_w[1][j_m];
    f_k = (s3 - xv_k) * pg.inv_w[2][k_m];
  } else {
    f_i = blm_div(s1 - xv_i, tab.xv[0][i_m + 1] - xv_i);
    f_j = blm_div(s2 - xv_j, tab.xv[1][j_m + 1] - xv_j);
    f_k = blm_div(s3 - xv_k, tab.xv[2][k_m + 1] - xv_k);
  }
  out.f_i = sampled ? f_i : 0.0;
  out.f_j = sampled ? f_j : 0.0;
  out.f_k = sampled ? f_k : 0.0;
  out.ph_unwrapped = (!live || cut) ? 0.0 : ph_unwrapped;
  out.status = !live ? (uint32_t)kSampleNone : (cut ? (uint32_t)kSampleCut : (off_grid ? (uint32_t)kSampleOffGrid : (uint32_t)kSampleInterp));
  out.cell = sampled ? (uint32_t)((k_m * pg.n_j + j_m) * pg.n_i + i_m) : 0u;
  if (margin != nullptr) {
    const double fj0 = tab.xf[1][j], fj1 = tab.xf[1][j + 1], fk0 = tab.xf[2][k], fk1 = tab.xf[2][k + 1];
    double m = std_min(blm_abs(s2 - fj0), blm_abs(s2 - fj1));
    m = std_min(m, blm_abs(s2 - xv_at_j));
    m = std_min(m, std_min(blm_abs(s3 - fk0), blm_abs(s3 - fk1)));
    m = std_min(m, blm_abs(s3 - xv_at_k));
    m = std_min(m, std_min(blm_abs(ph_unwrapped), blm_abs(ph_once - 2.0 * kPi)));
    // a guessed cell has to hold the coordinate (the search's "first upper face >= x" then names the same cell: the margin above
    // keeps x off the faces); off the grid the guess is the clamped end cell, as the search's
    if ((guess_mask & 2) && !off_grid && !(s2 >= fj0 && s2 <= fj1)) m = 0.0;
    if ((guess_mask & 4) && !off_grid && !(s3 >= fk0 && s3 <= fk1)) m = 0.0;
    *margin = m;
  }
  return out;
}
template <bool kSpinZero>
__device__ __forceinline__ PlainLocated locate_plain_sample(const BlSpacetime &st, const BlGridDevice &g, const PlainGrid &pg, double camera_r,
                                                            bool live, double x1, double x2, double x3) {
  // a dead slot may hold anything: the search runs on a harmless point instead
  x1 = live ? x1 : 1.0;
  x2 = live ? x2 : 1.0;
  x3 = live ? x3 : 1.0;
  double r2;
  const double r = bl_radial_coordinate2<kSpinZero>(st, x1, x2, x3, &r2);
  const bool cut = r > camera_r;                                   // simulation_sampling.cpp:238-243
  // ConvertFromCKS (radiation_geometry.cpp:37-57), as in locate_sample()
  const double th = bl_acos(blm_div(x3, r));
  const double ph_unwrapped = kSpinZero ? bl_atan2(x2, x1) : bl_atan2(x2, x1) - bl_atan(blm_div(st.bh_a, r));
  return locate_plain_from_angles(g, pg, live, cut, r, th, ph_unwrapped, nullptr);
}


// The loads of one sample: its located sample (what the gather needs) and its record halves (what the arithmetic needs)
struct FastLocated {
  double2 l0, l1;
  unsigned long long tag;
};
struct FastRay {
  double2 q0, q1, q2, q3;
};
__device__ __forceinline__ void fast_load_located(const BlShadeArgs &P, unsigned long long idx, FastLocated &r) {
  // f_i, f_j | f_k, tag: the locate kernel writes the tag where the exact tier keeps the azimuth (BlShadeArgs::tag_in_record)
  const double2 *loc = reinterpret_cast<const double2 *>(P.located + idx);
  r.l0 = loc[0];
  r.l1 = loc[1];
  r.tag = (unsigned long long)__double_as_longlong(r.l1.y);
}
__device__ __forceinline__ void fast_load_ray(const BlShadeArgs &P, unsigned long long idx, FastRay &r) {
  const double2 *hot = reinterpret_cast<const double2 *>(P.records_hot + (idx) * P.record_stride);
  const double2 *cold = reinterpret_cast<const double2 *>(P.records_cold + (idx) * P.record_stride);
  r.q0 = hot[0]; r.q1 = hot[1]; r.q2 = cold[0]; r.q3 = cold[1];
}


// The BlAuxSample of a sample in auxiliary-image mode (unpolarized.cpp:113-173): what bl_transfer_aux_kernel integrates besides
// (j, alpha). Shared by bl_shade_kernel and bl_shade_polarized2_kernel.
__device__ __forceinline__ void write_aux_record(const BlShadeArgs &P, const BlSpacetime &st, const BlKerrSchild &ks, unsigned long long idx_cur, size_t row,
                                                 const SampleShade &sh, const double kcov[4], double x1, double x2, double x3, double delta_lambda) {
  BlAuxSample aux;
  aux.delta_lambda = delta_lambda;
  aux.t = P.sample_t != nullptr ? P.sample_t[idx_cur] : 0.0;
  aux.plane = P.cam_x[1] * x1 + P.cam_x[2] * x2 + P.cam_x[3] * x3;
  aux.length_term = 0.0;
  aux.pad = 0.0;
  if (P.aux_need_length) {
    // unpolarized.cpp:115-129 with the renormalised sample momentum
    double gcov[4][4], gcon[4][4];
    if (st.ray_flat) {
      bl_minkowski(gcov);
      bl_minkowski(gcon);
    } else {
      bl_gcov_ks(ks, gcov);
      bl_gcon_ks(ks, gcon);
    }
    double temp_a[4] = {0.0, 0.0, 0.0, 0.0};
    for (int a = 1; a < 4; a++)
      for (int mu = 0; mu < 4; mu++) temp_a[a] += (gcon[a][mu] - gcon[0][a] * gcon[0][mu] / gcon[0][0]) * kcov[mu];
    double dl_dlambda_sq = 0.0;
    for (int a = 1; a < 4; a++)
      for (int b = 1; b < 4; b++) dl_dlambda_sq += gcov[a][b] * temp_a[a] * temp_a[b];
    aux.length_term = blm_sqrt(dl_dlambda_sq) * delta_lambda * P.x_unit;
  }
  const double nan = __longlong_as_double(0x7ff8000000000000ll);
  for (int a = 0; a < BL_NUM_CELL_VALUES; a++) aux.cell[a] = sh.have_cell ? sh.cell[a] : nan;
  P.aux[row] = aux;
}

// What a polarized run keeps of a sample besides the frame sample_finish_simulation() has written into its BlPolSample: position and
// length there, and the 64-byte BlCoefInputs of the record for bl_polarized_coefficients_kernel - the per-frequency formulas
// (Bessel functions, a dozen powers and exponentials) need few registers and many waves, that kernel evaluates them from these scalars.
__device__ __forceinline__ void write_polarized_inputs(const BlShadeArgs &P, unsigned long long idx_cur, size_t row, const SampleShade &sh, const double kcov[4],
                                                       const float pr[8], double x1, double x2, double x3, double delta_lambda) {
  BlPolSample *ps = P.pol_samples + row;
  ps->x[0] = x1; ps->x[1] = x2; ps->x[2] = x3;
  ps->delta_lambda = delta_lambda;
  BlCoefInputs ci;
  if (sh.have_coefficients) {
    ci.nu_fluid_over_nu = sh.nu_fluid_over_nu;
    ci.n_e_cgs = sh.n_e_cgs;
    ci.nu_c_cgs = sh.nu_c_cgs;
    ci.theta_e = sh.theta_e;
    ci.kb_tt_e_cgs = sh.kb_tt_e_cgs;
    ci.cos2_theta_b = sh.cos2_theta_b;
    ci.cos_sign = sh.cos_sign;
    ci.have_coefficients = 1.0;
  } else {
    // cut samples, cells cut or without field: the coefficient code never reached its tetrad, but the polarized transfer
    // needs the frame (with zero velocity / field where the sample was cut). Rare, and a frame's worth of registers:
    // bl_polarized_frame_kernel builds it from what is parked here in the fields nobody reads for such a sample -
    // the renormalised k_mu and the sampled velocity and field.
    ci.nu_fluid_over_nu = kcov[0];
    ci.n_e_cgs = kcov[1];
    ci.nu_c_cgs = kcov[2];
    ci.theta_e = kcov[3];
    ci.kb_tt_e_cgs = __hiloint2double(__float_as_int(pr[3]), __float_as_int(pr[2]));
    ci.cos2_theta_b = __hiloint2double(__float_as_int(pr[5]), __float_as_int(pr[4]));
    ci.cos_sign = __hiloint2double(__float_as_int(pr[7]), __float_as_int(pr[6]));
    ci.have_coefficients = 0.0;
    // ... and listed for that kernel (the list of the tolerant tier's deferred records, unused in polarized runs): one
    // atomic per wave for the lanes that are here; a full list makes the frame kernel scan every record instead
    if (P.redo_list != nullptr) {
      const unsigned long long here = __ballot(1);
      const unsigned int rank = __builtin_amdgcn_mbcnt_hi((unsigned int)(here >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)here, 0u));
      unsigned long long first = 0ull;
      if (rank == 0u) first = atomicAdd(&P.counters[BL_CNT_REDO], (unsigned long long)__popcll(here));
      const unsigned long long at = (((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(first >> 32)) << 32)
                                     | (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)first)) + rank;
      if (at < P.redo_capacity) P.redo_list[at] = idx_cur;
    }
  }
  P.coef_inputs[idx_cur] = ci;
}

// The corner cells of a located sample, requested (gather_issue) one sample ahead of their use (gather_finish): the two
// halves of sample_primitives(), same operations in the same order. Every load is unconditional - a sample without
// cells to read (cut, off the grid, dead slot) reads cell 0, a nearest-cell sample reads its cell sixteen times - so
// that the number of loads in flight is the same on every path and the compiler's s_waitcnt for an OLDER load (the
// per-ray constants) does not have to wait for these (it counts conservatively across branches).
__device__ __forceinline__ void gather_issue(const BlShadeArgs &P, int status, uint32_t cell, float4 (&lo)[8], float4 (&hi)[8]) {
  const BlGridDevice &g = P.grid;
  const bool interp = status == kSampleInterp;
  const size_t first = (interp || status == kSampleNearest) ? (size_t)cell : 0;
  const float4 *base = reinterpret_cast<const float4 *>(g.cells) + first * 2;
  const size_t row = interp ? (size_t)g.stride_row * 2 : 0, plane = interp ? (size_t)g.stride_plane * 2 : 0, next = interp ? 2 : 0;
#pragma unroll
  for (int corner = 0; corner < 8; corner++) {
    const float4 *p = base + (corner >> 2) * plane + ((corner >> 1) & 1) * row + (corner & 1) * next;
    lo[corner] = p[0];
    hi[corner] = p[1];
  }
}
// Slow light through the pipelined corner reads (bl_shade_fast_kernel): gather_issue() from the cell array of one time slice, and the
// eight values of that slice from the cells as they arrived - sample_slice_values()'s sums in its order, without fused
// multiply-adds: the exact tier's bits - then the blend in time of sample_primitives_slow()
__device__ __forceinline__ void gather_issue_slice(const float *slice_cells, const BlGridDevice &g, int status, uint32_t cell, float4 (&lo)[8], float4 (&hi)[8]) {
  const bool interp = status == kSampleInterp;
  const size_t first = (interp || status == kSampleNearest) ? (size_t)cell : 0;
  const float4 *base = reinterpret_cast<const float4 *>(slice_cells) + first * 2;
  const size_t row = interp ? (size_t)g.stride_row * 2 : 0, plane = interp ? (size_t)g.stride_plane * 2 : 0, next = interp ? 2 : 0;
#pragma unroll
  for (int corner = 0; corner < 8; corner++) {
    const float4 *p = base + (corner >> 2) * plane + ((corner >> 1) & 1) * row + (corner & 1) * next;
    lo[corner] = p[0];
    hi[corner] = p[1];
  }
}
// (straight-line: both kinds of sample take the sums, the nearest-cell sample then takes its cell - a branch around them puts the array
// in memory)
__device__ __forceinline__ void slice_values_from_cells(int status, const float4 (&lo)[8], const float4 (&hi)[8], double f_i, double f_j, double f_k, double val[8]) {
#pragma clang fp contract(off)
  const bool nearest = status == kSampleNearest;
  const double w_k[2] = {1.0 - f_k, f_k}, w_j[2] = {1.0 - f_j, f_j}, w_i[2] = {1.0 - f_i, f_i};
  double w[8];
  float c[8][8];
#pragma unroll
  for (int corner = 0; corner < 8; corner++) {
    w[corner] = w_k[corner >> 2] * w_j[(corner >> 1) & 1] * w_i[corner & 1];
    unpack_cell(lo[corner], hi[corner], c[corner]);
  }
#pragma unroll
  for (int q = 0; q < 8; q++) {
    double sum = w[0] * (double)c[0][q];
#pragma unroll
    for (int corner = 1; corner < 8; corner++) sum += w[corner] * (double)c[corner][q];
    if (q < 2) sum = sum <= 0.0 ? (double)c[0][q] : sum;
    val[q] = nearest ? (double)c[0][q] : sum;
  }
}
// ... the later slice's values from its cells, blended into the earlier slice's one quantity at a time (every sum in
// sample_slice_values()'s order)
__device__ __forceinline__ void slice_blend_from_cells(int status, const float4 (&lo)[8], const float4 (&hi)[8], double f_i, double f_j, double f_k, double t_frac,
                                                       double val[8]) {
#pragma clang fp contract(off)
  double next[8];
  slice_values_from_cells(status, lo, hi, f_i, f_j, f_k, next);
#pragma unroll
  for (int q = 0; q < 8; q++) val[q] = (1.0 - t_frac) * val[q] + t_frac * next[q];
}

// Inter-block interpolation: the eight anchor cells the locate kernel named (BlShadeArgs::anchors), requested the same way
struct FastAnchors {
  uint4 lo, hi;
};
__device__ __forceinline__ void fast_load_anchors(const BlShadeArgs &P, unsigned long long idx, FastAnchors &r) {
  const uint4 *p = reinterpret_cast<const uint4 *>(P.anchors + idx * 8);
  r.lo = p[0];
  r.hi = p[1];
}
__device__ __forceinline__ void gather_issue_anchors(const BlShadeArgs &P, int status, uint32_t cell_of_tag, const FastAnchors &anchors, float4 (&lo)[8],
                                                     float4 (&hi)[8]) {
  // a sample whose anchors all lie in its own block comes as an ordinary trilinear one (locate_sample_refined): corners from its cell
  const float4 *cells = reinterpret_cast<const float4 *>(P.grid.cells);
  const bool advanced = status == (int)kSampleAdvanced, interp = status == (int)kSampleInterp;
  const unsigned int listed[8] = {anchors.lo.x, anchors.lo.y, anchors.lo.z, anchors.lo.w, anchors.hi.x, anchors.hi.y, anchors.hi.z, anchors.hi.w};
  const unsigned int row = (unsigned int)P.grid.stride_row, plane = (unsigned int)P.grid.stride_plane;
#pragma unroll
  for (int corner = 0; corner < 8; corner++) {
    const unsigned int own = cell_of_tag + ((corner >> 2) ? plane : 0u) + (((corner >> 1) & 1) ? row : 0u) + (unsigned int)(corner & 1);
    const float4 *p = cells + (advanced ? (size_t)listed[corner] * 2 : (interp ? (size_t)own * 2 : 0));
    lo[corner] = p[0];
    hi[corner] = p[1];
  }
}
__device__ __forceinline__ void gather_finish(const BlShadeArgs &P, float fallback_rho, float fallback_pgas, int status, const float4 (&lo)[8],
                                              const float4 (&hi)[8], double f_i, double f_j, double f_k, float pr[8]) {
#pragma clang fp contract(off)
  const BlPlasmaDevice &pl = P.plasma;
  if (status == kSampleInterp) {
    // InterpolateSimple (simulation_sampling.cpp:1334-1351): w_c * v_c summed in the order mmm, mmp, mpm, mpp, pmm, ...
    const double w_k[2] = {1.0 - f_k, f_k}, w_j[2] = {1.0 - f_j, f_j}, w_i[2] = {1.0 - f_i, f_i};
    double val[8];
    float first[8];
#pragma unroll
    for (int corner = 0; corner < 8; corner++) {
      float v[8];
      unpack_cell(lo[corner], hi[corner], v);
      const double w = w_k[corner >> 2] * w_j[(corner >> 1) & 1] * w_i[corner & 1];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        if (corner == 0) {
          val[q] = w * (double)v[q];
          first[q] = v[q];
        } else {
          val[q] += w * (double)v[q];
        }
      }
    }
    if (val[0] <= 0.0) val[0] = (double)first[0];   // :822-825
    if (val[1] <= 0.0) val[1] = (double)first[1];
#pragma unroll
    for (int q = 0; q < 8; q++) pr[q] = (float)val[q];   // :830-839
  } else if (status == kSampleNearest) {
    unpack_cell(lo[0], hi[0], pr);
  } else if (status == kSampleOffGrid) {
    const float fnan = __int_as_float(0x7fc00000);
    pr[0] = pl.fallback_nan ? fnan : fallback_rho;    // :377-384, :678-706
    pr[1] = pl.fallback_nan ? fnan : fallback_pgas;
    for (int q = 2; q < 8; q++) pr[q] = pl.fallback_nan ? fnan : 0.0f;
  } else {
    for (int q = 0; q < 8; q++) pr[q] = 0.0f;
  }
}


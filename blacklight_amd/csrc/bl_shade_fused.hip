// bl_shade_fused.hip - the benchmark's coefficient kernel (gfx950): tolerant arithmetic tier, locate step inside, ONE frequency,
// one grid block whose faces are evenly spaced in log r, theta and phi and cover the whole sphere (what Athena++ / the mock write
// for a spherical Kerr-Schild run). Everything else the tier covers stays on a locate kernel + bl_shade_fast_kernel
// (bl_shade_fast.hip), whose arithmetic this kernel restates; what is different here is everything around the arithmetic:
//
//   * the loop body is one straight line. Every lane runs the whole sample - a dead slot, a cut or an off-grid sample on harmless
//     inputs - and the record is selected at the end; the only branches left are wave-uniform ones around things that almost never
//     happen (cell-cut thresholds that are switched off, a step that is not optically thin, a deferred sample). The general
//     kernel's nest of divergent branches cost more scalar and mask instructions than the arithmetic they skipped, and made the
//     compiler spill 109 scalar registers through v_writelane / v_readlane (round-3 ISA mix: 36 % of the loop's vector
//     instructions were fp64 arithmetic).
//   * what a cell search needs of an axis is ONE 64-byte row in LDS per cell (faces, centre, the centre and reciprocal width the
//     fraction is taken against on either side, the anchor shift), at a fixed place: one address computation and three 16-byte
//     reads per axis instead of five 8-byte reads from five tables with five address computations.
//   * cells are guessed arithmetically on every axis - floor((x - x0) / w) in theta and phi, floor((log2 r - l0) / w) with the
//     hardware's single-precision log2 in r - and confirmed against the row's faces: theta and phi with the guard band of the
//     tier's own angles, r exactly (radius, cut at the camera's sphere and radial cell are the exact tier's decisions, as before).
//     A guess its faces do not confirm is left to the exact kernel's second pass, like every other undecided sample.
//   * grid reads and per-ray constants are addressed by 32-bit offsets from scalar base pointers (a cell index x 32 bytes and a
//     ray slot x 8 bytes fit 32 bits), records by one running 64-bit pointer: no 64-bit address arithmetic per load.
//   * rarely used arguments are read from the kernel-argument segment where they are used (through a pointer the optimiser
//     cannot see through), not held in scalar registers across the loop.
//
// Parity contract unchanged (bl_sampling_fast.h): sample_num, flags, status, cell and every cut decision are the exact tier's;
// intensities within the tier's tolerance (tests/test_gpu_tolerant.py asserts 1e-11 of the image maximum against the exact tier).
// Reference arithmetic restated: simulation_sampling.cpp:201-575, :806-839; simulation_coefficients.cpp:253-701; unpolarized.cpp:74-110.
#include "bl_sampling_fast.h"

#pragma clang fp contract(fast)

namespace fused2 {

// One cell of one axis, as the search wants it (LDS, 64 bytes)
struct alignas(16) AxisRow {
  double xf_lo, xf_hi;   // faces of cell c
  double xv;             // centre of cell c: the anchor is c when x >= xv, else c - 1 (simulation_sampling.cpp:485-490) ...
  uint32_t dj_ge, dj_lt; // ... except at the block's ends: anchor = c - dj
  double xv_ge, w_ge;    // centre of the anchor cell and 1 / (distance to the next centre) when x >= xv
  double xv_lt, w_lt;    // ... and when x < xv
};
static_assert(sizeof(AxisRow) == 64, "axis row must be 64 bytes");

typedef const __attribute__((address_space(4))) BlShadeArgs *KernArgs;
// The kernel-argument segment through a pointer the optimiser cannot trace: fields read through it are loaded (s_load) where
// they are used instead of being kept in scalar registers from the kernel's entry on
__device__ __forceinline__ KernArgs kernargs() {
  KernArgs p = (KernArgs)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}
__device__ __forceinline__ double uniform_value(double v) {   // a wave-uniform double into a scalar register pair
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// ---- 64-bit constants as scalar operands (bl_fastmath.h: fma_k, add_k). KS(c) hands a literal through a scalar register pair - two
// s_mov_b32 where the compiler would put two v_mov_b32: the same instruction count, but on the scalar unit and, more to the point,
// none at all for the addends of the Horner steps, which fma_k takes as scalar operands.
#if defined(BLV_VECTOR_LITERALS) || !defined(__HIP_DEVICE_COMPILE__)   // (A/B variant: the compiler's own choice)
#define KS(c) (c)
__device__ __forceinline__ double fma_k(double a, double b, double c_uniform) { return __builtin_fma(a, b, c_uniform); }
__device__ __forceinline__ double add_k(double a, double c_uniform) { return a + c_uniform; }
#else
#define KS(c) BLM_K(c)
using fastmath::add_k;
using fastmath::fma_k;
#endif
// The tier's elementary functions (bl_fastmath.h: same polynomials, same reductions, same accuracy) with their coefficients as
// scalar operands
__device__ __forceinline__ double expm1_core_k(double x, int *k) {
  const double kd = __builtin_rint(x * KS(0x1.71547652b82fep+0));
  double r = __builtin_fma(-kd, KS(0x1.62e42feep-1), x);
  r = __builtin_fma(-kd, KS(0x1.a39ef35793c76p-33), r);
  double q = add_k(r * KS(0x1.94328fcb8199cp-37), 0x1.61bfaa228dde5p-33);
  q = fma_k(q, r, 0x1.1eed7a01fc8b7p-29);
  q = fma_k(q, r, 0x1.ae642c82e33d5p-26);
  q = fma_k(q, r, 0x1.27e4fb7a2782ap-22);
  q = fma_k(q, r, 0x1.71de3a5aa7bb7p-19);
  q = fma_k(q, r, 0x1.a01a01a019b63p-16);
  q = fma_k(q, r, 0x1.a01a01a0196acp-13);
  q = fma_k(q, r, 0x1.6c16c16c16c17p-10);
  q = fma_k(q, r, 0x1.1111111111111p-7);
  q = fma_k(q, r, 0x1.5555555555555p-5);
  q = fma_k(q, r, 0x1.5555555555555p-3);
  const double r2 = r * r;
  *k = (int)kd;
  return r + __builtin_fma(r2 * r, q, 0.5 * r2);
}
__device__ __forceinline__ double exp_k(double x) {
  x = x > KS(710.0) ? KS(710.0) : (x < KS(-746.0) ? KS(-746.0) : x);
  int k;
  const double e = expm1_core_k(x, &k);
  return __builtin_amdgcn_ldexp(1.0 + e, k);
}
__device__ __forceinline__ double expm1_k(double x) {
  x = x > KS(710.0) ? KS(710.0) : (x < KS(-40.0) ? KS(-40.0) : x);
  int k;
  const double e = expm1_core_k(x, &k);
  const double t = __builtin_amdgcn_ldexp(1.0, k);
  return (t - 1.0) + t * e;
}
// 1 / sqrt(x) for finite x > 0 from v_rsq_f64 (2^-26) and ONE Newton step: 4e-16, which is what the tier's reciprocal has as well
// (bl_fastmath.h's rsqrt takes two: a quarter of its instructions for a last place this kernel's results never show)
__device__ __forceinline__ double rsqrt_k(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double e = __builtin_fma(-0.5 * x * y, y, 0.5);
  return __builtin_fma(y, e, y);
}
__device__ __forceinline__ double sqrt_k(double x) { return x > 0.0 ? x * rsqrt_k(x) : 0.0; }
__device__ __forceinline__ double cbrt_k(double x) {
  const int e = __builtin_amdgcn_frexp_exp(x);
  const double mant = __builtin_amdgcn_frexp_mant(x);
  const int q = (int)(((unsigned int)(e + 3072) * 43691u) >> 17) - 1024;
  const double m = __builtin_amdgcn_ldexp(mant, e - 3 * q);
  const float seed = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf((float)m) * -0.33333334f);
  const double third = KS(0x1.5555555555555p-2);
  double z = (double)seed;                                           // m^(-1/3) to 2^-21
  const double h = __builtin_fma(-m, z * z * z, 1.0);
  z = __builtin_fma(z * h, third, z);                                // 2^-41 (one Newton step; the correction below squares it again)
  double c = m * z * z;
  c = __builtin_fma(__builtin_fma(-c * c, c, m), z * z * third, c);
  const double res = __builtin_amdgcn_ldexp(c, q);
  return __builtin_amdgcn_class(x, 0x263) ? x : res;
}
__device__ __forceinline__ double acos_k(double x) {
  const double ax = __builtin_fabs(x);
  const bool small = ax < 0.5;
  const double z = small ? x * x : (1.0 - ax) * 0.5;
  double r = add_k(z * KS(0x1.e58a4f278e007p-6), -0x1.3bd7e353ddbc2p-6);
  r = fma_k(r, z, 0x1.40c91fa8deb7ep-6);
  r = fma_k(r, z, 0x1.8dcdf11997e0fp-9);
  r = fma_k(r, z, 0x1.31777489dfd29p-7);
  r = fma_k(r, z, 0x1.3b462d121c5d2p-7);
  r = fma_k(r, z, 0x1.7b02ef007d23ep-7);
  r = fma_k(r, z, 0x1.c990a42b32b03p-7);
  r = fma_k(r, z, 0x1.1c4efd20ebb99p-6);
  r = fma_k(r, z, 0x1.6e8ba121b9d5fp-6);
  r = fma_k(r, z, 0x1.f1c71c7a5e151p-6);
  r = fma_k(r, z, 0x1.6db6db6dac0eap-5);
  r = fma_k(r, z, 0x1.3333333333389p-4);
  r = fma_k(r, z, 0x1.5555555555555p-3);
  r *= z;
  const double s = small ? x : (z > 0.0 ? z * fastmath::rsqrt(z) : 0.0);
  const double asin_s = __builtin_fma(s, r, s);
  const double pio2 = KS(0x1.921fb54442d18p+0), pio2_lo = KS(0x1.1a62633145c07p-54);
  const double res_small = (pio2 - asin_s) + pio2_lo;
  const double res_neg = __builtin_fma(-2.0, asin_s, 2.0 * pio2) + 2.0 * pio2_lo;
  return small ? res_small : (x < 0.0 ? res_neg : 2.0 * asin_s);
}
__device__ __forceinline__ double atan2_k(double y, double x) {
  const double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
  const double mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
  const bool low = 16.0 * mn <= KS(7.0) * mx, mid = 16.0 * mn < KS(11.0) * mx;
  const double c = low ? 0.0 : (mid ? 0.5 : 1.0);
  const double hi = low ? 0.0 : (mid ? KS(0x1.dac670561bb4fp-2) : KS(0x1.921fb54442d18p-1));
  const double num = __builtin_fma(-c, mx, mn), den = __builtin_fma(c, mn, mx);
  const double u = den > 0.0 ? num * fastmath::rcp(den) : 0.0;
  const double w = u * u;
  double a = add_k(w * KS(-0x1.9a0e3d8214a3cp-7), 0x1.dde84abd3489ap-6);
  a = fma_k(a, w, -0x1.4ac01ab40659fp-5);
  a = fma_k(a, w, 0x1.812cf294b38a9p-5);
  a = fma_k(a, w, -0x1.ae800c7915a0cp-5);
  a = fma_k(a, w, 0x1.e1d239c838f12p-5);
  a = fma_k(a, w, -0x1.1110907ae84ccp-4);
  a = fma_k(a, w, 0x1.3b13abac1919fp-4);
  a = fma_k(a, w, -0x1.745d171e2e854p-4);
  a = fma_k(a, w, 0x1.c71c71c673bd9p-4);
  a = fma_k(a, w, -0x1.24924924918e2p-3);
  a = fma_k(a, w, 0x1.999999999998fp-3);
  a = fma_k(a, w, -0x1.5555555555555p-2);
  double res = hi + __builtin_fma(u * w, a, u);               // atan(mn / mx)
  const double pio2 = KS(0x1.921fb54442d18p+0), pi = KS(0x1.921fb54442d18p+1);
  res = ay > ax ? pio2 - res : res;
  res = x < 0.0 ? pi - res : res;
  return y < 0.0 ? -res : res;
}

// sqrt(x) correctly rounded (blm_sqrt_n's operations, bit for bit) together with 1 / sqrt(x) to ~2e-16 (what the Newton step
// leaves behind): the exact radius for the decisions and its reciprocal for the tolerant angle from one v_rsq
__device__ __forceinline__ double sqrt_with_reciprocal(double x, double *inv) {
#pragma clang fp contract(off)
  double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = y * 0.5;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  *inv = h + h;
  return g;   // (callers pass finite x > 0)
}

struct Located {
  double f_i, f_j, f_k;
  uint32_t status;        // kSample... | kPlainUndecided
  uint32_t cell_bytes;    // byte offset of cell (k_m, j_m, i_m) in the cell array; 0 without a cell to read
};

// What the loop keeps of the grid in scalar registers
struct GridScalars {
  double th_x0, th_inv_w, ph_x0, ph_inv_w;   // theta, phi: cell = floor((x - x0) * inv_w)
  float r_l0, r_linv;                        // r: cell = floor((log2 r - l0) * linv)
  double r_in, r_out;                        // first and last radial face
  int n_i1, n_j1, n_k1;                      // n - 1 per axis
  uint32_t n_i, n_j;                         // cell index = (k n_j + j) n_i + i
  uint32_t lds_r, lds_th, lds_ph;            // LDS byte addresses of the three row tables
  // mesh with refinement (kRefined): the box of the block lattice is guessed like a cell, its descriptor names the block's cells and rows
  float box_l0, box_linv;
  double box_th_x0, box_th_inv_w, box_ph_x0, box_ph_inv_w;
  int n_box_i1, n_box_j1, n_box_k1;          // boxes per axis - 1
  uint32_t n_box_i, n_box_j;
  uint32_t lds_desc;                         // LDS byte address of the descriptors
  int block_interp;                          // inter-block interpolation: a sample with an anchor beyond its own block is the exact pass's
};

// LDS reads by byte address (the row tables live behind the kernel's extern array; LDS addresses are 32-bit numbers, which is
// what lets a row's address be computed with one shift-and-add). Device code only; the host pass sees stubs.
typedef double v2d __attribute__((ext_vector_type(2)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ v2d lds_read2(uint32_t addr) { return *(const __attribute__((address_space(3))) v2d *)addr; }
__device__ __forceinline__ v4u lds_read_bits(uint32_t addr) { return *(const __attribute__((address_space(3))) v4u *)addr; }
__device__ __forceinline__ uint32_t lds_address(const void *p) { return (uint32_t)(const __attribute__((address_space(3))) char *)p; }
#else
__device__ __forceinline__ v2d lds_read2(uint32_t) { return v2d{0.0, 0.0}; }
__device__ __forceinline__ v4u lds_read_bits(uint32_t) { return v4u{0u, 0u, 0u, 0u}; }
__device__ __forceinline__ uint32_t lds_address(const void *) { return 0u; }
#endif

// The row tables of the three axes, one after the other (r, theta, phi), built by the workgroup from the grid's face and centre
// tables. kReciprocal: the anchor's width enters as its reciprocal (tolerant tier: the fraction is one multiplication); otherwise
// as the width itself, xv[c + 1] - xv[c] - the divisor of the exact tier's quotient, from the same subtraction.
template <bool kReciprocal>
__device__ __forceinline__ void stage_axis_rows(const BlGridDevice &g, AxisRow *rows) {
  for (int a = 0; a < 3; a++) {
    const int n = g.n[a];
    const double *xf = g.xf[a], *xv = g.xv[a];
    for (int c = threadIdx.x; c < n; c += blockDim.x) {
      const int c_ge = c == n - 1 ? c - 1 : c, c_lt = c == 0 ? 0 : c - 1;
      AxisRow row;
      row.xf_lo = xf[c];
      row.xf_hi = xf[c + 1];
      row.xv = xv[c];
      row.dj_ge = (uint32_t)(c - c_ge);
      row.dj_lt = (uint32_t)(c - c_lt);
      row.xv_ge = xv[c_ge];
      row.xv_lt = xv[c_lt];
      const double width_ge = xv[c_ge + 1] - xv[c_ge], width_lt = xv[c_lt + 1] - xv[c_lt];
      row.w_ge = kReciprocal ? 1.0 / width_ge : width_ge;
      row.w_lt = kReciprocal ? 1.0 / width_lt : width_lt;
      rows[c] = row;
    }
    rows += n;
  }
}
__device__ __forceinline__ GridScalars grid_scalars(const BlGridDevice &g, uint32_t lds_rows) {
  GridScalars G;
  G.th_x0 = g.cell_x0[1];
  G.th_inv_w = g.cell_inv_w[1];
  G.ph_x0 = g.cell_x0[2];
  G.ph_inv_w = g.cell_inv_w[2];
  G.r_l0 = g.log_l0;
  G.r_linv = g.log_inv_w;
  G.r_in = g.r_face_in;
  G.r_out = g.r_face_out;
  G.n_i1 = g.n[0] - 1;
  G.n_j1 = g.n[1] - 1;
  G.n_k1 = g.n[2] - 1;
  G.n_i = (uint32_t)g.n[0];
  G.n_j = (uint32_t)g.n[1];
  G.block_interp = 0;
  G.lds_desc = 0u;   // (one block: no descriptors)
  G.box_l0 = G.box_linv = 0.0f;
  G.box_th_x0 = G.box_th_inv_w = G.box_ph_x0 = G.box_ph_inv_w = 0.0;
  G.n_box_i1 = G.n_box_j1 = G.n_box_k1 = 0;
  G.n_box_i = G.n_box_j = 1u;
  G.lds_r = lds_rows;
  G.lds_th = G.lds_r + 64u * (uint32_t)g.n[0];
  G.lds_ph = G.lds_th + 64u * (uint32_t)g.n[1];
  return G;
}

// Mesh with refinement: per distinct coordinate row of the blocks a chunk - 16 bytes of header (where the row's cell guess starts and
// how many cells a unit of the coordinate holds: two floats for log2 r, two doubles for theta and phi), then the row's cells as above,
// the anchor rule of a block's ends at every row's ends - and behind the chunks one descriptor per box of the block lattice
// (BlGridDevice::fused_desc, its chunk offsets made LDS addresses here).
template <bool kReciprocal>
__device__ __forceinline__ void stage_refined_rows(const BlGridDevice &g, char *chunks, uint32_t chunks_address) {
  char *at = chunks;
  for (int a = 0; a < 3; a++) {
    const int n = g.nb[a], n_rows = g.n_rows[a];
    const size_t chunk = 16 + 64 * (size_t)n;
    for (int t = threadIdx.x; t < n_rows * n; t += blockDim.x) {
      const int q = t / n, c = t - q * n;
      const double *xf = g.bxf[a] + (size_t)q * (n + 1), *xv = g.bxv[a] + (size_t)q * n;
      const int c_ge = c == n - 1 ? c - 1 : c, c_lt = c == 0 ? 0 : c - 1;
      AxisRow row;
      row.xf_lo = xf[c];
      row.xf_hi = xf[c + 1];
      row.xv = xv[c];
      row.dj_ge = (uint32_t)(c - c_ge);
      row.dj_lt = (uint32_t)(c - c_lt);
      row.xv_ge = xv[c_ge];
      row.xv_lt = xv[c_lt];
      const double width_ge = xv[c_ge + 1] - xv[c_ge], width_lt = xv[c_lt + 1] - xv[c_lt];
      row.w_ge = kReciprocal ? 1.0 / width_ge : width_ge;
      row.w_lt = kReciprocal ? 1.0 / width_lt : width_lt;
      *reinterpret_cast<AxisRow *>(at + (size_t)q * chunk + 16 + 64 * (size_t)c) = row;
    }
    for (int q = threadIdx.x; q < n_rows; q += blockDim.x) {
      const double *guess = g.row_guess[a] + 3 * (size_t)q;
      if (a == 0) *reinterpret_cast<float4 *>(at + (size_t)q * chunk) = make_float4((float)guess[1], (float)guess[2], 0.0f, 0.0f);
      else *reinterpret_cast<double2 *>(at + (size_t)q * chunk) = make_double2(guess[1], guess[2]);
    }
    at += (size_t)n_rows * chunk;
  }
  uint4 *desc = reinterpret_cast<uint4 *>(at);
  const int n_boxes = g.n_edge[0] * g.n_edge[1] * g.n_edge[2];
  for (int t = threadIdx.x; t < n_boxes; t += blockDim.x) {
    uint4 d = reinterpret_cast<const uint4 *>(g.fused_desc)[t];
    d.y += chunks_address;
    d.z += chunks_address;
    d.w += chunks_address;
    desc[t] = d;
  }
}
__device__ __forceinline__ GridScalars grid_scalars_refined(const BlGridDevice &g, uint32_t chunks_address) {
  GridScalars G;
  G.th_x0 = G.th_inv_w = G.ph_x0 = G.ph_inv_w = 0.0;
  G.r_l0 = G.r_linv = 0.0f;
  G.r_in = g.r_face_in;
  G.r_out = g.r_face_out;
  G.n_i1 = g.nb[0] - 1;
  G.n_j1 = g.nb[1] - 1;
  G.n_k1 = g.nb[2] - 1;
  G.n_i = (uint32_t)g.nb[0];
  G.n_j = (uint32_t)g.nb[1];
  G.lds_r = G.lds_th = G.lds_ph = 0u;
  G.box_l0 = g.box_l0;
  G.box_linv = g.box_linv;
  G.box_th_x0 = g.box_x0[0];
  G.box_th_inv_w = g.box_inv_w[0];
  G.box_ph_x0 = g.box_x0[1];
  G.box_ph_inv_w = g.box_inv_w[1];
  G.n_box_i1 = g.n_edge[0] - 1;
  G.n_box_j1 = g.n_edge[1] - 1;
  G.n_box_k1 = g.n_edge[2] - 1;
  G.block_interp = g.block_interp;
  G.n_box_i = (uint32_t)g.n_edge[0];
  G.n_box_j = (uint32_t)g.n_edge[1];
  uint32_t bytes = 0u;
  for (int a = 0; a < 3; a++) bytes += (uint32_t)g.n_rows[a] * (16u + 64u * (uint32_t)g.nb[a]);
  G.lds_desc = chunks_address + bytes;
  return G;
}

// One axis: row of the guessed cell -> anchor shift, fraction, signed distance to the nearest face (negative: the guess is wrong
// or the coordinate lies beyond the grid) and distance to the centre
__device__ __forceinline__ void axis_lookup(uint32_t row_addr, double s, double *frac, uint32_t *dj, double *face_margin, double *centre_margin,
                                            bool *upper = nullptr) {
  const v2d faces = lds_read2(row_addr);
  const v4u mid = lds_read_bits(row_addr + 16u);
  const double xv = __hiloint2double((int)mid.y, (int)mid.x);
  const bool ge = s >= xv;
  if (upper != nullptr) *upper = ge;
  const v2d anchor = lds_read2(row_addr + (ge ? 32u : 48u));
  *dj = ge ? mid.z : mid.w;
  *frac = (s - anchor.x) * anchor.y;
  const double d_lo = s - faces.x, d_hi = faces.y - s;
  *face_margin = d_lo < d_hi ? d_lo : d_hi;
  *centre_margin = __builtin_fabs(s - xv);
}

// The locate step (locate_plain_sample_tolerant's results; bl_sampling_fast.h) for the grids this kernel takes
// kRefined: mesh with refinement - the sample's box of the block lattice is guessed like a cell, the box's descriptor names the block's
// three rows, and the cell is guessed and confirmed in those. A wrong box names a block whose rows do not hold the coordinate: the
// row's faces fail to confirm it (blocks do not overlap, so rows that do confirm all three coordinates are the sample's block's),
// and the sample is left to the exact kernel like any other the margins do not decide.
template <bool kSpinZero, bool kRefined = false>
__device__ __forceinline__ Located locate(const BlSpacetime &st, const GridScalars &G, double camera_r, double band, bool live, double x, double y, double z) {
  x = live ? x : 1.0;
  y = live ? y : 1.0;
  z = live ? z : 1.0;
  double r2;
  {
#pragma clang fp contract(off)
    const double rr2 = x * x + y * y + z * z;   // (the exact tier's operations: bl_radial_coordinate2)
    r2 = rr2;
    if (!kSpinZero) {
      const double a2 = st.bh_a * st.bh_a;
      r2 = 0.5 * (rr2 - a2 + bl_hypot_g(rr2 - a2, 2.0 * st.bh_a * z));
    }
  }
  double r_inv;
  const double r = sqrt_with_reciprocal(r2, &r_inv);
  const bool cut = r > camera_r;                                   // simulation_sampling.cpp:238-243
  const bool off_grid = r < G.r_in || r > G.r_out;                 // :352-394 (theta and phi cover the sphere)
  // ConvertFromCKS (radiation_geometry.cpp:37-57) with the tier's inverse trigonometric functions
  // (cos theta = z / r: the product with the reciprocal, corrected once by its residual, is the correctly rounded quotient - the
  // exact tier's argument bit for bit - in all but a few cases in a million; the arccosine amplifies what is left by 1 / sin theta)
  double cth = z * r_inv;
  cth = __builtin_fma(__builtin_fma(-r, cth, z), r_inv, cth);
  const double th = acos_k(cth);
  double ph = kSpinZero ? atan2_k(y, x) : atan2_k(y, x) - atan2_k(st.bh_a, r);
  const double ph_unwrapped = ph;
  const double two_pi = KS(2.0 * kPi);
  ph += ph < 0.0 ? two_pi : 0.0;
  const double ph_once = ph;
  ph -= ph >= two_pi ? two_pi : 0.0;
  // guessed cells
  const float log2_r = __builtin_amdgcn_logf((float)r);
  int gi, gj, gk;
  uint32_t rows_r = G.lds_r, rows_th = G.lds_th, rows_ph = G.lds_ph, block_bytes = 0u;
  if (kRefined) {
    int bi = (int)((log2_r - G.box_l0) * G.box_linv);
    int bj = (int)((th - G.box_th_x0) * G.box_th_inv_w);
    int bk = (int)((ph - G.box_ph_x0) * G.box_ph_inv_w);
    bi = bi < 0 ? 0 : (bi > G.n_box_i1 ? G.n_box_i1 : bi);
    bj = bj < 0 ? 0 : (bj > G.n_box_j1 ? G.n_box_j1 : bj);
    bk = bk < 0 ? 0 : (bk > G.n_box_k1 ? G.n_box_k1 : bk);
    const v4u desc = lds_read_bits(G.lds_desc + ((__umul24(__umul24((uint32_t)bk, G.n_box_j) + (uint32_t)bj, G.n_box_i) + (uint32_t)bi) << 4));
    block_bytes = desc.x;
    const v4u head_r = lds_read_bits(desc.y);
    const v2d head_th = lds_read2(desc.z), head_ph = lds_read2(desc.w);
    rows_r = desc.y + 16u;
    rows_th = desc.z + 16u;
    rows_ph = desc.w + 16u;
    gi = (int)((log2_r - __uint_as_float(head_r.x)) * __uint_as_float(head_r.y));
    gj = (int)((th - head_th.x) * head_th.y);
    gk = (int)((ph - head_ph.x) * head_ph.y);
  } else {
    gi = (int)((log2_r - G.r_l0) * G.r_linv);
    gj = (int)((th - G.th_x0) * G.th_inv_w);
    gk = (int)((ph - G.ph_x0) * G.ph_inv_w);
  }
  gi = gi < 0 ? 0 : (gi > G.n_i1 ? G.n_i1 : gi);
  gj = gj < 0 ? 0 : (gj > G.n_j1 ? G.n_j1 : gj);
  gk = gk < 0 ? 0 : (gk > G.n_k1 ? G.n_k1 : gk);
  double f_i, f_j, f_k, m_i, m_j, m_k, c_i, c_j, c_k;
  uint32_t di, dj, dk;
  bool up_i = false, up_j = false, up_k = false;
  axis_lookup(rows_r + ((uint32_t)gi << 6), r, &f_i, &di, &m_i, &c_i, kRefined ? &up_i : nullptr);
  axis_lookup(rows_th + ((uint32_t)gj << 6), th, &f_j, &dj, &m_j, &c_j, kRefined ? &up_j : nullptr);
  axis_lookup(rows_ph + ((uint32_t)gk << 6), ph, &f_k, &dk, &m_k, &c_k, kRefined ? &up_k : nullptr);
  // Inter-block interpolation (simulation_sampling.cpp:505-546): the anchor is c or c - 1 by the centre alone, also at a block's ends, and
  // the cell beyond the block is another block's (FindNearbyInds). The row's anchor shift says where that happens - above the last
  // cell's centre it is 1 (the plain rule steps back), below the first cell's it is 0 (it stays): such a sample is the exact pass's.
  // Every other sample has all eight anchors in its own block and the plain rule's anchor and fraction (locate_sample_refined).
  const bool beyond_the_block = kRefined && G.block_interp != 0 && ((up_i ? di == 1u : di == 0u) || (up_j ? dj == 1u : dj == 0u) || (up_k ? dk == 1u : dk == 0u));
  // r is the exact tier's r: its cell is confirmed exactly (first c with xf[c + 1] >= r: xf[c] < r <= xf[c + 1]; m_i is
  // min(r - xf[c], xf[c + 1] - r)) - except on a face itself, where the signed minimum is zero either way: left to the exact pass
  // theta, phi: the tier's own angles, so every value they are compared with must be further away than the band
  double m = m_j < m_k ? m_j : m_k;
  m = m < c_j ? m : c_j;
  m = m < c_k ? m : c_k;
  const double e0 = __builtin_fabs(ph_unwrapped), e1 = __builtin_fabs(ph_once - two_pi);
  m = m < e0 ? m : e0;
  m = m < e1 ? m : e1;
  const bool sampled = live && !cut && !off_grid;
  const bool undecided = sampled && (!(m > band) || !(m_i > 0.0) || beyond_the_block);
  Located out;
  out.f_i = f_i;
  out.f_j = f_j;
  out.f_k = f_k;
  const uint32_t cell = __umul24(__umul24((uint32_t)gk - dk, G.n_j) + ((uint32_t)gj - dj), G.n_i) + ((uint32_t)gi - di);
  out.cell_bytes = sampled ? (cell << 5) + block_bytes : 0u;
  out.status = (!live ? (uint32_t)kSampleNone : (cut ? (uint32_t)kSampleCut : (off_grid ? (uint32_t)kSampleOffGrid : (uint32_t)kSampleInterp)))
      | (undecided ? kPlainUndecided : 0u);
  return out;
}

// The eight corner cells of a located sample: sixteen 16-byte loads at 32-bit offsets from the scalar base pointer
__device__ __forceinline__ void gather_issue(const char *cells, uint32_t cell_bytes, bool interp, uint32_t row_bytes, uint32_t plane_bytes, float4 (&lo)[8],
                                             float4 (&hi)[8]) {
  const uint32_t row = interp ? row_bytes : 0u, plane = interp ? plane_bytes : 0u, next = interp ? 32u : 0u;
#pragma unroll
  for (int corner = 0; corner < 8; corner++) {
    const uint32_t off = cell_bytes + ((corner >> 2) ? plane : 0u) + (((corner >> 1) & 1) ? row : 0u) + ((corner & 1) ? next : 0u);
    const float4 *p = reinterpret_cast<const float4 *>(cells + (size_t)off);
    lo[corner] = p[0];
    hi[corner] = p[1];
  }
}

// gather_finish_tolerant() for an interpolated sample (bl_shade_fast.hip): the trilinear read with fused multiply-adds, the <= 0
// rule, the rounding to float; true when a sum lies too close to the midpoint of two floats for the tier to decide the rounding
__device__ __forceinline__ bool trilinear(const float4 (&lo)[8], const float4 (&hi)[8], double f_i, double f_j, double f_k, float pr[8]) {
  const double w_k[2] = {1.0 - f_k, f_k}, w_j[2] = {1.0 - f_j, f_j}, w_i[2] = {1.0 - f_i, f_i};
  double val[8];
  float first[2];
#pragma unroll
  for (int corner = 0; corner < 8; corner++) {
    float v[8];
    unpack_cell(lo[corner], hi[corner], v);
    const double w = w_k[corner >> 2] * w_j[(corner >> 1) & 1] * w_i[corner & 1];
#pragma unroll
    for (int q = 0; q < 8; q++) {
      if (corner == 0) {
        val[q] = w * (double)v[q];
        if (q < 2) first[q] = v[q];
      } else {
        val[q] = __builtin_fma(w, (double)v[q], val[q]);
      }
    }
  }
  bool near_midpoint = false;
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const uint32_t below = (uint32_t)__double_as_longlong(val[q]) & 0x1fffffffu;
    const uint32_t window = q < 2 ? 2048u : 4096u;
    near_midpoint = near_midpoint | ((below - (0x10000000u - window)) <= 2u * window);
  }
  if (val[0] <= 0.0) val[0] = (double)first[0];   // simulation_sampling.cpp:822-825
  if (val[1] <= 0.0) val[1] = (double)first[1];
#pragma unroll
  for (int q = 0; q < 8; q++) pr[q] = (float)val[q];   // :830-839
  return near_midpoint;
}

// fast_shade_sample() (bl_shade_fast.hip) for one frequency, thermal electrons, spherical Kerr-Schild simulation, as one straight
// line: every lane computes, `have` says whether the lane's sample has coefficients. Returns the record (a, c) of I <- a I + c for
// a sample with coefficients (anything else: the caller's selection) and *undecided when a cut decision is the exact kernel's.
// cut_table (LDS): per cut quantity (rho, n_e, p_gas, Theta_e, B, sigma, 1 / beta) six numbers: lower and upper threshold, the
// guard band around the lower one (lo, hi), the guard band around the upper one (lo, hi); a switched-off threshold is -inf / +inf
// with an empty band.
// kFactors (several frequencies, BlShadeArgs::freq_split): the function ends at the sample's factors - fast_shade_sample()'s, what
// is left once everything that does not depend on the frequency has been evaluated (BlFreqInputs) - for bl_transfer_freq_kernel,
// and stores them itself (in row factors_row of `factors`; -1 for a sample the caller has already left to the exact kernel: eight
// more live doubles across the caller's loop are eight too many).
template <bool kSpinZero, bool kFactors = false>
__device__ __forceinline__ double2 shade(const BlSpacetime &st, const double (&K)[6], double freq, double freq_inv, double x_unit, int cut_mask,
                                         uint32_t cut_table, const float pr[8], double x, double y, double z, double kx, double ky, double kz, double kt,
                                         double momentum_factor, double delta_lambda, bool *have_out, bool *undecided_out, double2 (*factors)[4] = nullptr) {
  const double bh_m = st.bh_m;
  const double bh_a = kSpinZero ? 0.0 : st.bh_a;
  const double a2 = bh_a * bh_a;
  const double rho = pr[0], pgas = pr[1], uu1 = pr[2], uu2 = pr[3], uu3 = pr[4], bb1 = pr[5], bb2 = pr[6], bb3 = pr[7];
  // ---- Kerr-Schild scalars (radiation_geometry.cpp:18-25, :138-262)
  const double pp2 = x * x + y * y;
  const double rr2 = pp2 + z * z;
  double r2 = rr2;
  if (!kSpinZero) {
    const double u = rr2 - a2, v = 2.0 * bh_a * z;
    r2 = 0.5 * (u + bl_sqrt_g(u * u + v * v));
  }
  const double r_inv = rsqrt_k(r2);
  const double r = r2 * r_inv;
  const double ra2 = r2 + a2;
  const double ra_inv = kSpinZero ? r_inv * r_inv : fastmath::rcp(ra2);
  const double lx = kSpinZero ? x * r_inv : (r * x + bh_a * y) * ra_inv;
  const double ly = kSpinZero ? y * r_inv : (r * y - bh_a * x) * ra_inv;
  const double lz = z * r_inv;                     // also cos(theta)
  const double sigma = kSpinZero ? r2 : r2 + a2 * lz * lz;
  const double hh = kSpinZero ? 2.0 * bh_m * r_inv : 2.0 * bh_m * r * fastmath::rcp(sigma);   // 2 m r / Sigma
  const double f = kSpinZero ? hh : 2.0 * bh_m * r2 * r * fastmath::rcp(r2 * r2 + a2 * z * z);
  // ---- null-condition renormalisation of the stored momentum (geodesics.cpp:352-371)
  double lk = lx * kx + ly * ky + lz * kz;
  {
    const double kk = kx * kx + ky * ky + kz * kz;
    const double ta = kk - f * lk * lk;                  // g^ij k_i k_j
    const double tb = 2.0 * kt * f * lk;                 // 2 g^0i k_0 k_i
    const double tc = -(1.0 + f) * kt * kt;              // g^00 k_0 k_0
    const double td = sqrt_k(tb * tb - 4.0 * ta * tc);
    const double factor = (tb < 0.0 ? td - tb : -2.0 * tc) * fastmath::rcp(tb < 0.0 ? 2.0 * ta : tb + td);
    kx *= factor;
    ky *= factor;
    kz *= factor;
    lk *= factor;
  }
  // ---- simulation metric, spherical Kerr-Schild (radiation_geometry.cpp:421-573); x^2 + y^2 = (r^2 + a^2) sin^2
  const double sth2 = pp2 * ra_inv;
  const double g_rr = 1.0 + hh;
  const double g_thth = sigma;
  const double g_tph = kSpinZero ? 0.0 : -hh * bh_a * sth2;
  const double g_rph = kSpinZero ? 0.0 : -g_rr * bh_a * sth2;
  const double g_phph = kSpinZero ? pp2 : (ra2 + hh * a2 * sth2) * sth2;
  // ---- u^mu from the normal-frame velocities (simulation_coefficients.cpp:297-313): u^t and 1 / u^t from one reciprocal square root
  const double u0n2 = 1.0 + g_rr * uu1 * uu1 + 2.0 * g_rph * uu1 * uu3 + g_thth * uu2 * uu2 + g_phph * uu3 * uu3;
  const double ut2 = u0n2 * g_rr;
  const double ut_inv = rsqrt_k(ut2);
  const double ut = ut2 * ut_inv;
  const double ur = uu1 - hh * fastmath::rcp(g_rr) * ut;         // shift^r = (2 m r / Sigma) / (1 + 2 m r / Sigma)
  const double u_r = hh * ut + g_rr * ur + g_rph * uu3;
  const double u_th = g_thth * uu2;
  const double u_ph = g_tph * ut + g_rph * ur + g_phph * uu3;
  // ---- b^mu (:316-330); b.b = (B.B + (u.B)^2) / (u^t)^2
  const double bt = u_r * bb1 + u_th * bb2 + u_ph * bb3;
  const double br = (bb1 + bt * ur) * ut_inv;
  const double bth = (bb2 + bt * uu2) * ut_inv;
  const double bph = (bb3 + bt * uu3) * ut_inv;
  const double bb_sq_lab = g_rr * bb1 * bb1 + 2.0 * g_rph * bb1 * bb3 + g_thth * bb2 * bb2 + g_phph * bb3 * bb3;
  const double b_sq = (bb_sq_lab + bt * bt) * ut_inv * ut_inv;
  // ---- k_i in the simulation's coordinates (Jacobian of radiation_geometry.cpp:69-126)
  const double sth_inv = rsqrt_k(sth2);
  const double k_r = lk;
  const double k_th = (lz * (x * kx + y * ky) - r * sth2 * kz) * sth_inv;
  const double k_ph = x * ky - y * kx;
  const double k_u = kt * ut + k_r * ur + k_th * uu2 + k_ph * uu3;
  const double k_b = kt * bt + k_r * br + k_th * bth + k_ph * bph;
  // ---- plasma state (:274-358), constants folded on the host (BlShadeArgs::fast_k)
  // 1 / rho and 1 / p from one reciprocal where both are positive (single-precision values: the product is an ordinary double)
  double rho_inv, pgas_inv;
  {
    const double rp = rho * pgas;
    const bool both = rho > 0.0 && rp > 0.0 && rp < __builtin_inf();
    const double t = fastmath::rcp(both ? rp : rho);
    rho_inv = both ? t * pgas : t;
    pgas_inv = t * rho;
    if (__builtin_expect(!both, 0)) pgas_inv = fastmath::rcp(pgas);
  }
  const double sigma_cut = b_sq * rho_inv;
  const double beta_inv = 0.5 * b_sq * pgas_inv;
  const double bi2 = beta_inv * beta_inv;
  const double dd = 1.0 + bi2;
  // T_i / T_e = N / D, N = rat_high + rat_low / beta^2, D = 1 + 1 / beta^2: k T_e = (1 + c) k T_tot D / (N + c D). What the
  // coefficients need is 1 / (k T_e): one reciprocal; k T_e itself only where a Theta_e cut looks at it
  const double kte_inv = (K[1] + K[2] * bi2 + K[3] * dd) * fastmath::rcp(K[0] * (pgas * rho_inv) * dd);
  // ---- cell cuts (:361-375): decided here unless a value sits within the guard band of an active threshold
  bool cell_cut = false, undecided = pp2 == 0.0;   // (on the polar axis: the exact kernel's business)
  if (cut_mask != 0) {
    const double bb = (cut_mask & 0x300) ? sqrt_k(b_sq) : 0.0;   // only the field-strength cuts need |b| itself
    const double kb_tt_e = (cut_mask & 0xc0) ? fastmath::rcp(kte_inv) : 0.0;
    const double value[7] = {rho, rho, pgas, kb_tt_e, bb, sigma_cut, beta_inv};   // against thresholds in these units
#pragma unroll
    for (int v = 0; v < 7; v++)
      if ((cut_mask >> (2 * v)) & 3) {   // (wave-uniform)
        const double q = value[v];
        const v2d t = lds_read2(cut_table + 48u * v), b_lo = lds_read2(cut_table + 48u * v + 16u), b_hi = lds_read2(cut_table + 48u * v + 32u);
        cell_cut = cell_cut | (q < t.x) | (q > t.y);
        undecided = undecided | ((q >= b_lo.x) & (q <= b_lo.y)) | ((q >= b_hi.x) & (q <= b_hi.y));
      }
  }
  const bool no_field = bb1 == 0.0 && bb2 == 0.0 && bb3 == 0.0;   // :394
  const bool have = !cell_cut && !no_field;
  *have_out = have;
  *undecided_out = undecided;
  // cos^2 = (k.b)^2 / ((k.u)^2 b.b) (:434-455 in invariant form) and 1 / (k.u) from one reciprocal
  const double t_ku = fastmath::rcp(k_u * b_sq);
  const double k_u_inv = t_ku * b_sq;
  double cos2 = k_b * k_b * (t_ku * k_u_inv);
  cos2 = cos2 < 1.0 ? cos2 : 1.0;
  // |b| sin(theta_B) and its reciprocal from one reciprocal square root
  const double bs2 = b_sq * (1.0 - cos2);
  const double b_sin_inv = rsqrt_k(bs2);                   // (inf along the field: nu / nu_s = inf there, as from 1 / 0)
  const double b_sin = bs2 > 0.0 ? bs2 * b_sin_inv : 0.0;
  const double mf_inv = fastmath::rcp(momentum_factor);
  if (kFactors) {
    // nu = s_nu f_l, x = nu / nu_s = s_x f_l: the roots of s_x, h nu / (k T_e), j_nu nu^2 e^(x^(1/3)) / var_c^2 and the length at unit frequency
    const double s_nu = -k_u * momentum_factor;
    const double s_x = s_nu * b_sin_inv * (kte_inv * kte_inv) * K[4];
    const double s_1_3 = cbrt_k(s_x);
    const double s_1_6 = sqrt_k(s_1_3);
    const double s_nu_inv = -k_u_inv * mf_inv;
    (*factors)[0] = make_double2(have ? 1.0 : 0.0, s_1_6 * s_1_3);   // flag 1: factors follow; 0: nothing to add (cut cell, no field)
    (*factors)[1] = make_double2(s_1_3, s_1_6);
    (*factors)[2] = make_double2(KS(kH) * s_nu * kte_inv, K[5] * (rho * b_sin) * (s_nu_inv * s_nu_inv));
    (*factors)[3] = make_double2(delta_lambda * x_unit * mf_inv, 0.0);
    return make_double2(1.0, 0.0);
  }
  // ---- coefficients at the one frequency (simulation_coefficients.cpp:464-523) and the transfer record (unpolarized.cpp:74-110)
  const double nu = -k_u * momentum_factor * freq;                 // :461-463 times the camera frequency
  const double nu_inv = -k_u_inv * mf_inv * freq_inv;
  const double xx = nu * b_sin_inv * (kte_inv * kte_inv) * K[4];   // nu / nu_s
  const double x_1_3 = cbrt_k(xx);
  const double x_1_6 = sqrt_k(x_1_3);
  const double x_1_2 = x_1_6 * x_1_3;
  const double var_c = x_1_2 + KS(kPow2_11_12) * x_1_6;
  const double j_val = K[5] * (rho * b_sin) * (nu_inv * nu_inv) * exp_k(-x_1_3) * var_c * var_c;
  const double xp = KS(kH) * nu * kte_inv;                             // h nu / (k T_e)
  double planck = xp * (1.0 + 0.5 * xp * (1.0 + KS(1.0 / 3.0) * xp * (1.0 + 0.25 * xp)));
  if (__builtin_expect(have && !(xp < KS(0x1p-10)), 0)) planck = expm1_k(xp);
  double alpha_val = j_val * (planck * KS(kC * kC / (2.0 * kH)));    // j / B_nu
  if (alpha_val * alpha_val <= KS(0x1p-1024)) alpha_val = 0.0;         // :513-523
  const double dl_cgs = delta_lambda * x_unit * mf_inv * freq_inv; // unpolarized.cpp:75-76
  const double delta_tau = alpha_val * dl_cgs;
  // optically thin step: neither expm1 nor a division
  const double p = 1.0 - 0.5 * delta_tau * (1.0 - KS(1.0 / 3.0) * delta_tau * (1.0 - 0.25 * delta_tau));
  const bool absorbing = alpha_val > 0.0;
  double2 rec = make_double2(absorbing ? 1.0 - delta_tau * p : 1.0, j_val * dl_cgs * (absorbing ? p : 1.0));
  if (__builtin_expect(have && absorbing && !(delta_tau < KS(0x1p-10)), 0)) {
    const double source = j_val * fastmath::rcp(alpha_val);
    if (delta_tau <= KS(kDeltaTauMax)) {
      const double e1 = expm1_k(-delta_tau);
      rec = make_double2(1.0 + e1, -source * e1);
    } else {
      rec = make_double2(BL_AFFINE_THICK, source);
    }
  }
  return rec;
}

}  // namespace fused2

// Three samples in flight per lane:
//   next: its position record was requested an iteration ago and is located at the end of this one (row tables in LDS);
//   cur:  located -> corner cells and momentum record requested after the trilinear read has freed the landing registers;
//   prev: cells and records arrived -> trilinear read, arithmetic, record.
// kComposed: the affine maps I <- a I + c of a ray's samples that sit side by side in one DPP row - a SEGMENT, numbered by the
// geodesic kernel (BlTraceArgs::segment_rows; the record carries the segment's number where it otherwise carries the sample's) -
// are composed near -> far by a scan over the row, and the segment's last lane stores ONE map in row ray_offset + segment
// (BlShadeArgs::composed): a seventh of the records a sample-by-sample kernel writes, and as many fewer for the transfer kernel to
// read. A wave that holds a sample left to the exact kernel, or an optically thick step (whose map replaces what lies behind it,
// a NaN included: not a product of numbers), writes its samples' own records instead, by record index, and marks its segments'
// rows as standing for those (BL_COMPOSED_EXPANDED).
// kFactors: several frequencies - a sample leaves as its factors (BlFreqInputs, row ray_offset + n) instead of a transfer record.
// kRefined: a mesh with refinement whose lattice of blocks and whose rows are evenly spaced (BlGridDevice::fused_lds_bytes > 0).
// (Its workgroups may be 512 lanes, one to a compute unit - the same two waves per SIMD - so that one copy of the tables serves eight waves
// and may take most of the compute unit's 160 KiB: a mesh of thousands of small blocks fits. bl_launch_shade_fused2 chooses.)
template <bool kSpinZero, bool kComposed, bool kFactors = false, bool kRefined = false>
__global__ void __launch_bounds__(kRefined ? 512 : 256, kRefined ? 1 : BL_FAST_WAVES) bl_shade_fused2_kernel(const BlShadeArgs P) {
  using namespace fused2;
  extern __shared__ double lds[];
  // ---- LDS: the cut table (7 x 6 doubles), then one row per cell for r, theta, phi
  const uint32_t lds_base = lds_address(lds);
  {
    const BlShadeCold &cc = *P.cold;
    const double inf = __builtin_inf();
    for (int i = threadIdx.x; i < 42; i += blockDim.x) {
      const int v = i / 6, w = i - 6 * v;
      const bool lower_on = (P.plasma.cut_mask >> (2 * v)) & 1, upper_on = (P.plasma.cut_mask >> (2 * v + 1)) & 1;
      double value;
      if (w == 0) value = lower_on ? cc.fast_cut[2 * v] : -inf;
      else if (w == 1) value = upper_on ? cc.fast_cut[2 * v + 1] : inf;
      else if (w == 2) value = lower_on ? cc.fast_cut_lo[2 * v] : inf;
      else if (w == 3) value = lower_on ? cc.fast_cut_hi[2 * v] : -inf;
      else if (w == 4) value = upper_on ? cc.fast_cut_lo[2 * v + 1] : inf;
      else value = upper_on ? cc.fast_cut_hi[2 * v + 1] : -inf;
      lds[i] = value;
    }
    if (kRefined) stage_refined_rows<true>(P.grid, reinterpret_cast<char *>(lds + 48), lds_base + 48u * 8u);
    else stage_axis_rows<true>(P.grid, reinterpret_cast<AxisRow *>(lds + 48));
  }
  __syncthreads();
  const uint32_t n_records = (uint32_t)P.counters_in[BL_CNT_RECORDS];   // (a scratch set holds fewer than 2^32 records)
  const uint32_t first_record = 0u;
  if (n_records <= first_record) return;
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t last = n_records - 1u;
  const BlSpacetime st = P.st;
  const GridScalars G = kRefined ? grid_scalars_refined(P.grid, lds_base + 48u * 8u) : grid_scalars(P.grid, lds_base + 48u * 8u);
  const uint32_t cut_table = lds_base;
  const int cut_mask = P.plasma.cut_mask;
  const double camera_r = P.cuts.camera_r;
  const double band = P.fast_angle_band;
  const double K[6] = {P.fast_k[0], P.fast_k[1], P.fast_k[2], P.fast_k[3], P.fast_k[4], P.fast_k[5]};
  const double freq = uniform_value(P.frequencies[0]);
  const double freq_inv = uniform_value(fastmath::rcp(freq));
  const double x_unit = P.x_unit;
  const bool fallback_nan = P.plasma.fallback_nan != 0;
  const char *cells = reinterpret_cast<const char *>(P.grid.cells);
  const uint32_t row_bytes = (uint32_t)P.grid.stride_row * 32u, plane_bytes = (uint32_t)P.grid.stride_plane * 32u;
  const char *ray_kt = reinterpret_cast<const char *>(P.ray_kt), *ray_factor = reinterpret_cast<const char *>(P.ray_factor);
  const char *ray_offset = reinterpret_cast<const char *>(P.ray_offset);
  const double nan = __longlong_as_double(0x7ff8000000000000ll);
  unsigned long long gathers_wave = 0ull;   // (wave-uniform: counted with ballots)

  // Records are addressed from a wave-uniform base that advances by one grid stride per iteration (scalar arithmetic) plus the
  // lane's own constant 32-bit offset; a position beyond the last record reads the base's record and comes back dead.
  const char *records = reinterpret_cast<const char *>(P.records_hot);
  const uint32_t lane_index = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane_bytes = lane_index << 6;
  uint32_t base_index = first_record;   // record index of lane 0 of block 0 for the sample `prev` ... (wave-uniform)
  auto record_base = [&](uint32_t first) { return records + ((size_t)(first < last ? first : last) << 6); };
  // Three slots take the roles prev -> next -> cur -> prev in turn, and the loop body is written out once per assignment of roles:
  // a sample's registers stay where its requests landed from its first iteration to its last, instead of moving down the pipeline
  // with 46 copies an iteration (the compiler does not unroll a loop whose exit is a wave vote by itself).
  struct Slot {
    double2 h0, h1, c0, c1;   // the sample's record: x y | z (ray, row) | k_x k_y | k_z length
    Located loc;
    double kt, factor;        // per-ray constants, requested with the cells
    long long row;
    bool in;
  };
  Slot s0, s1, s2;
  s0.h0 = s0.c0 = s0.c1 = make_double2(0.0, 0.0);
  s0.h1 = make_double2(0.0, __longlong_as_double((long long)BL_DEAD_RAY));
  s0.kt = 0.0;
  s0.factor = 1.0;
  s0.row = 0;
  s0.loc.f_i = s0.loc.f_j = s0.loc.f_k = 0.0;
  s0.loc.status = kSampleNone;
  s0.loc.cell_bytes = 0u;
  s0.in = false;
  float4 lo[8], hi[8];
#pragma unroll
  for (int c = 0; c < 8; c++) lo[c] = hi[c] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  // (32-bit record indices: the launcher keeps n_records + 3 strides below 2^32)
  s1.in = first_record + lane_index < n_records;
  {
    const double2 *rec = reinterpret_cast<const double2 *>(record_base(first_record) + (size_t)(s1.in ? lane_bytes : 0u));
    s1.h0 = rec[0];
    s1.h1 = rec[1];
    s1.h1.y = s1.in ? s1.h1.y : __longlong_as_double((long long)BL_DEAD_RAY);
  }
  s1.loc = locate<kSpinZero, kRefined>(st, G, camera_r, band, (uint32_t)__double_as_longlong(s1.h1.y) != BL_DEAD_RAY, s1.h0.x, s1.h0.y, s1.h1.x);
  // p (`prev`) is sample base_index - stride + lane_index (none in the first iteration), c (`cur`) base_index + lane_index, x (`next`)
  // one stride on
  // (kRefined) the wave's samples for the exact pass, collected in LDS behind the tables and handed to the list 64 at a time
  uint32_t *deferred_here = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(lds) + (kRefined ? P.grid.fused_lds_bytes : 0)) + (kRefined ? (threadIdx.x >> 6) * 128u : 0u);
  uint32_t n_deferred_here = 0u;   // (wave-uniform)
  auto hand_over_deferred = [&](uint32_t first, uint32_t count) __attribute__((always_inline)) {
    KernArgs args = kernargs();
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long at = 0ull;
    if (lane == 0u) at = atomicAdd(&args->counters[BL_CNT_REDO], (unsigned long long)count);
    at = (unsigned long long)__builtin_amdgcn_readfirstlane((int)(at >> 32)) << 32 | (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)at);
    if (lane < count && at + lane < args->redo_capacity) args->redo_list[at + lane] = (unsigned long long)deferred_here[first + lane];
  };
  auto iteration = [&](Slot &p, Slot &c, Slot &x) __attribute__((always_inline)) {
    const uint32_t ray = (uint32_t)__double_as_longlong(p.h1.y);
    const bool live = ray != BL_DEAD_RAY;
    const uint32_t n = (uint32_t)(((unsigned long long)__double_as_longlong(p.h1.y)) >> 32);
    const uint32_t status = p.loc.status & 0xffu;
    const bool interp = status == (uint32_t)kSampleInterp;
    // the position record of `next`: the oldest request of the iteration, used at its end
    const uint32_t next_first = base_index + stride;
    x.in = next_first + lane_index < n_records;
    {
      const double2 *rec = reinterpret_cast<const double2 *>(record_base(next_first) + (size_t)(x.in ? lane_bytes : 0u));
      x.h0 = rec[0];
      x.h1 = rec[1];
    }
    const double kt = p.kt, momentum_factor = p.factor;
    const long long row_first = p.row;
    float pr[8];
    const bool near_midpoint = trilinear(lo, hi, p.loc.f_i, p.loc.f_j, p.loc.f_k, pr);
    if (kFactors) {
      // (the eight values are only read inside the branch below, and the optimiser would move the trilinear sums there - behind
      // the requests that reuse the landing registers they read, which then have to be copied first: 50 registers, 8 of them spilled)
#pragma unroll
      for (int q = 0; q < 8; q++) asm volatile("" : "+v"(pr[q]));
    }
    gathers_wave += (unsigned long long)__popcll(__ballot(interp));
    fused2::gather_issue(cells, c.loc.cell_bytes, (c.loc.status & 0xffu) == (uint32_t)kSampleInterp, row_bytes, plane_bytes, lo, hi);
    {
      const double2 *rec = reinterpret_cast<const double2 *>(record_base(base_index) + (size_t)(c.in ? lane_bytes : 0u));
      c.c0 = rec[2];
      c.c1 = rec[3];
    }
    // ... and its per-ray constants (random 8-byte reads: a whole iteration ahead of their use, like the cells)
    {
      const uint32_t ray_cur = (uint32_t)__double_as_longlong(c.h1.y);
      const uint32_t ray_bytes = (ray_cur != BL_DEAD_RAY ? ray_cur : 0u) << 3;
      c.kt = *reinterpret_cast<const double *>(ray_kt + (size_t)ray_bytes);
      c.factor = *reinterpret_cast<const double *>(ray_factor + (size_t)ray_bytes);
      c.row = *reinterpret_cast<const long long *>(ray_offset + (size_t)ray_bytes);
    }
    // ---- arithmetic of `prev` (ReverseGeodesics: sample_len = -geodesic_len, geodesics.cpp:840)
    // (Inside a branch - taken by every wave that holds a sample on the grid - not for the lanes it skips: a basic block is the
    // unit the instruction scheduler arranges, and with the arithmetic in one block with the trilinear read, the requests and the
    // search it interleaves all of them and needs 277 vector registers; as a block of its own the kernel fits 220.)
    bool have = false, undecided_cut = false;
    double2 rec = make_double2(1.0, 0.0);
    double2 factors[4];
    if (kFactors) factors[0] = factors[1] = factors[2] = factors[3] = make_double2(0.0, 0.0);
    if (interp)
      rec = shade<kSpinZero, kFactors>(st, K, freq, freq_inv, x_unit, cut_mask, cut_table, pr, p.h0.x, p.h0.y, p.h1.x, p.c0.x, p.c0.y, p.c1.x, kt, momentum_factor,
                                       -p.c1.y, &have, &undecided_cut, kFactors ? &factors : nullptr);
    // a sample off the grid has fallback primitives without a field (no coefficients: I <- I) or NaN ones (I <- I + NaN,
    // simulation_sampling.cpp:377-384); a cut sample has none either
    const bool defer = interp && (undecided_cut || near_midpoint || (p.loc.status & kPlainUndecided) != 0u);
    if (!(interp && have)) rec = make_double2(1.0, (status == (uint32_t)kSampleOffGrid && fallback_nan) ? nan : 0.0);
    if (kFactors) {
      // flag 0: nothing to add (cut sample); 2: NaN primitives off the grid
      if (live && !defer) {
        if (!interp) factors[0].x = (status == (uint32_t)kSampleOffGrid && fallback_nan) ? 2.0 : 0.0;
        double2 *out = reinterpret_cast<double2 *>(P.freq_inputs + (size_t)(row_first + (long long)n));
        out[0] = factors[0];
        out[1] = factors[1];
        out[2] = factors[2];
        out[3] = factors[3];
      }
    } else if (!kComposed) {
      if (live && !defer) {
        double2 *out = P.transfer + (size_t)(row_first + (long long)n);
        *out = rec;
      }
    } else {
      // ---- one map per segment. `n` is the segment's number; lanes of one segment are neighbours within a DPP row and carry
      // the same (ray, segment) word. first: lanes that begin a segment (or a row).
      const uint32_t key_lo = ray, key_hi = n;
      const uint32_t below_lo = (uint32_t)BL_DPP((int)~key_lo, (int)key_lo, 0x111, 0xf), below_hi = (uint32_t)BL_DPP((int)key_hi, (int)key_hi, 0x111, 0xf);
      const unsigned long long first = __ballot(below_lo != key_lo || below_hi != key_hi) | 0x0001000100010001ull;
      // Rows of 16 lanes that hold a sample for the exact kernel or an optically thick step keep their samples' own records
      // (below); the other rows of the wave compose. (By the wave it was 14 % of the samples of the benchmark frame - one
      // deferred sample in 300 - by the row it is 5 %.)
      const unsigned long long needs_records = __ballot(live && (defer || BL_IS_AFFINE_THICK(rec.x)));
      bool expand = false;
      if (__builtin_expect(needs_records != 0ull, 0)) {
        unsigned long long m = needs_records;
        m |= m >> 1; m |= m >> 2; m |= m >> 4; m |= m >> 8;   // bit 0 of every row: any lane of the row
        expand = __builtin_amdgcn_inverse_ballot_w64((m & 0x0001000100010001ull) * 0xffffull);
      }
      if (!expand) {
        // inclusive scan of the maps over the lanes of a segment, nearer sample first: (a', c') o (a, c) = (a' a, a' c + c').
        // Whether the lane d below belongs to the same segment is a matter of where segments begin: scalar masks.
        const unsigned long long g2 = first | (first << 1), g4 = g2 | (g2 << 2), g8 = g4 | (g4 << 4);
        double a = rec.x, c = rec.y;
#define BL_COMPOSE_STEP(CTRL, STARTS)                                                                                          \
        {                                                                                                                      \
          const double a_below = __hiloint2double(BL_DPP(0, __double2hiint(a), CTRL, 0xf), BL_DPP(0, __double2loint(a), CTRL, 0xf)); \
          const double c_below = __hiloint2double(BL_DPP(0, __double2hiint(c), CTRL, 0xf), BL_DPP(0, __double2loint(c), CTRL, 0xf)); \
          if (__builtin_amdgcn_inverse_ballot_w64(~(STARTS))) {                                                                \
            c = __builtin_fma(a_below, c, c_below);                                                                            \
            a = a_below * a;                                                                                                   \
          }                                                                                                                    \
        }
        BL_COMPOSE_STEP(0x111, first)
        BL_COMPOSE_STEP(0x112, g2)
        BL_COMPOSE_STEP(0x114, g4)
        BL_COMPOSE_STEP(0x118, g8)
#undef BL_COMPOSE_STEP
        // the last lane of a segment holds its map
        const unsigned long long last_lanes = (first >> 1) | 0x8000800080008000ull;
        if (__builtin_amdgcn_inverse_ballot_w64(last_lanes) && live) P.composed[(size_t)(row_first + (long long)n)] = make_double2(a, c);
      } else {
        // a row with a sample for the exact kernel or a thick step: its samples' own records, by record index, and rows that say so
        const uint32_t record = base_index - stride + lane_index;
        if (live && !defer) P.transfer[record] = rec;
        const unsigned long long last_lanes = (first >> 1) | 0x8000800080008000ull;
        const uint32_t lane = threadIdx.x & 63u;
        if (((last_lanes >> lane) & 1ull) != 0ull && live) {
          const unsigned long long starts_below = first & ((2ull << lane) - 1ull);   // (bit 0 of every row is set: never empty)
          const uint32_t length = lane - (63u - (uint32_t)__builtin_clzll(starts_below)) + 1u;
          P.composed[(size_t)(row_first + (long long)n)] =
              make_double2(-(double)length, __longlong_as_double((long long)(unsigned long long)(record - (length - 1u))));
        }
      }
    }
    if (kRefined) {
      // Over a mesh with inter-block interpolation one sample in twenty is the exact pass's: one atomic per sample on the list's one
      // counter would serialise the launch. The wave collects its samples' record indices in LDS (128 entries behind the tables) and
      // hands them over 64 at a time: one atomic and one coalesced store per 64 entries.
      const unsigned long long waiting = __ballot(live && defer);
      if (__builtin_expect(waiting != 0ull, 0)) {
        const uint32_t lane = threadIdx.x & 63u;
        if (live && defer) deferred_here[n_deferred_here + (uint32_t)__popcll(waiting & ((1ull << lane) - 1ull))] = base_index - stride + lane_index;
        n_deferred_here += (uint32_t)__popcll(waiting);
        if (n_deferred_here >= 64u) {
          n_deferred_here -= 64u;
          hand_over_deferred(n_deferred_here, 64u);
        }
      }
    } else if (__builtin_expect(live && defer, 0)) {
      KernArgs args = kernargs();
      unsigned long long *counters = args->counters;
      const unsigned long long at = atomicAdd(&counters[BL_CNT_REDO], 1ull);
      if (at < args->redo_capacity) args->redo_list[at] = (unsigned long long)(base_index - stride + lane_index);
    }
    // ---- the search for `next`
    x.loc = locate<kSpinZero, kRefined>(st, G, camera_r, band, x.in && (uint32_t)__double_as_longlong(x.h1.y) != BL_DEAD_RAY, x.h0.x, x.h0.y, x.h1.x);
    x.h1.y = x.in ? x.h1.y : __longlong_as_double((long long)BL_DEAD_RAY);
    base_index = next_first;
  };
  for (;;) {
    if (!__any(s0.in || s1.in)) break;
    iteration(s0, s1, s2);
    if (!__any(s1.in || s2.in)) break;
    iteration(s1, s2, s0);
    if (!__any(s2.in || s0.in)) break;
    iteration(s2, s0, s1);
  }
  if (kRefined && n_deferred_here != 0u) hand_over_deferred(0u, n_deferred_here);
  if ((threadIdx.x & 63) == 0 && gathers_wave != 0ull) atomicAdd(&P.counters[BL_CNT_GATHERS], gathers_wave);
}
#pragma clang fp contract(off)

// =================================================================================================
// Exact tier, locate step inside (bl_shade_exact2_kernel): the same pipeline around the exact tier's own arithmetic - every
// operation of bl_locate_plain_kernel and bl_shade_exact_kernel (bl_shade.hip) on the same operands, so the same bits - without
// the 40 bytes of located sample per record that the two-kernel path writes and reads back (54 GB per benchmark frame). Compiled
// with contraction off like everything exact.
// =================================================================================================
namespace fused2 {

struct LocatedExact {
  double f_i, f_j, f_k, ph_unwrapped;
  uint32_t status, cell_bytes;
};

// One axis of the exact search: the cell find_cell() returns (the first c whose upper face is >= s, simulation_sampling.cpp:
// 458-466) is the guessed one when s <= its upper face and, unless it is the first cell, s > its lower face; any other guess - a
// coordinate within rounding of a face of an unevenly rounded table - is searched for in the tables where they lie in HBM. Then
// the anchor (:485-490) and the fraction, an IEEE quotient by the distance between the centres (row.w = that distance).
__device__ __forceinline__ bool guess_confirmed(uint32_t table, int guess, double s) {
  const v2d faces = lds_read2(table + ((uint32_t)guess << 6));
  return s <= faces.y && (guess == 0 || s > faces.x);
}
__device__ __forceinline__ void axis_lookup_exact(uint32_t table, int cell, double s, double *frac, uint32_t *anchor) {
  const uint32_t row_addr = table + ((uint32_t)cell << 6);
  const v4u mid = lds_read_bits(row_addr + 16u);
  const bool ge = s >= __hiloint2double((int)mid.y, (int)mid.x);
  const v2d centre = lds_read2(row_addr + (ge ? 32u : 48u));
  *anchor = (uint32_t)cell - (ge ? mid.z : mid.w);
  *frac = blm_div(s - centre.x, centre.y);
}

// locate_plain_sample() (bl_sampling.h) on the row tables
// kMeshes (the polarized kernel): G may describe a mesh with refinement (G.lds_desc != 0: stage_refined_rows<false>, a wave-uniform
// branch) - the box of the block lattice is guessed like a cell, its descriptor names the block's rows, and the guessed cell is confirmed
// exactly by the row's faces (both of them: a wrong box names a block whose rows do not hold the coordinate); anything else walks from
// there, box by box and cell by cell, to what locate_sample_refined() - the locate kernel's function - finds: its results bit for bit.
template <bool kSpinZero, bool kMeshes = false>
__device__ __forceinline__ LocatedExact locate_exact(const BlSpacetime &st, const BlGridDevice &g, const GridScalars &G, double camera_r, bool live, double x1, double x2,
                                                     double x3) {
  x1 = live ? x1 : 1.0;
  x2 = live ? x2 : 1.0;
  x3 = live ? x3 : 1.0;
  double r2;
  const double r = bl_radial_coordinate2<kSpinZero>(st, x1, x2, x3, &r2);
  const bool cut = r > camera_r;                                   // simulation_sampling.cpp:238-243
  const double th = bl_acos(blm_div(x3, r));                       // ConvertFromCKS (radiation_geometry.cpp:37-57)
  const double ph_unwrapped = kSpinZero ? bl_atan2(x2, x1) : bl_atan2(x2, x1) - bl_atan(blm_div(st.bh_a, r));
  double ph = ph_unwrapped;
  ph += ph < 0.0 ? 2.0 * kPi : 0.0;
  ph -= ph >= 2.0 * kPi ? 2.0 * kPi : 0.0;
  const bool off_grid = r < G.r_in || r > G.r_out;                 // :352-394 (theta and phi cover the sphere: bl_fused2_applicable)
  const bool sampled = live && !cut && !off_grid;
  if (kMeshes && G.lds_desc != 0u) {
    // (the lattice's scalars from the kernel arguments where they are needed: held in scalar registers across the sample loop they cost
    // the one-block case - the benchmark's exact kernel - eighty spilled scalars and 1.5 ms of its 63.5)
    KernArgs A = kernargs();
    const int n_box_i1 = A->grid.n_edge[0] - 1, n_box_j1 = A->grid.n_edge[1] - 1, n_box_k1 = A->grid.n_edge[2] - 1;
    const uint32_t n_box_i = (uint32_t)A->grid.n_edge[0], n_box_j = (uint32_t)A->grid.n_edge[1];
    const float log2_r = __builtin_amdgcn_logf((float)r);
    int bi = (int)((log2_r - A->grid.box_l0) * A->grid.box_linv);
    int bj = (int)((th - A->grid.box_x0[0]) * A->grid.box_inv_w[0]);
    int bk = (int)((ph - A->grid.box_x0[1]) * A->grid.box_inv_w[1]);
    bi = bi < 0 ? 0 : (bi > n_box_i1 ? n_box_i1 : bi);
    bj = bj < 0 ? 0 : (bj > n_box_j1 ? n_box_j1 : bj);
    bk = bk < 0 ? 0 : (bk > n_box_k1 ? n_box_k1 : bk);
    v4u desc = lds_read_bits(G.lds_desc + ((__umul24(__umul24((uint32_t)bk, n_box_j) + (uint32_t)bj, n_box_i) + (uint32_t)bi) << 4));
    const v4u head_r = lds_read_bits(desc.y);
    const v2d head_th = lds_read2(desc.z), head_ph = lds_read2(desc.w);
    int ci = (int)((log2_r - __uint_as_float(head_r.x)) * __uint_as_float(head_r.y));
    int cj = (int)((th - head_th.x) * head_th.y);
    int ck = (int)((ph - head_ph.x) * head_ph.y);
    ci = ci < 0 ? 0 : (ci > G.n_i1 ? G.n_i1 : ci);
    cj = cj < 0 ? 0 : (cj > G.n_j1 ? G.n_j1 : cj);
    ck = ck < 0 ? 0 : (ck > G.n_k1 ? G.n_k1 : ck);
    auto faces_of = [&](uint32_t rows, int c) { return lds_read2(rows + 16u + ((uint32_t)c << 6)); };
    auto inside = [&](uint32_t rows, int c, double x) {   // (strictly above the lower face: on it the sample is the block's below)
      const v2d faces = faces_of(rows, c);
      return x > faces.x && x <= faces.y;
    };
    if (__builtin_expect(sampled && !(inside(desc.y, ci, r) && inside(desc.z, cj, th) && inside(desc.w, ck, ph)), 0)) {
      // The guesses do not hold (the last place of a float log2, a coordinate on a face): walk. First to the box - the block of the
      // box at hand spans [first face of its row, last face]: a coordinate at or below the first belongs to a box further down, one
      // above the last to a box further up (the rule of locate_sample_refined: the first box whose upper edge is >= the coordinate) -
      // then along the block's rows to the first cell whose upper face is >= the coordinate. All from the rows in LDS.
      for (int step = 0; step < 3 * 4096; step++) {
        const double first_r = faces_of(desc.y, 0).x, last_r = faces_of(desc.y, G.n_i1).y;
        const double first_th = faces_of(desc.z, 0).x, last_th = faces_of(desc.z, G.n_j1).y;
        const double first_ph = faces_of(desc.w, 0).x, last_ph = faces_of(desc.w, G.n_k1).y;
        const int move_i = (r <= first_r && bi > 0) ? -1 : ((r > last_r && bi < n_box_i1) ? 1 : 0);
        const int move_j = (th <= first_th && bj > 0) ? -1 : ((th > last_th && bj < n_box_j1) ? 1 : 0);
        const int move_k = (ph <= first_ph && bk > 0) ? -1 : ((ph > last_ph && bk < n_box_k1) ? 1 : 0);
        if (move_i == 0 && move_j == 0 && move_k == 0) break;
        bi += move_i;
        bj += move_j;
        bk += move_k;
        desc = lds_read_bits(G.lds_desc + ((__umul24(__umul24((uint32_t)bk, n_box_j) + (uint32_t)bj, n_box_i) + (uint32_t)bi) << 4));
      }
      while (ci < G.n_i1 && r > faces_of(desc.y, ci).y) ci++;
      while (ci > 0 && r <= faces_of(desc.y, ci).x) ci--;
      while (cj < G.n_j1 && th > faces_of(desc.z, cj).y) cj++;
      while (cj > 0 && th <= faces_of(desc.z, cj).x) cj--;
      while (ck < G.n_k1 && ph > faces_of(desc.w, ck).y) ck++;
      while (ck > 0 && ph <= faces_of(desc.w, ck).x) ck--;
    }
    LocatedExact out;
    uint32_t i_m, j_m, k_m;
    axis_lookup_exact(desc.y + 16u, ci, r, &out.f_i, &i_m);
    axis_lookup_exact(desc.z + 16u, cj, th, &out.f_j, &j_m);
    axis_lookup_exact(desc.w + 16u, ck, ph, &out.f_k, &k_m);
    out.cell_bytes = desc.x + ((__umul24(__umul24(k_m, G.n_j) + j_m, G.n_i) + i_m) << 5);
    const uint32_t status = (uint32_t)kSampleInterp;
    const bool read = sampled && status == (uint32_t)kSampleInterp;
    out.f_i = read ? out.f_i : 0.0;
    out.f_j = read ? out.f_j : 0.0;
    out.f_k = read ? out.f_k : 0.0;
    out.ph_unwrapped = (!live || cut) ? 0.0 : ph_unwrapped;
    out.status = !live ? (uint32_t)kSampleNone : (cut ? (uint32_t)kSampleCut : (off_grid ? (uint32_t)kSampleOffGrid : status));
    out.cell_bytes = read ? out.cell_bytes : 0u;
    return out;
  }
  int gi = (int)((__builtin_amdgcn_logf((float)r) - G.r_l0) * G.r_linv);
  int gj = (int)((th - G.th_x0) * G.th_inv_w);
  int gk = (int)((ph - G.ph_x0) * G.ph_inv_w);
  gi = gi < 0 ? 0 : (gi > G.n_i1 ? G.n_i1 : gi);
  gj = gj < 0 ? 0 : (gj > G.n_j1 ? G.n_j1 : gj);
  gk = gk < 0 ? 0 : (gk > G.n_k1 ? G.n_k1 : gk);
  // (one copy of the search for all three axes, behind one branch that is almost never taken)
  const bool ok_i = guess_confirmed(G.lds_r, gi, r), ok_j = guess_confirmed(G.lds_th, gj, th), ok_k = guess_confirmed(G.lds_ph, gk, ph);
  if (__builtin_expect(!(ok_i && ok_j && ok_k), 0)) {
    // find_cell() (bl_sampling.h) over the tables where they lie in HBM: the bucket's first candidate, then the forward scan
    // (its tables and scales are read from the kernel arguments here, where they are needed: kernargs())
    KernArgs A = kernargs();
    auto search = [&](int a, double x) {
      const int n_bucket = A->grid.n_bucket[a], n = A->grid.n[a];
      int bucket = (int)((x - A->grid.bucket_x0[a]) * A->grid.bucket_inv_w[a]);
      bucket = bucket < 0 ? 0 : (bucket >= n_bucket ? n_bucket - 1 : bucket);
      int c = A->grid.bucket[a][bucket];
      const double *xf = A->grid.xf[a];
      while (c < n - 1 && !(xf[c + 1] >= x)) c++;
      return c;
    };
    if (!ok_i) gi = search(0, r);
    if (!ok_j) gj = search(1, th);
    if (!ok_k) gk = search(2, ph);
  }
  LocatedExact out;
  uint32_t i_m, j_m, k_m;
  axis_lookup_exact(G.lds_r, gi, r, &out.f_i, &i_m);
  axis_lookup_exact(G.lds_th, gj, th, &out.f_j, &j_m);
  axis_lookup_exact(G.lds_ph, gk, ph, &out.f_k, &k_m);
  out.f_i = sampled ? out.f_i : 0.0;
  out.f_j = sampled ? out.f_j : 0.0;
  out.f_k = sampled ? out.f_k : 0.0;
  out.ph_unwrapped = (!live || cut) ? 0.0 : ph_unwrapped;
  out.status = !live ? (uint32_t)kSampleNone : (cut ? (uint32_t)kSampleCut : (off_grid ? (uint32_t)kSampleOffGrid : (uint32_t)kSampleInterp));
  out.cell_bytes = sampled ? (__umul24(__umul24(k_m, G.n_j) + j_m, G.n_i) + i_m) << 5 : 0u;
  return out;
}

}  // namespace fused2

template <bool kSpinZero>
__global__ void __launch_bounds__(256, 2) bl_shade_exact2_kernel(const BlShadeArgs P) {
  using namespace fused2;
  extern __shared__ double lds[];
  const uint32_t lds_base = lds_address(lds);
  const bool mesh = P.grid.n_blocks > 0;   // (a mesh with refinement: as in the polarized kernel below)
  if (mesh) stage_refined_rows<false>(P.grid, reinterpret_cast<char *>(lds), lds_base);
  else stage_axis_rows<false>(P.grid, reinterpret_cast<AxisRow *>(lds));
  __syncthreads();
  const uint32_t n_records = (uint32_t)P.counters_in[BL_CNT_RECORDS];
  const uint32_t first_record = 0u;
  if (n_records <= first_record) return;
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t last = n_records - 1u;
  const BlSpacetime st = P.st;
  const GridScalars G = mesh ? grid_scalars_refined(P.grid, lds_base) : grid_scalars(P.grid, lds_base);
  const double camera_r = P.cuts.camera_r;
  const float fallback_rho = P.cold->fallback_rho, fallback_pgas = P.cold->fallback_pgas;
  const char *cells = reinterpret_cast<const char *>(P.grid.cells);
  const uint32_t row_bytes = (uint32_t)P.grid.stride_row * 32u, plane_bytes = (uint32_t)P.grid.stride_plane * 32u;
  const char *ray_kt = reinterpret_cast<const char *>(P.ray_kt), *ray_factor = reinterpret_cast<const char *>(P.ray_factor);
  const char *ray_offset = reinterpret_cast<const char *>(P.ray_offset);
  unsigned long long gathers_wave = 0ull;
  const double freq = uniform_value(P.frequencies[0]);
  const char *records = reinterpret_cast<const char *>(P.records_hot);
  const uint32_t lane_index = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane_bytes = lane_index << 6;
  uint32_t base_index = first_record;
  auto record_base = [&](uint32_t first) { return records + ((size_t)(first < last ? first : last) << 6); };
  double2 prev0 = make_double2(0.0, 0.0), prev1 = make_double2(0.0, __longlong_as_double((long long)BL_DEAD_RAY)), prev2 = prev0, prev3 = prev0;
  double2 cur0, cur1;
  double kt_prev = 0.0, factor_prev = 1.0;
  long long row_prev = 0;
  LocatedExact loc_prev, loc_cur;
  loc_prev.f_i = loc_prev.f_j = loc_prev.f_k = loc_prev.ph_unwrapped = 0.0;
  loc_prev.status = kSampleNone;
  loc_prev.cell_bytes = 0u;
  float4 lo[8], hi[8];
#pragma unroll
  for (int c = 0; c < 8; c++) lo[c] = hi[c] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  bool cur_in = first_record + lane_index < n_records;
  {
    const double2 *rec = reinterpret_cast<const double2 *>(record_base(first_record) + (size_t)(cur_in ? lane_bytes : 0u));
    cur0 = rec[0];
    cur1 = rec[1];
    cur1.y = cur_in ? cur1.y : __longlong_as_double((long long)BL_DEAD_RAY);
  }
  loc_cur = locate_exact<kSpinZero, true>(st, P.grid, G, camera_r, (uint32_t)__double_as_longlong(cur1.y) != BL_DEAD_RAY, cur0.x, cur0.y, cur1.x);
  bool prev_in = false;
  while (__any(prev_in || cur_in)) {
    const uint32_t ray = (uint32_t)__double_as_longlong(prev1.y);
    const bool live = ray != BL_DEAD_RAY;
    const uint32_t n = (uint32_t)(((unsigned long long)__double_as_longlong(prev1.y)) >> 32);
    const int status = (int)loc_prev.status;
    const uint32_t next_first = base_index + stride;
    const bool next_in = next_first + lane_index < n_records;
    double2 next0, next1;
    {
      const double2 *rec = reinterpret_cast<const double2 *>(record_base(next_first) + (size_t)(next_in ? lane_bytes : 0u));
      next0 = rec[0];
      next1 = rec[1];
    }
    const double kt = kt_prev, momentum_factor = factor_prev;
    const long long row_first = row_prev;
    float pr[8];
    gather_finish(P, fallback_rho, fallback_pgas, status, lo, hi, loc_prev.f_i, loc_prev.f_j, loc_prev.f_k, pr);
    gathers_wave += (unsigned long long)__popcll(__ballot(status == (int)kSampleInterp));
    fused2::gather_issue(cells, loc_cur.cell_bytes, loc_cur.status == (uint32_t)kSampleInterp, row_bytes, plane_bytes, lo, hi);
    double2 cold0, cold1;
    {
      const double2 *rec = reinterpret_cast<const double2 *>(record_base(base_index) + (size_t)(cur_in ? lane_bytes : 0u));
      cold0 = rec[2];
      cold1 = rec[3];
    }
    {
      const uint32_t ray_cur = (uint32_t)__double_as_longlong(cur1.y);
      const uint32_t ray_bytes = (ray_cur != BL_DEAD_RAY ? ray_cur : 0u) << 3;
      kt_prev = *reinterpret_cast<const double *>(ray_kt + (size_t)ray_bytes);
      factor_prev = *reinterpret_cast<const double *>(ray_factor + (size_t)ray_bytes);
      row_prev = *reinterpret_cast<const long long *>(ray_offset + (size_t)ray_bytes);
    }
    if (live) {
      // ---- bl_shade_exact_kernel's body (bl_shade.hip), call for call
      const double x1 = prev0.x, x2 = prev0.y, x3 = prev1.x;
      const double delta_lambda = -prev3.y;   // ReverseGeodesics: sample_len = -geodesic_len (geodesics.cpp:840)
      double kcov[4] = {kt, prev2.x, prev2.y, prev3.x};
      BlKerrSchild ks;
      bl_kerr_schild<kSpinZero>(st, x1, x2, x3, &ks);
      {   // per-sample renormalisation of the stored momentum (geodesics.cpp:352-371)
        double gcon[4][4];
        bl_gcon_ks(ks, gcon);
        const double factor = bl_renormalization_factor_g(gcon, kcov[0], kcov[1], kcov[2], kcov[3]);
        kcov[1] *= factor;
        kcov[2] *= factor;
        kcov[3] *= factor;
      }
      SampleShade sh;
      sh.have_coefficients = false;
      sh.nu_fluid_over_nu = 0.0;
      sh.n_e_cgs = sh.nu_c_cgs = sh.theta_e = sh.sin_theta_b = sh.kb_tt_e_cgs = 0.0;
      sh.cos_theta_b = sh.sin2_theta_b = sh.cos2_theta_b = 0.0;
      sh.cos_sign = 1.0;
      sh.n_n0_fluid = 0.0;
      sh.fu[0] = sh.fu[1] = sh.fu[2] = sh.fu[3] = 0.0;
      sh.have_cell = false;
      if (status != kSampleCut) sample_finish_simulation<false, true>(P, st, ks, x3 / ks.r, loc_prev.ph_unwrapped, pr, 0.0f, kcov, 1, &sh, nullptr);
      double j_val = 0.0, alpha_val = 0.0;
      if (sh.have_coefficients) simulation_coefficients<false>(P, sh, freq, momentum_factor, &j_val, &alpha_val);
      const double delta_lambda_cgs = bl_div_g(delta_lambda * P.x_unit, freq * momentum_factor);   // unpolarized.cpp:75-76
      P.transfer[(size_t)(row_first + (long long)n)] = transfer_record(j_val, alpha_val, delta_lambda_cgs);
    }
    const bool live_next = next_in && (uint32_t)__double_as_longlong(next1.y) != BL_DEAD_RAY;
    LocatedExact loc_next = loc_cur;
    if (__any(live_next)) loc_next = locate_exact<kSpinZero, true>(st, P.grid, G, camera_r, live_next, next0.x, next0.y, next1.x);
    else loc_next.status = kSampleNone, loc_next.cell_bytes = 0u;
    prev0 = cur0;
    prev1 = cur1;
    prev2 = cold0;
    prev3 = cold1;
    loc_prev = loc_cur;
    prev_in = cur_in;
    cur0 = next0;
    cur1 = next1;
    cur1.y = next_in ? next1.y : __longlong_as_double((long long)BL_DEAD_RAY);
    loc_cur = loc_next;
    cur_in = next_in;
    base_index = next_first;
  }
  if ((threadIdx.x & 63) == 0 && gathers_wave != 0ull) atomicAdd(&P.counters[BL_CNT_GATHERS], gathers_wave);
}

// =================================================================================================
// Polarized runs, locate step inside (bl_shade_polarized2_kernel): bl_locate_plain_kernel + bl_shade_kernel<simulation, aux, extended,
// curved SKS, polarized> behind bl_shade_exact2_kernel's pipeline - the same calls on the same operands, so the same BlPolSample frames,
// BlCoefInputs and auxiliary records, bit for bit (both tiers: the frame of a polarized sample is exact arithmetic in either) -
// without the locate kernel's launch and the 40 bytes of located sample per record it writes and this kernel would read back.
// Reference: simulation_sampling.cpp:201-575, :806-839; simulation_coefficients.cpp:253-455; polarized.cpp:163-265.
// =================================================================================================
// kAuxRecords: the run keeps BlAuxSample records (an auxiliary row besides tau, or a rendering); without them the kernel is 40
// registers lighter.
// kCoefficients: one frequency, thermal electrons only (configuration 4): the sample's eight polarized coefficients are evaluated here,
// from the scalars the frame's arithmetic has just left in registers, instead of by bl_polarized_coefficients_kernel from a 64-byte
// BlCoefInputs written and read back through HBM (per 1024^2 frame: 48 GB written, 100 GB fetched - with the record's tag - and a
// kernel's worth of launch and tail). The same functions on the same operands: the same bits.
template <bool kSpinZero, bool kAuxRecords, bool kCoefficients = false>
__global__ void __launch_bounds__(256, 2) bl_shade_polarized2_kernel(const BlShadeArgs P) {
  using namespace fused2;
  extern __shared__ double lds[];
  const uint32_t lds_base = lds_address(lds);
  // (a mesh with refinement that the tolerant tier's fused kernel would take - BlGridDevice::fused_lds_bytes - has its row chunks and box
  // descriptors here too, with widths where that kernel keeps reciprocals: locate_exact<., true>)
  const bool mesh = P.grid.n_blocks > 0;
  if (mesh) stage_refined_rows<false>(P.grid, reinterpret_cast<char *>(lds), lds_base);
  else stage_axis_rows<false>(P.grid, reinterpret_cast<AxisRow *>(lds));
  __syncthreads();
  const uint32_t n_records = (uint32_t)P.counters_in[BL_CNT_RECORDS];
  const uint32_t first_record = 0u;
  if (n_records <= first_record) return;
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t last = n_records - 1u;
  const BlSpacetime st = P.st;
  const GridScalars G = mesh ? grid_scalars_refined(P.grid, lds_base) : grid_scalars(P.grid, lds_base);
  const double camera_r = P.cuts.camera_r;
  const float fallback_rho = P.cold->fallback_rho, fallback_pgas = P.cold->fallback_pgas;
  const bool nan_rays = P.plasma.fallback_nan != 0;
  const char *cells = reinterpret_cast<const char *>(P.grid.cells);
  const uint32_t row_bytes = (uint32_t)P.grid.stride_row * 32u, plane_bytes = (uint32_t)P.grid.stride_plane * 32u;
  const char *ray_kt = reinterpret_cast<const char *>(P.ray_kt), *ray_offset = reinterpret_cast<const char *>(P.ray_offset);
  unsigned long long gathers_wave = 0ull;
  const char *records = reinterpret_cast<const char *>(P.records_hot);
  const uint32_t lane_index = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane_bytes = lane_index << 6;
  uint32_t base_index = first_record;
  auto record_base = [&](uint32_t first) { return records + ((size_t)(first < last ? first : last) << 6); };
  double2 prev0 = make_double2(0.0, 0.0), prev1 = make_double2(0.0, __longlong_as_double((long long)BL_DEAD_RAY)), prev2 = prev0, prev3 = prev0;
  double2 cur0, cur1;
  double kt_prev = 0.0, factor_prev = 1.0;
  long long row_prev = 0;
  unsigned char flag_prev = 0;
  LocatedExact loc_prev, loc_cur;
  loc_prev.f_i = loc_prev.f_j = loc_prev.f_k = loc_prev.ph_unwrapped = 0.0;
  loc_prev.status = kSampleNone;
  loc_prev.cell_bytes = 0u;
  float4 lo[8], hi[8];
#pragma unroll
  for (int c = 0; c < 8; c++) lo[c] = hi[c] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  bool cur_in = first_record + lane_index < n_records;
  {
    const double2 *rec = reinterpret_cast<const double2 *>(record_base(first_record) + (size_t)(cur_in ? lane_bytes : 0u));
    cur0 = rec[0];
    cur1 = rec[1];
    cur1.y = cur_in ? cur1.y : __longlong_as_double((long long)BL_DEAD_RAY);
  }
  loc_cur = locate_exact<kSpinZero, true>(st, P.grid, G, camera_r, (uint32_t)__double_as_longlong(cur1.y) != BL_DEAD_RAY, cur0.x, cur0.y, cur1.x);
  bool prev_in = false;
  while (__any(prev_in || cur_in)) {
    const uint32_t ray = (uint32_t)__double_as_longlong(prev1.y);
    const bool live = ray != BL_DEAD_RAY;
    const uint32_t n = (uint32_t)(((unsigned long long)__double_as_longlong(prev1.y)) >> 32);
    int status = (int)loc_prev.status;
    const uint32_t next_first = base_index + stride;
    const bool next_in = next_first + lane_index < n_records;
    double2 next0, next1;
    {
      const double2 *rec = reinterpret_cast<const double2 *>(record_base(next_first) + (size_t)(next_in ? lane_bytes : 0u));
      next0 = rec[0];
      next1 = rec[1];
    }
    const double kt = kt_prev, momentum_factor = factor_prev;
    const long long row_first = row_prev;
    const unsigned char ray_flag = flag_prev;
    float pr[8];
    gather_finish(P, fallback_rho, fallback_pgas, status, lo, hi, loc_prev.f_i, loc_prev.f_j, loc_prev.f_k, pr);
    gathers_wave += (unsigned long long)__popcll(__ballot(status == (int)kSampleInterp));
    double2 cold0, cold1;
    {
      const double2 *rec = reinterpret_cast<const double2 *>(record_base(base_index) + (size_t)(cur_in ? lane_bytes : 0u));
      cold0 = rec[2];
      cold1 = rec[3];
    }
    {
      const uint32_t ray_cur = (uint32_t)__double_as_longlong(cur1.y);
      const uint32_t ray_slot = ray_cur != BL_DEAD_RAY ? ray_cur : 0u;
      kt_prev = *reinterpret_cast<const double *>(ray_kt + (size_t)(ray_slot << 3));
      if (kCoefficients) factor_prev = P.ray_factor[ray_slot];
      row_prev = *reinterpret_cast<const long long *>(ray_offset + (size_t)(ray_slot << 3));
      flag_prev = P.ray_flags[ray_slot];
    }
    if (live) {
      // ---- bl_shade_kernel's body for a polarized run (bl_shade.hip), call for call
      const unsigned long long idx_cur = (unsigned long long)(base_index - stride + lane_index);
      const size_t row = (size_t)(row_first + (long long)n);
      const double x1 = prev0.x, x2 = prev0.y, x3 = prev1.x;
      const double delta_lambda = -prev3.y;   // ReverseGeodesics: sample_len = -geodesic_len (geodesics.cpp:840)
      double kcov[4] = {kt, prev2.x, prev2.y, prev3.x};
      BlKerrSchild ks;
      bl_kerr_schild<kSpinZero>(st, x1, x2, x3, &ks);
      {   // per-sample renormalisation of the stored momentum (geodesics.cpp:352-371)
        double gcon[4][4];
        bl_gcon_ks(ks, gcon);
        const double factor = bl_renormalization_factor_g(gcon, kcov[0], kcov[1], kcov[2], kcov[3]);
        kcov[1] *= factor;
        kcov[2] *= factor;
        kcov[3] *= factor;
      }
      SampleShade sh;
      sh.have_coefficients = false;
      sh.nu_fluid_over_nu = 0.0;
      sh.n_e_cgs = sh.nu_c_cgs = sh.theta_e = sh.sin_theta_b = sh.kb_tt_e_cgs = 0.0;
      sh.cos_theta_b = sh.sin2_theta_b = sh.cos2_theta_b = 0.0;
      sh.cos_sign = 1.0;
      sh.n_n0_fluid = 0.0;
      sh.fu[0] = sh.fu[1] = sh.fu[2] = sh.fu[3] = 0.0;
      sh.have_cell = false;
      // rays that ended on ray_max_steps / retries with fallback_nan sample NaN primitives at every sample, cuts not applied
      // (simulation_sampling.cpp:211-216)
      if (nan_rays && ray_flag != 0) {
        const float fnan = __int_as_float(0x7fc00000);
        for (int v = 0; v < 8; v++) pr[v] = fnan;
        status = kSampleOffGrid;
      }
      if (status != kSampleCut)
        sample_finish_simulation<true, true>(P, st, ks, x3 / ks.r, loc_prev.ph_unwrapped, pr, 0.0f, kcov, P.aux_need_coefficients, &sh, P.pol_samples + row);
      // (what these two read of the arguments - output arrays, the list of samples without coefficients - they read where they use it)
      const BlShadeArgs &A = kernel_arguments_in_place<BlShadeArgs>();
      if (kAuxRecords) write_aux_record(A, st, ks, idx_cur, row, sh, kcov, x1, x2, x3, delta_lambda);
      if (!kCoefficients) {
        write_polarized_inputs(A, idx_cur, row, sh, kcov, pr, x1, x2, x3, delta_lambda);
      } else {
        // a sample with coefficients leaves them; one without leaves what bl_polarized_frame_kernel builds its frame from, as ever
        if (!sh.have_coefficients) write_polarized_inputs(A, idx_cur, row, sh, kcov, pr, x1, x2, x3, delta_lambda);
        else {
          BlPolSample *ps = A.pol_samples + row;
          ps->x[0] = x1; ps->x[1] = x2; ps->x[2] = x3;
          ps->delta_lambda = delta_lambda;
        }
        A.have_flags[idx_cur] = sh.have_coefficients ? 1 : 0;
        // bl_polarized_coefficients_kernel<false, true>'s body (bl_coefficients_freq.hip), call for call
        sh.sin2_theta_b = 1.0 - sh.cos2_theta_b;
        sh.sin_theta_b = bl_sqrt_g(sh.sin2_theta_b);
        sh.cos_theta_b = bl_sqrt_g(sh.cos2_theta_b) * sh.cos_sign;
        sh.theta_e_096 = sh.kk_0 = sh.kk_1 = sh.kk_2 = 0.0;
        if (sh.have_coefficients && A.plasma.plasma_thermal_frac != 0.0) {
          sh.theta_e_096 = bl_pow(sh.theta_e, 0.96);
          if (sh.theta_e >= 0.01) bl_cyl_bessel_k012(1.0 / sh.theta_e, &sh.kk_0, &sh.kk_1, &sh.kk_2);
        }
        const double freq = A.frequencies[0];
        double j_val = 0.0, alpha_val = 0.0;
        double2 pc[3] = {make_double2(0.0, 0.0), make_double2(0.0, 0.0), make_double2(0.0, 0.0)};
        if (sh.have_coefficients) simulation_coefficients<false>(A, sh, freq, momentum_factor, &j_val, &alpha_val);
        polarized_coefficients<false>(A, sh, freq, momentum_factor, j_val, alpha_val, pc);
        double2 *out = A.pol_coeffs + row * 4;
        out[0] = make_double2(j_val, alpha_val);
        out[1] = pc[0];
        out[2] = pc[1];
        out[3] = pc[2];
      }
    }
    // the corner cells of `cur`: requested behind the frame's arithmetic, whose tetrad needs the sixty-four registers they land in
    // (with the requests in front of it the kernel keeps 3 ... 24 registers in scratch memory); the search for `next` and the
    // other wave of the SIMD cover their way
    fused2::gather_issue(cells, loc_cur.cell_bytes, loc_cur.status == (uint32_t)kSampleInterp, row_bytes, plane_bytes, lo, hi);
    const bool live_next = next_in && (uint32_t)__double_as_longlong(next1.y) != BL_DEAD_RAY;
    LocatedExact loc_next = loc_cur;
    if (__any(live_next)) loc_next = locate_exact<kSpinZero, true>(st, P.grid, G, camera_r, live_next, next0.x, next0.y, next1.x);
    else loc_next.status = kSampleNone, loc_next.cell_bytes = 0u;
    prev0 = cur0;
    prev1 = cur1;
    prev2 = cold0;
    prev3 = cold1;
    loc_prev = loc_cur;
    prev_in = cur_in;
    cur0 = next0;
    cur1 = next1;
    cur1.y = next_in ? next1.y : __longlong_as_double((long long)BL_DEAD_RAY);
    loc_cur = loc_next;
    cur_in = next_in;
    base_index = next_first;
  }
  if ((threadIdx.x & 63) == 0 && gathers_wave != 0ull) atomicAdd(&P.counters[BL_CNT_GATHERS], gathers_wave);
}

extern "C" hipError_t bl_launch_shade_polarized2(const BlShadeArgs *args, int grid, hipStream_t stream) {
  const BlGridDevice &g = args->grid;
  // (a mesh with refinement: the fused kernel's tables without its 48 doubles of cut thresholds - bl_polarized2_refined_applicable)
  const size_t lds = g.n_blocks > 0 ? (size_t)g.fused_lds_bytes - 48 * sizeof(double) : 64 * (size_t)(g.n[0] + g.n[1] + g.n[2]);
  if (args->pol_samples == nullptr || args->coef_inputs == nullptr) return hipErrorInvalidValue;
  const bool spin_zero = args->st.bh_a == 0.0, records = args->aux_record_unused == 0;
  if (records && args->aux == nullptr) return hipErrorInvalidValue;
  const bool inside = args->have_flags != nullptr && !records && spin_zero;   // (bl_render.hip: one frequency, thermal electrons only, no spin)
  if (lds > 64 * 1024) {   // (two 256-lane workgroups to a compute unit: up to 76 KiB each)
    const void *kernel = inside ? reinterpret_cast<const void *>(&bl_shade_polarized2_kernel<true, false, true>)
        : (spin_zero ? (records ? reinterpret_cast<const void *>(&bl_shade_polarized2_kernel<true, true>) : reinterpret_cast<const void *>(&bl_shade_polarized2_kernel<true, false>))
                     : (records ? reinterpret_cast<const void *>(&bl_shade_polarized2_kernel<false, true>) : reinterpret_cast<const void *>(&bl_shade_polarized2_kernel<false, false>)));
    const hipError_t err = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 76 * 1024);
    if (err != hipSuccess) return err;
  }
#define BL_LAUNCH_P2(S, A) hipLaunchKernelGGL((bl_shade_polarized2_kernel<S, A>), dim3(grid), dim3(256), lds, stream, *args)
  if (inside) {
    hipLaunchKernelGGL((bl_shade_polarized2_kernel<true, false, true>), dim3(grid), dim3(256), lds, stream, *args);
  } else if (spin_zero && records) BL_LAUNCH_P2(true, true);
  else if (spin_zero) BL_LAUNCH_P2(true, false);
  else if (records) BL_LAUNCH_P2(false, true);
  else BL_LAUNCH_P2(false, false);
#undef BL_LAUNCH_P2
  return hipGetLastError();
}

// (the exact tier's use of the fused kernel: one frequency, plain image; bl_render.hip checks the rest with bl_fused2_applicable)
extern "C" hipError_t bl_launch_shade_exact2(const BlShadeArgs *args, int grid, hipStream_t stream) {
  const BlGridDevice &g = args->grid;
  const size_t lds = g.n_blocks > 0 ? (size_t)g.fused_lds_bytes - 48 * sizeof(double) : 64 * (size_t)(g.n[0] + g.n[1] + g.n[2]);
  if (lds > 64 * 1024) {   // (a mesh's tables, two workgroups to a compute unit: bl_polarized2_refined_applicable)
    const void *kernel = args->st.bh_a == 0.0 ? reinterpret_cast<const void *>(&bl_shade_exact2_kernel<true>) : reinterpret_cast<const void *>(&bl_shade_exact2_kernel<false>);
    const hipError_t err = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 76 * 1024);
    if (err != hipSuccess) return err;
  }
  if (args->st.bh_a == 0.0) hipLaunchKernelGGL((bl_shade_exact2_kernel<true>), dim3(grid), dim3(256), lds, stream, *args);
  else hipLaunchKernelGGL((bl_shade_exact2_kernel<false>), dim3(grid), dim3(256), lds, stream, *args);
  return hipGetLastError();
}

// Whether a render can take this kernel (the caller has checked what the locate step inside needs, one frequency without the
// per-frequency split, interleaved records whose momenta are not renormalised yet): one block, faces evenly spaced in log r / theta
// / phi with the angles covering the sphere, at least two cells per axis, a cell array and ray slots that 32-bit byte offsets cover
// (record indices always do: PlanScratch), room in LDS for the row tables.
extern "C" int bl_fused2_applicable(const BlGridDevice *grid, int n_nu, long long n_rays) {
  const BlGridDevice &g = *grid;
  if (n_nu != 1 || n_rays >= (1ll << 29)) return 0;   // (ray slot x 8 bytes as a 32-bit offset)
  if (g.n_blocks != 0 || g.fmks || g.nb[0] != g.n[0] || g.nb[1] != g.n[1] || g.nb[2] != g.n[2]) return 0;
  if ((g.uniform_mask & 6) != 6 || !g.log_uniform || !g.full_sphere) return 0;
  if (g.n[0] < 2 || g.n[1] < 2 || g.n[2] < 2) return 0;
  const unsigned long long n_cells = (unsigned long long)g.n[0] * g.n[1] * g.n[2];
  if (n_cells * 32ull >= (1ull << 32) || (unsigned long long)g.n[1] * g.n[2] >= (1ull << 24) || g.n[0] >= (1 << 24)) return 0;
  if (g.stride_row != g.n[0] || g.stride_plane != g.n[0] * g.n[1]) return 0;
  const size_t lds = 48 * sizeof(double) + 64 * (size_t)(g.n[0] + g.n[1] + g.n[2]);
  return lds <= 64u * 1024u ? 1 : 0;
}

// The polarized kernel over a mesh with refinement: the same tables (widths for reciprocals), 256-lane workgroups two to a compute unit
extern "C" int bl_polarized2_refined_applicable(const BlGridDevice *grid, long long n_rays) {
  return (grid->n_blocks > 0 && grid->fused_lds_bytes > 0 && grid->fused_lds_bytes <= 76 * 1024 && !grid->block_interp && n_rays < (1ll << 29)) ? 1 : 0;
}

// ... or its instantiation for a mesh with refinement (one frequency, composed maps: what bl_render.hip asks for beside this; the
// geometry was checked when the mesh was staged, UploadRefinedGrid)
// With inter-block interpolation the samples with an anchor beyond their own block - the outer half cell of every block - go to the exact
// pass (bl_shade_kernel<..., kRedo>, which finds their anchor cells): 5 % of the samples with 64^3 blocks, 18 % with 16^3; blocks of
// fewer than twelve cells along an axis (a quarter of the samples and more) stay on the locate kernel + bl_shade_fast_kernel.
extern "C" int bl_fused2_refined_applicable(const BlGridDevice *grid, int n_nu, long long n_rays) {
  if (grid->block_interp && (grid->nb[0] < 12 || grid->nb[1] < 12 || grid->nb[2] < 12)) return 0;
  return (grid->n_blocks > 0 && grid->fused_lds_bytes > 0 && n_nu == 1 && n_rays < (1ll << 29)) ? 1 : 0;
}

extern "C" hipError_t bl_launch_shade_fused2(const BlShadeArgs *args, int grid, hipStream_t stream) {
  const BlGridDevice &g = args->grid;
  if (g.n_blocks > 0) {
    if (args->composed == nullptr || args->freq_split || g.fused_lds_bytes <= 0) return hipErrorInvalidValue;
    const void *kernel = args->st.bh_a == 0.0 ? reinterpret_cast<const void *>(&bl_shade_fused2_kernel<true, true, false, true>)
                                              : reinterpret_cast<const void *>(&bl_shade_fused2_kernel<false, true, false, true>);
    if (g.fused_lds_bytes + 4096 > 64 * 1024) {   // (more dynamic LDS than a launch gets unasked: up to BL_FUSED_REFINED_LDS of the compute unit's 160 KiB)
      const hipError_t err = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BL_FUSED_REFINED_LDS + 4096);
      if (err != hipSuccess) return err;
    }
    // Tables that fit twice into a compute unit's LDS: 256-lane workgroups as for one block (`grid` of them). Larger ones: one 512-lane
    // workgroup to a compute unit, and one round of them (a workgroup's LDS is free for the next only when its last wave has ended:
    // with several rounds every round's tail idles seven waves; measured 29.6 against 27.5 ms on the mesh that fits either way)
    const bool two_to_a_unit = g.fused_lds_bytes <= 76 * 1024;
    const dim3 blocks(two_to_a_unit ? grid : (grid >= 8 ? grid / 8 : 1)), lanes(two_to_a_unit ? 256 : 512);
    const size_t lds_bytes = (size_t)g.fused_lds_bytes + (size_t)(lanes.x / 64) * 512;   // (+ the waves' lists of samples for the exact pass)
    if (args->st.bh_a == 0.0) hipLaunchKernelGGL((bl_shade_fused2_kernel<true, true, false, true>), blocks, lanes, lds_bytes, stream, *args);
    else hipLaunchKernelGGL((bl_shade_fused2_kernel<false, true, false, true>), blocks, lanes, lds_bytes, stream, *args);
    return hipGetLastError();
  }
  const size_t lds = 48 * sizeof(double) + 64 * (size_t)(g.n[0] + g.n[1] + g.n[2]);
  const bool spin_zero = args->st.bh_a == 0.0, composed = args->composed != nullptr;
#define BL_LAUNCH_F2(S, C) hipLaunchKernelGGL((bl_shade_fused2_kernel<S, C>), dim3(grid), dim3(256), lds, stream, *args)
  if (args->freq_split) {
    if (spin_zero) hipLaunchKernelGGL((bl_shade_fused2_kernel<true, false, true>), dim3(grid), dim3(256), lds, stream, *args);
    else hipLaunchKernelGGL((bl_shade_fused2_kernel<false, false, true>), dim3(grid), dim3(256), lds, stream, *args);
  } else if (spin_zero && composed) BL_LAUNCH_F2(true, true);
  else if (spin_zero) BL_LAUNCH_F2(true, false);
  else if (composed) BL_LAUNCH_F2(false, true);
  else BL_LAUNCH_F2(false, false);
#undef BL_LAUNCH_F2
  return hipGetLastError();
}

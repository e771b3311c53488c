// bl_polarized.hip - polarized radiative transfer (image_polarization = true), gfx950.
//
// Replaces RadiationIntegrator::IntegratePolarizedRadiation (reference src/radiation_integrator/polarized.cpp:
// 51-949) for the rays of one chunk: one ray per lane, frequencies in sequence. Per sample (far -> near) the
// coherency tensor N^{mu nu} (4 x 4 complex) is parallel-transported half a step with the connection and k^mu
// averaged over the previous and the current sample (:181-198), taken into the fluid's orthonormal tetrad
// (:259-285), turned into Stokes parameters (:288-292), coupled to the plasma over the sample's length - rotation
// split from emission / absorption (:388-568) or the joint analytic solution (:571-779) - limited to physical
// states (:782-790), put back (:793-813) and transported the second half step (:816-833). At the end N is
// projected on the camera's tetrad (:875-939) and scaled by nu^3 (:942-949): image rows 4 l + (I, Q, U, V).
//
// Inputs are what the coefficient kernel left in HBM in auxiliary-image mode: (j_I, alpha_I) pairs, the three
// polarized coefficient pairs, and one BlPolSample per sample: position, length, k^mu and rows 1 and 2 of the fluid
// tetrad - the frame is sample-parallel work and the coefficient kernel has already built it, so the sequential
// part of a ray keeps only what depends on N. The other image rows of a polarized run come from
// bl_transfer_aux_kernel.
//
// What is sequential, per sample: the connection at the sample (it enters through its average with the previous
// sample's, element by element, so the previous one is carried: 64 doubles per ray, kept in LDS), two half-step transports, the
// projection on the tetrad, the coupling, the way back. Projection and its inverse only read rows 1 and 2 of the
// tetrad and only produce / consume the (1,1), (1,2), (2,1), (2,2) tetrad components (:288-292, :795-798); the
// sums the reference runs over the other rows add products with exact zeros, which change nothing (see
// from_stokes). State per ray: N and N_temp (32 doubles each, see transport), the previous connection (64), the previous k^mu (4).
//
// Arithmetic: plain IEEE double operations in the reference's order (no contraction: -ffp-contract=off), the
// pinned elementary functions of blmath.h; complex numbers as (re, im) pairs with the component-wise
// semantics libstdc++ gives real x complex and complex + complex.
#include <hip/hip_runtime.h>

#include "bl_device.h"
#include "bl_pol_frame.h"

namespace {

constexpr double kPi = 3.141592653589793;
constexpr double kDeltaTauMax = 100.0;   // radiation_integrator.hpp:191

struct Cplx {
  double re, im;
};

__device__ __forceinline__ Cplx cscale(double a, Cplx z) { return Cplx{a * z.re, a * z.im}; }
__device__ __forceinline__ Cplx cadd(Cplx a, Cplx b) { return Cplx{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ Cplx csub(Cplx a, Cplx b) { return Cplx{a.re - b.re, a.im - b.im}; }

// GeodesicConnection (radiation_geometry.cpp:274-412) from the Kerr-Schild scalars of the point (r, r^2, f, l_i:
// the expressions of :283-292 are those of bl_kerr_schild). The connection is not materialised on its own: every
// component, as soon as it is known, (1) enters gk_new[mu][beta] += k[alpha] Gamma[mu][alpha][beta], the contraction
// the second half step uses (:816-822), (2) is averaged with the previous sample's component and enters
// gk_avg[mu][beta] += k_avg[alpha] Gamma_avg[mu][alpha][beta] of the first half step (:150-180), and (3) replaces the
// previous sample's component. For fixed (mu, beta) the alpha terms arrive in ascending order, as in the reference.
__device__ __forceinline__ void connection_contractions(const BlSpacetime &st, double x, double y, double z, const BlKerrSchild &ks,
                                                        const double kcon[4], const double kcon_avg[4],
                                                        double *connection_old, double gk_avg[4][4], double gk_new[4][4]) {
  for (int mu = 0; mu < 4; mu++)
    for (int beta = 0; beta < 4; beta++) gk_avg[mu][beta] = gk_new[mu][beta] = 0.0;
  if (st.ray_flat) {
#pragma unroll
    for (int mu = 0; mu < 4; mu++)
#pragma unroll
      for (int alpha = 0; alpha < 4; alpha++)
#pragma unroll
        for (int beta = 0; beta < 4; beta++) {
          const double g = 0.0;
          double *old = connection_old + ((mu * 4 + alpha) * 4 + beta) * 64;
          gk_new[mu][beta] += kcon[alpha] * g;
          gk_avg[mu][beta] += kcon_avg[alpha] * (0.5 * (*old + g));
          *old = g;
        }
    return;
  }
  const double bh_a = st.bh_a;
  const double a2 = ks.a2, rr2 = ks.rr2, r2 = ks.r2, r = ks.r, f = ks.f;
  const double l[4] = {-1.0, ks.l[0], ks.l[1], ks.l[2]};
  double gcon[4][4];
  for (int mu = 0; mu < 4; mu++)
    for (int nu = 0; nu < 4; nu++) gcon[mu][nu] = -f * l[mu] * l[nu];
  gcon[0][0] = -f * l[0] * l[0] - 1.0;
  gcon[1][1] = -f * l[1] * l[1] + 1.0;
  gcon[2][2] = -f * l[2] * l[2] + 1.0;
  gcon[3][3] = -f * l[3] * l[3] + 1.0;
  double dr[4], df[4], dl[4][4];   // derivatives with respect to x^a, a = 1..3
  dr[1] = r * x / (2.0 * r2 - rr2 + a2);
  dr[2] = r * y / (2.0 * r2 - rr2 + a2);
  dr[3] = (r * z + a2 * z / r) / (2.0 * r2 - rr2 + a2);
  df[1] = -(r2 * r2 - 3.0 * a2 * z * z) * dr[1] / (r * (r2 * r2 + a2 * z * z)) * f;
  df[2] = -(r2 * r2 - 3.0 * a2 * z * z) * dr[2] / (r * (r2 * r2 + a2 * z * z)) * f;
  df[3] = -((r2 * r2 - 3.0 * a2 * z * z) * dr[3] + 2.0 * a2 * r * z) / (r * (r2 * r2 + a2 * z * z)) * f;
  for (int a = 1; a < 4; a++) dl[0][a] = 0.0;
  dl[1][1] = ((x - 2.0 * r * l[1]) * dr[1] + r) / (r2 + a2);
  dl[1][2] = ((x - 2.0 * r * l[1]) * dr[2] + bh_a) / (r2 + a2);
  dl[1][3] = (x - 2.0 * r * l[1]) * dr[3] / (r2 + a2);
  dl[2][1] = ((y - 2.0 * r * l[2]) * dr[1] - bh_a) / (r2 + a2);
  dl[2][2] = ((y - 2.0 * r * l[2]) * dr[2] + r) / (r2 + a2);
  dl[2][3] = (y - 2.0 * r * l[2]) * dr[3] / (r2 + a2);
  dl[3][1] = -z / r2 * dr[1];
  dl[3][2] = -z / r2 * dr[2];
  dl[3][3] = -z / r2 * dr[3] + 1.0 / r;
  // d g_{mu nu} / d x^a = +-(df l_mu l_nu + f dl_mu l_nu + f l_mu dl_nu), minus when exactly one index is 0
  double dgcov[4][4][4];
  for (int mu = 0; mu < 4; mu++)
    for (int nu = 0; nu < 4; nu++) dgcov[0][mu][nu] = 0.0;
#pragma unroll
  for (int a = 1; a < 4; a++)
#pragma unroll
    for (int mu = 0; mu < 4; mu++)
#pragma unroll
      for (int nu = 0; nu < 4; nu++) {
        const double val = df[a] * l[mu] * l[nu] + f * dl[mu][a] * l[nu] + f * l[mu] * dl[nu][a];
        dgcov[a][mu][nu] = ((mu == 0) != (nu == 0)) ? -val : val;
      }
#pragma unroll
  for (int mu = 0; mu < 4; mu++)
#pragma unroll
    for (int alpha = 0; alpha < 4; alpha++)
#pragma unroll
      for (int beta = 0; beta < 4; beta++) {
        double g = 0.0;
#pragma unroll
        for (int nu = 0; nu < 4; nu++)
          g += 0.5 * gcon[mu][nu] * (dgcov[alpha][beta][nu] + dgcov[beta][alpha][nu] - dgcov[nu][alpha][beta]);
        double *old = connection_old + ((mu * 4 + alpha) * 4 + beta) * 64;
        gk_new[mu][beta] += kcon[alpha] * g;
        gk_avg[mu][beta] += kcon_avg[alpha] * (0.5 * (*old + g));
        *old = g;
      }
}

// The connection alone, for the first sample of a ray (its "previous" connection is its own: :150-154)
__device__ __forceinline__ void connection_first(const BlSpacetime &st, double x, double y, double z, const BlKerrSchild &ks,
                                                 double *connection_old) {
  const double zero[4] = {0.0, 0.0, 0.0, 0.0};
  double gk_a[4][4], gk_b[4][4];
  for (int c = 0; c < 64; c++) connection_old[c * 64] = 0.0;
  connection_contractions(st, x, y, z, ks, zero, zero, connection_old, gk_a, gk_b);
}

// Half a step of dN/dlambda = -(Gamma^mu_{alpha beta} k^alpha N^{beta nu} + (mu <-> nu)) (:181-198, :816-833), with
// gk[mu][beta] = k^alpha Gamma^mu_{alpha beta}: target += dN/dlambda(source) dl. The reference's two half steps are
// not symmetric: the first adds the derivative at the current N to N_temp - which still holds N as it was before
// the previous sample's second half step - and then sets N = N_temp; the second copies N to N_temp and adds the
// derivative to N. Both N and N_temp are therefore carried from sample to sample.
__device__ __forceinline__ void transport(const double gk[4][4], double dl, const Cplx nn[4][4], Cplx target[4][4]) {
  Cplx delta[4][4];
#pragma unroll
  for (int mu = 0; mu < 4; mu++)
#pragma unroll
    for (int nu = 0; nu < 4; nu++) {
      Cplx d = {0.0, 0.0};
#pragma unroll
      for (int beta = 0; beta < 4; beta++)
        d = csub(d, cadd(cscale(gk[mu][beta], nn[beta][nu]), cscale(gk[nu][beta], nn[mu][beta])));
      delta[mu][nu] = cscale(dl, d);   // complex * real: component-wise
    }
#pragma unroll
  for (int mu = 0; mu < 4; mu++)
#pragma unroll
    for (int nu = 0; nu < 4; nu++) target[mu][nu] = cadd(target[mu][nu], delta[mu][nu]);
}

// Stokes parameters of N^{mu nu} in a tetrad (:268-292, :904-932). N_{(a)(b)} = e_(a)^mu e_(b)^nu g g N is evaluated in
// the reference's nesting; only a, b in {1, 2} enter I, Q, U, V, so the last two contractions run over rows 1 and 2 of
// the tetrad only (the entries they skip are never read).
__device__ __forceinline__ void to_stokes(const double gcov[4][4], const double e1[4], const double e2[4], const Cplx nn_con[4][4],
                                          double ss[4]) {
  Cplx temp_c[4][4];
  {
    Cplx temp_b[4][4];
#pragma unroll
    for (int nu = 0; nu < 4; nu++)
#pragma unroll
      for (int alpha = 0; alpha < 4; alpha++) {
        Cplx acc = {0.0, 0.0};
#pragma unroll
        for (int beta = 0; beta < 4; beta++) acc = cadd(acc, cscale(gcov[nu][beta], nn_con[alpha][beta]));
        temp_b[nu][alpha] = acc;
      }
#pragma unroll
    for (int mu = 0; mu < 4; mu++)
#pragma unroll
      for (int nu = 0; nu < 4; nu++) {
        Cplx acc = {0.0, 0.0};
#pragma unroll
        for (int alpha = 0; alpha < 4; alpha++) acc = cadd(acc, cscale(gcov[mu][alpha], temp_b[nu][alpha]));
        temp_c[mu][nu] = acc;
      }
  }
  Cplx d1[4], d2[4];   // temp_d[b][mu] for b = 1, 2
#pragma unroll
  for (int mu = 0; mu < 4; mu++) {
    Cplx a1 = {0.0, 0.0}, a2 = {0.0, 0.0};
#pragma unroll
    for (int nu = 0; nu < 4; nu++) {
      a1 = cadd(a1, cscale(e1[nu], temp_c[mu][nu]));
      a2 = cadd(a2, cscale(e2[nu], temp_c[mu][nu]));
    }
    d1[mu] = a1;
    d2[mu] = a2;
  }
  Cplx n11 = {0.0, 0.0}, n12 = {0.0, 0.0}, n21 = {0.0, 0.0}, n22 = {0.0, 0.0};   // nn_tet_cov[a][b] = sum e_a[mu] temp_d[b][mu]
#pragma unroll
  for (int mu = 0; mu < 4; mu++) {
    n11 = cadd(n11, cscale(e1[mu], d1[mu]));
    n12 = cadd(n12, cscale(e1[mu], d2[mu]));
    n21 = cadd(n21, cscale(e2[mu], d1[mu]));
    n22 = cadd(n22, cscale(e2[mu], d2[mu]));
  }
  ss[0] = 0.5 * cadd(n11, n22).re;   // I 14
  ss[1] = 0.5 * csub(n11, n22).re;
  ss[2] = 0.5 * cadd(n12, n21).re;
  ss[3] = 0.5 * csub(n21, n12).im;
}

// Back to coordinates (:793-813): N^{mu nu} = e_(a)^mu e_(b)^nu N^{(a)(b)} with N^{(a)(b)} non-zero for a, b in {1, 2}
// only. The reference sums over all four a and b; its extra terms are products of a tetrad component with an exact
// (0, 0): +-0, which leave a sum that started at +0 unchanged (x + (+-0) = x, and (+0) + (-0) = +0), or NaN when that
// tetrad component is not finite - and then rows 1 and 2, which are built from rows 0 and 3, are NaN throughout and
// N is NaN throughout either way. "0.0 +" below is the reference's zero-initialised accumulator.
__device__ __forceinline__ void from_stokes(const double e1[4], const double e2[4], const double ss[4], Cplx nn_con[4][4]) {
  const Cplx t11 = {ss[0] + ss[1], 0.0};
  const Cplx t22 = {ss[0] - ss[1], 0.0};
  // ss_2 -+ i ss_3 with libstdc++'s real -+ complex: i * ss_3 = (0 * ss_3, 1 * ss_3)
  const Cplx t12 = {-(0.0 * ss[3]) + ss[2], -(1.0 * ss[3])};
  const Cplx t21 = {0.0 * ss[3] + ss[2], 1.0 * ss[3]};
  const Cplx zero = {0.0, 0.0};
  Cplx f1[4], f2[4];   // temp_f[nu][a] = sum_b e_b[nu] N^{(a)(b)} for a = 1, 2
#pragma unroll
  for (int nu = 0; nu < 4; nu++) {
    f1[nu] = cadd(cadd(zero, cscale(e1[nu], t11)), cscale(e2[nu], t12));
    f2[nu] = cadd(cadd(zero, cscale(e1[nu], t21)), cscale(e2[nu], t22));
  }
#pragma unroll
  for (int mu = 0; mu < 4; mu++)
#pragma unroll
    for (int nu = 0; nu < 4; nu++)
      nn_con[mu][nu] = cadd(cadd(zero, cscale(e1[mu], f1[nu])), cscale(e2[mu], f2[nu]));
}

struct Coupling {
  double j_s[4], alpha_s[4], rho_s[4];
  double alpha_sq, alpha_p, rho_sq, rho_p;
  double delta_lambda_cgs, delta_tau;
  bool optically_thin;
};

// The elementary functions of the coupling: the pinned library in the exact tier, the tolerant tier's in its sequential kernel
// (CouplingMathTolerant, further down). The formulas around them are the same text for both.
struct CouplingMathExact {
  static __device__ __forceinline__ double exp(double x) { return bl_exp(x); }
  static __device__ __forceinline__ double expm1(double x) { return bl_expm1(x); }
  static __device__ __forceinline__ double sinh(double x) { return bl_sinh(x); }
  static __device__ __forceinline__ double cosh(double x) { return bl_cosh(x); }
  static __device__ __forceinline__ double sin(double x) { return bl_sin(x); }
  static __device__ __forceinline__ double cos(double x) { return bl_cos(x); }
};

// Emission and absorption over a length dl without rotation (I A14-A17 and its limits; :391-451, :580-654)
template <typename M>
__device__ void absorb(const Coupling &c, double dl, double dtau, const double ss_start[4], double ss_end[4]) {
  const double *j_s = c.j_s, *alpha_s = c.alpha_s;
  if (alpha_s[0] == 0.0) {
    for (int a = 0; a < 4; a++) ss_end[a] = ss_start[a] + j_s[a] * dl;
  } else if (c.alpha_p == 0.0) {
    if (c.optically_thin) {
      const double exp_neg = M::exp(-dtau);
      const double expm1 = M::expm1(dtau);
      for (int a = 0; a < 4; a++) ss_end[a] = exp_neg * (ss_start[a] + j_s[a] / alpha_s[0] * expm1);
    } else {
      for (int a = 0; a < 4; a++) ss_end[a] = j_s[a] / alpha_s[0];
    }
  } else if (c.optically_thin) {
    const double alpha_p = c.alpha_p, alpha_sq = c.alpha_sq;
    const double exp_neg_i = M::exp(-dtau);
    const double exp_neg_p = M::exp(-alpha_p * dl);
    const double sinh_p = M::sinh(alpha_p * dl);
    const double cosh_p = M::cosh(alpha_p * dl);
    const double coshm1_p = 0.5 * (M::expm1(alpha_p * dl) + exp_neg_p - 1.0);
    const double alpha_ss = alpha_s[1] * ss_start[1] + alpha_s[3] * ss_start[3];
    const double alpha_j = alpha_s[1] * j_s[1] + alpha_s[3] * j_s[3];
    const double alpha_i_p_factor = 1.0 / (alpha_s[0] * alpha_s[0] - alpha_sq);
    ss_end[0] = (ss_start[0] * cosh_p - alpha_ss / alpha_p * sinh_p) * exp_neg_i
        + alpha_j * alpha_i_p_factor * (-1.0 + (alpha_s[0] * sinh_p + alpha_p * cosh_p) / alpha_p * exp_neg_p)
        + alpha_s[0] * j_s[0] * alpha_i_p_factor * (1.0 - (alpha_s[0] * cosh_p + alpha_p * sinh_p) / alpha_s[0] * exp_neg_p);
    for (int a = 1; a < 4; a++) {
      const double term_1 = (ss_start[a] + alpha_s[a] * alpha_ss / alpha_sq * coshm1_p
          - ss_start[0] * alpha_s[a] / alpha_p * sinh_p) * exp_neg_i;
      const double term_2 = j_s[a] * (1.0 - exp_neg_i) / alpha_s[0];
      const double term_3 = alpha_j * alpha_s[a] / alpha_s[0] * alpha_i_p_factor * (1.0 - (1.0
          - alpha_s[0] * alpha_s[0] / alpha_sq - alpha_s[0] / alpha_sq * (alpha_s[0] * cosh_p + alpha_p * sinh_p)) * exp_neg_i);
      const double term_4 = j_s[0] * alpha_s[a] / alpha_p * alpha_i_p_factor * (-alpha_p
          + (alpha_p * cosh_p + alpha_s[0] * sinh_p) * exp_neg_i);
      ss_end[a] = term_1 + term_2 + term_3 + term_4;
    }
  } else {
    const double alpha_j = alpha_s[1] * j_s[1] + alpha_s[3] * j_s[3];
    ss_end[0] = (alpha_s[0] * j_s[0] - alpha_j) / (alpha_s[0] * alpha_s[0] - c.alpha_sq);
    for (int a = 1; a < 4; a++) ss_end[a] = (j_s[a] - alpha_s[a] * ss_end[0]) / alpha_s[0];
  }
}

// Faraday rotation and conversion over the whole step without absorption (I A2-A5; :470-486, :598-612)
template <typename M>
__device__ void rotate(const Coupling &c, const double ss_start[4], double ss_end[4]) {
  const double *rho_s = c.rho_s;
  const double cos_rho = M::cos(c.rho_p * c.delta_lambda_cgs);
  const double sin_rho = M::sin(c.rho_p * c.delta_lambda_cgs);
  double sin_sq_rho = M::sin(c.rho_p * c.delta_lambda_cgs / 2.0);
  sin_sq_rho = sin_sq_rho * sin_sq_rho;
  const double rho_ss = rho_s[1] * ss_start[1] + rho_s[3] * ss_start[3];
  ss_end[0] = ss_start[0];
  ss_end[1] = ss_start[1] * cos_rho + 2.0 * rho_s[1] * rho_ss / c.rho_sq * sin_sq_rho - rho_s[3] * ss_start[2] / c.rho_p * sin_rho;
  ss_end[2] = ss_start[2] * cos_rho + (rho_s[3] * ss_start[1] - rho_s[1] * ss_start[3]) / c.rho_p * sin_rho;
  ss_end[3] = ss_start[3] * cos_rho + 2.0 * rho_s[3] * rho_ss / c.rho_sq * sin_sq_rho + rho_s[1] * ss_start[2] / c.rho_p * sin_rho;
}

__device__ void limit_polarization(double ss[4]) {
  const double ss_pol = ss[1] * ss[1] + ss[2] * ss[2] + ss[3] * ss[3];
  if (ss_pol > ss[0] * ss[0]) {
    const double factor = blm_sqrt(ss[0] * ss[0] / ss_pol);
    ss[1] *= factor;
    ss[2] *= factor;
    ss[3] *= factor;
  }
}

// Absorption and rotation together (L 10, I 24; :657-778). The coupling matrices are built exactly as the
// reference writes them: entry [1][2] of matrices 2 and 3 is assigned twice and [0][2], [1][3], [2][3] stay zero.
template <typename M>
__device__ void couple_jointly(const Coupling &c, const double ss_start[4], double ss_end[4]) {
  const double *j_s = c.j_s, *alpha_s = c.alpha_s, *rho_s = c.rho_s;
  const double alpha_sq = c.alpha_sq, rho_sq = c.rho_sq;
  const double alpha_rho = alpha_s[1] * rho_s[1] + alpha_s[3] * rho_s[3];
  const double alpha_sq_rho_sq = alpha_sq - rho_sq;
  const double lambda_a = blm_sqrt(alpha_sq_rho_sq * alpha_sq_rho_sq / 4.0 + alpha_rho * alpha_rho);
  const double lambda_b = alpha_sq_rho_sq / 2.0;
  const double lambda_1 = blm_sqrt(lambda_a + lambda_b);
  const double lambda_2 = blm_sqrt(lambda_a - lambda_b);
  const double coefficient_theta = lambda_1 * lambda_1 + lambda_2 * lambda_2;
  const double sg = alpha_rho >= 0.0 ? 1.0 : -1.0;
  double mm_1[4][4], mm_2[4][4], mm_3[4][4], mm_4[4][4];
  for (int a = 0; a < 4; a++)
    for (int b = 0; b < 4; b++) {
      mm_1[a][b] = a == b ? 1.0 : 0.0;
      mm_2[a][b] = mm_3[a][b] = mm_4[a][b] = 0.0;
    }
  mm_2[0][1] = lambda_2 * alpha_s[1] - sg * lambda_1 * rho_s[1];
  mm_2[0][3] = lambda_2 * alpha_s[3] - sg * lambda_1 * rho_s[3];
  mm_2[1][2] = sg * lambda_1 * alpha_s[1] + lambda_2 * rho_s[1];
  mm_2[1][0] = mm_2[0][1];
  mm_2[2][0] = mm_2[0][2];
  mm_2[3][0] = mm_2[0][3];
  mm_2[2][1] = -mm_2[1][2];
  mm_2[3][1] = -mm_2[1][3];
  mm_2[3][2] = -mm_2[2][3];
  for (int a = 0; a < 4; a++)
    for (int b = 0; b < 4; b++) mm_2[a][b] *= 1.0 / coefficient_theta;
  mm_3[0][1] = lambda_1 * alpha_s[1] + sg * lambda_2 * rho_s[1];
  mm_3[0][3] = lambda_1 * alpha_s[3] + sg * lambda_2 * rho_s[3];
  mm_3[1][2] = -(sg * lambda_2 * alpha_s[1] - lambda_1 * rho_s[1]);
  mm_3[1][0] = mm_3[0][1];
  mm_3[2][0] = mm_3[0][2];
  mm_3[3][0] = mm_3[0][3];
  mm_3[2][1] = -mm_3[1][2];
  mm_3[3][1] = -mm_3[1][3];
  mm_3[3][2] = -mm_3[2][3];
  for (int a = 0; a < 4; a++)
    for (int b = 0; b < 4; b++) mm_3[a][b] *= 1.0 / coefficient_theta;
  mm_4[0][0] = (alpha_sq + rho_sq) / 2.0;
  mm_4[1][1] = alpha_s[1] * alpha_s[1] + rho_s[1] * rho_s[1] - (alpha_sq + rho_sq) / 2.0;
  mm_4[2][2] = -(alpha_sq + rho_sq) / 2.0;
  mm_4[3][3] = alpha_s[3] * alpha_s[3] + rho_s[3] * rho_s[3] - (alpha_sq + rho_sq) / 2.0;
  mm_4[0][2] = alpha_s[1] * rho_s[3] - alpha_s[3] * rho_s[1];
  mm_4[1][3] = alpha_s[3] * alpha_s[1] + rho_s[3] * rho_s[1];
  mm_4[1][0] = -mm_4[0][1];
  mm_4[2][0] = -mm_4[0][2];
  mm_4[3][0] = -mm_4[0][3];
  mm_4[2][1] = mm_4[1][2];
  mm_4[3][1] = mm_4[1][3];
  mm_4[3][2] = mm_4[2][3];
  for (int a = 0; a < 4; a++)
    for (int b = 0; b < 4; b++) mm_4[a][b] *= 2.0 / coefficient_theta;
  double exp_v = 0.0, sin_v = 0.0, cos_v = 0.0, sinh_v = 0.0, cosh_v = 0.0;
  if (c.optically_thin) {
    exp_v = M::exp(-c.delta_tau);
    sin_v = M::sin(lambda_2 * c.delta_lambda_cgs);
    cos_v = M::cos(lambda_2 * c.delta_lambda_cgs);
    sinh_v = M::sinh(lambda_1 * c.delta_lambda_cgs);
    cosh_v = M::cosh(lambda_1 * c.delta_lambda_cgs);
  }
  const double f_1 = 1.0 / (alpha_s[0] * alpha_s[0] - lambda_1 * lambda_1);
  const double f_2 = 1.0 / (alpha_s[0] * alpha_s[0] + lambda_2 * lambda_2);
  for (int a = 0; a < 4; a++) ss_end[a] = 0.0;
  for (int a = 0; a < 4; a++)
    for (int b = 0; b < 4; b++) {
      const double cosh_term = -lambda_1 * f_1 * mm_3[a][b] + 0.5 * alpha_s[0] * f_1 * (mm_1[a][b] + mm_4[a][b]);
      const double cos_term = -lambda_2 * f_2 * mm_2[a][b] + 0.5 * alpha_s[0] * f_2 * (mm_1[a][b] - mm_4[a][b]);
      double pp = cosh_term + cos_term;
      if (c.optically_thin) {
        const double sin_term = -alpha_s[0] * f_2 * mm_2[a][b] - 0.5 * lambda_2 * f_2 * (mm_1[a][b] - mm_4[a][b]);
        const double sinh_term = -alpha_s[0] * f_1 * mm_3[a][b] + 0.5 * lambda_1 * f_1 * (mm_1[a][b] + mm_4[a][b]);
        pp -= exp_v * (cosh_term * cosh_v + cos_term * cos_v + sin_term * sin_v + sinh_term * sinh_v);
        const double oo = exp_v * (0.5 * (mm_1[a][b] + mm_4[a][b]) * cosh_v + 0.5 * (mm_1[a][b] - mm_4[a][b]) * cos_v
            - mm_2[a][b] * sin_v - mm_3[a][b] * sinh_v);
        ss_end[a] += pp * j_s[b] + oo * ss_start[b];
      } else {
        ss_end[a] += pp * j_s[b];
      }
    }
}

// The coupling of one sample (:388-779) with its guards (:781-790): rotation split from emission / absorption, or the
// analytic cases. c holds the coefficients on entry; the derived quantities are filled in here.
template <typename M>
__device__ __forceinline__ void couple_sample(Coupling *cp, bool rotation_split, double delta_lambda_cgs, double ss_start[4], double ss_end[4]) {
  Coupling &c = *cp;
  c.delta_lambda_cgs = delta_lambda_cgs;
  c.delta_tau = c.alpha_s[0] * delta_lambda_cgs;
  c.optically_thin = c.delta_tau <= kDeltaTauMax;
  c.alpha_sq = c.alpha_s[1] * c.alpha_s[1] + c.alpha_s[3] * c.alpha_s[3];
  c.alpha_p = blm_sqrt(c.alpha_sq);
  c.rho_sq = c.rho_s[1] * c.rho_s[1] + c.rho_s[3] * c.rho_s[3];
  c.rho_p = blm_sqrt(c.rho_sq);
  if (rotation_split) {   // :388-568
    absorb<M>(c, delta_lambda_cgs / 2.0, c.delta_tau / 2.0, ss_start, ss_end);
    ss_end[0] = (ss_end[0] < 0.0) ? 0.0 : ss_end[0];   // std::max(ss_end[0], 0.0)
    limit_polarization(ss_end);
    for (int a = 0; a < 4; a++) ss_start[a] = ss_end[a];
    if (c.rho_p != 0.0) rotate<M>(c, ss_start, ss_end);
    limit_polarization(ss_end);
    for (int a = 0; a < 4; a++) ss_start[a] = ss_end[a];
    absorb<M>(c, delta_lambda_cgs / 2.0, c.delta_tau / 2.0, ss_start, ss_end);
  } else if (c.alpha_s[0] == 0.0 && c.rho_p == 0.0) {
    for (int a = 0; a < 4; a++) ss_end[a] = ss_start[a] + c.j_s[a] * delta_lambda_cgs;
  } else if (c.alpha_p == 0.0 && c.rho_p == 0.0) {
    absorb<M>(c, delta_lambda_cgs, c.delta_tau, ss_start, ss_end);
  } else if (c.alpha_s[0] == 0.0) {
    rotate<M>(c, ss_start, ss_end);
    for (int a = 0; a < 4; a++) ss_end[a] += c.j_s[a] * delta_lambda_cgs;
  } else if (c.rho_p == 0.0) {
    absorb<M>(c, delta_lambda_cgs, c.delta_tau, ss_start, ss_end);
  } else {
    couple_jointly<M>(c, ss_start, ss_end);
  }
  // std::max(ss_end[0], 0.0) (:781): (a < b) ? b : a, so a NaN intensity stays NaN
  ss_end[0] = (ss_end[0] < 0.0) ? 0.0 : ss_end[0];
  limit_polarization(ss_end);
}

}  // namespace

__global__ void __launch_bounds__(64) bl_transfer_polarized_kernel(BlTransferArgs P) {
  // The previous sample's connection, one column of 64 components per lane ([component][lane]: every access of a
  // wave touches 64 consecutive doubles): 32 KiB per wave, which leaves the registers to N, N_temp and the step
  __shared__ double connection_lds[64 * 64];
  double *connection_old = connection_lds + threadIdx.x;
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= bl_rays_done(P.counters, P.chunk_rays)) return;
  const BlSpacetime st = P.st;
  const int num = P.ray_sample_num[slot];
  const long long out_index = P.ray_out_index[slot];
  const double momentum_factor = P.ray_factor[slot];
  const size_t row = (size_t)P.n_rays_total;
  double *img = P.image + out_index;
  const BlPolSample *samples = P.pol_samples + (size_t)P.ray_offset[slot];
  const double2 *pc = P.pol_coeffs + (size_t)P.ray_offset[slot] * P.n_nu * 4;
  for (int l = 0; l < P.n_nu; l++) {
    const double freq = P.frequencies[l];
    if (num <= 0) {   // :94-96: nothing integrated; rows stay as the auxiliary kernel zeroed them
      for (int a = 0; a < 4; a++) img[(size_t)(4 * l + a) * row] = 0.0;
      continue;
    }
    // N is carried from sample to sample. N_temp is not: after a sample it is what from_stokes made of that sample's final
    // Stokes parameters and tetrad (the second half step copies N to N_temp before it advances N), so it is rebuilt from
    // those twelve numbers - the same operations on the same operands, the same bits - instead of living in 64 registers.
    Cplx nn_con[4][4];
    for (int mu = 0; mu < 4; mu++)
      for (int nu = 0; nu < 4; nu++) nn_con[mu][nu] = Cplx{0.0, 0.0};
    double ss_carried[4] = {0.0, 0.0, 0.0, 0.0}, e1_carried[4] = {0.0, 0.0, 0.0, 0.0}, e2_carried[4] = {0.0, 0.0, 0.0, 0.0};
    bool first_sample = true;
    // reference sample order is reversed integration order (geodesics.cpp:832-840): its n = 0 is record num - 1.
    // At n = 0 the "previous" connection and k^mu are the sample's own (:150-154, :167-169); averaging a value
    // with itself returns it, so the loop below needs no first-sample case.
    double delta_lambda_old = 0.0, tau = 0.0;
    double kcon_old[4];
    {
      const BlPolSample s = samples[num - 1];
      BlKerrSchild ks;
      bl_kerr_schild(st, s.x[0], s.x[1], s.x[2], &ks);
      connection_first(st, s.x[0], s.x[1], s.x[2], ks, connection_old);
      for (int mu = 0; mu < 4; mu++) kcon_old[mu] = s.kcon[mu];
    }
    for (int rec = num - 1; rec >= 0; rec--) {
      const BlPolSample s = samples[rec];
      const double delta_lambda = s.delta_lambda;
      const double delta_lambda_new = rec > 0 ? samples[rec - 1].delta_lambda : delta_lambda;
      const double delta_lambda_cgs = delta_lambda * P.x_unit / (freq * momentum_factor);
      const double x1 = s.x[0], x2 = s.x[1], x3 = s.x[2];
      Coupling c;
      {
        const size_t at = (size_t)rec * P.n_nu + l;
        const double2 c0 = pc[at * 4 + 0], c1 = pc[at * 4 + 1], c2 = pc[at * 4 + 2], c3 = pc[at * 4 + 3];
        c.j_s[0] = c0.x; c.j_s[1] = c1.x; c.j_s[2] = 0.0; c.j_s[3] = c1.y;
        c.alpha_s[0] = c0.y; c.alpha_s[1] = c2.x; c.alpha_s[2] = 0.0; c.alpha_s[3] = c2.y;
        c.rho_s[0] = 0.0; c.rho_s[1] = c3.x; c.rho_s[2] = 0.0; c.rho_s[3] = c3.y;
      }

      BlKerrSchild ks;
      bl_kerr_schild(st, x1, x2, x3, &ks);
      double kcon[4], kcon_avg[4];
      for (int mu = 0; mu < 4; mu++) {
        kcon[mu] = s.kcon[mu];
        kcon_avg[mu] = 0.5 * (kcon_old[mu] + kcon[mu]);
      }
      double gk_avg[4][4], gk_new[4][4];
      connection_contractions(st, x1, x2, x3, ks, kcon, kcon_avg, connection_old, gk_avg, gk_new);

      // first half step (:171-198)
      {
        Cplx nn_con_temp[4][4];
        if (first_sample) {
          for (int mu = 0; mu < 4; mu++)
            for (int nu = 0; nu < 4; nu++) nn_con_temp[mu][nu] = Cplx{0.0, 0.0};
        } else {
          from_stokes(e1_carried, e2_carried, ss_carried, nn_con_temp);
        }
        transport(gk_avg, (delta_lambda_old + delta_lambda) / 2.0, nn_con, nn_con_temp);
        for (int mu = 0; mu < 4; mu++)
          for (int nu = 0; nu < 4; nu++) nn_con[mu][nu] = nn_con_temp[mu][nu];
      }

      // fluid frame: Stokes parameters before the coupling (:268-292)
      double ss_start[4], ss_end[4] = {0.0, 0.0, 0.0, 0.0};
      {
        double gcov[4][4];
        if (st.ray_flat)
          bl_minkowski(gcov);
        else
          bl_gcov_ks(ks, gcov);
        to_stokes(gcov, s.e1, s.e2, nn_con, ss_start);
      }

      couple_sample<CouplingMathExact>(&c, P.rotation_split != 0, delta_lambda_cgs, ss_start, ss_end);
      tau += c.delta_tau;   // unpolarized.cpp:139-140 (BlAuxImages::polarized_rows_only: written below)

      // back to coordinates (:793-813), second half step (:816-833)
      {
        Cplx nn_con_temp[4][4];
        from_stokes(s.e1, s.e2, ss_end, nn_con_temp);
        for (int mu = 0; mu < 4; mu++)
          for (int nu = 0; nu < 4; nu++) nn_con[mu][nu] = nn_con_temp[mu][nu];
        transport(gk_new, (delta_lambda + delta_lambda_new) / 4.0, nn_con_temp, nn_con);
      }
      for (int a = 0; a < 4; a++) {
        ss_carried[a] = ss_end[a];
        e1_carried[a] = s.e1[a];
        e2_carried[a] = s.e2[a];
      }
      first_sample = false;

      delta_lambda_old = delta_lambda;
      for (int mu = 0; mu < 4; mu++) kcon_old[mu] = kcon[mu];
    }

    // camera frame (:875-939) and nu^3 (:942-949)
    {
      const double *cp = P.camera_pos + 4 * out_index, *cd = P.camera_dir + 4 * out_index;
      const double kcov[4] = {cd[0], cd[1], cd[2], cd[3]};
      double gcov[4][4], gcon[4][4], kcon[4], up_con[4], tetrad[4][4];
      bl_pol::geodesic_metric(st, cp[1], cp[2], cp[3], gcov, gcon);
      for (int mu = 0; mu < 4; mu++) {
        double acc = 0.0;
        for (int nu = 0; nu < 4; nu++) acc += gcon[mu][nu] * kcov[nu];
        kcon[mu] = acc;
      }
      const double *u_con = P.cam_u_con, *u_cov = P.cam_u_cov, *vert = P.cam_vert_con_c;
      up_con[0] = u_con[0] * vert[0] - (u_cov[1] * vert[1] + u_cov[2] * vert[2] + u_cov[3] * vert[3]) / u_cov[0];
      up_con[1] = vert[1] + u_con[1] * vert[0];
      up_con[2] = vert[2] + u_con[2] * vert[0];
      up_con[3] = vert[3] + u_con[3] * vert[0];
      bl_pol::tetrad_frame(u_con, u_cov, kcon, kcov, up_con, gcov, gcon, tetrad);
      double ss[4];
      to_stokes(gcov, tetrad[1], tetrad[2], nn_con, ss);
      const double nu_cu = freq * freq * freq;
      for (int a = 0; a < 4; a++) img[(size_t)(4 * l + a) * row] = ss[a] * nu_cu;
      if (P.aux_images.polarized_rows_only && P.aux_images.image_tau) img[(size_t)(P.aux_images.offset_tau + l) * row] = tau;
    }
  }
}

extern "C" hipError_t bl_launch_transfer_polarized(const BlTransferArgs *args, hipStream_t stream) {
  const int grid = (args->chunk_rays + 63) / 64;
  hipLaunchKernelGGL(bl_transfer_polarized_kernel, dim3(grid), dim3(64), 0, stream, *args);
  return hipGetLastError();
}

// =====================================================================================================================
// Tolerant arithmetic tier (bl_set_arithmetic(BL_ARITH_TOLERANT)): transport matrices.
//
// Between two couplings everything the reference does to a ray's state is linear in the four Stokes parameters the
// previous coupling left: from_stokes with the previous sample's tetrad (N0 = E_p T E_p^T, T the 2 x 2 Hermitian matrix of
// I, Q, U, V), the previous sample's second half step (N1 = N0 + a D_1(N0), D(N) = -(G N + N G^T), G_1 = k^alpha Gamma at
// the previous sample), this sample's first half step, which adds to N_temp = N0 (N2 = N0 + b D_2(N1), G_2 from the
// averaged connection and momentum), and to_stokes with this sample's metric and tetrad. So
//     ss_start(n) = M_n ss_end(n - 1)
// with a real 4 x 4 matrix M_n that depends on the geometry of samples n - 1 and n only: sample-parallel work. M_n is built
// without ever forming N: with covectors c_a = g e'_a, u_a = G_2^T c_a and the projections P(x) = (x . e_1, x . e_2),
// R(x) = P(G_1^T x) onto the previous tetrad, x^T N1 y = P(x)^T T P(y) - a [R(x)^T T P(y) + P(x)^T T R(y)], and
// n_ab = c_a^T N2 c_b = sum_cd K_ab^cd t_cd with K_ab = P(c_a) P(c_b)^T - b [X(u_a, c_b) + X(c_a, u_b)]. K is real, so
// (I, Q, U) and V decouple: M is a 3 x 3 block and one number (10 doubles per sample). The contraction
// G^mu_beta = k^alpha Gamma^mu_{alpha beta} of the Kerr-Schild metric g = eta + f l l is evaluated in closed form
// (s = k.df, v = (k.d) l, w_beta = k^alpha d_beta l_alpha, lambda = l.k; Q = s l l^T + f (v l^T + l v^T) + T2 - T2^T with
// T2 = lambda (df l^T + f dl) + f l w^T; G = (eta Q - f l^up (l^up Q)) / 2) instead of through the 64 components, and only
// ever applied to covectors (apply_transposed), never stored.
// bl_transport_matrix_kernel does that for every sample at full occupancy; what stays sequential per ray
// (bl_transfer_polarized_matrix_kernel) is ss = M ss, the coupling and its guards - the same coupling code as the exact tier.
// The same mathematics as polarized.cpp:150-198, :259-292, :793-833 in another association of the floating-point
// operations: images agree with the exact tier to ~1e-13 of the peak intensity (tests/test_gpu_tolerant.py), NaN masks and
// all integer results are the exact tier's. Not used with ray_flat (the exact kernel serves that).
#pragma clang fp contract(fast)

namespace {
#include "bl_fastmath.h"

// exp, expm1 and the hyperbolic functions of the coupling from the tier's exponential core; sine and cosine stay the pinned
// ones (Faraday-thick samples hand them arguments of any size: they need the real range reduction)
struct CouplingMathTolerant {
  static __device__ __forceinline__ double exp(double x) { return fastmath::exp(x); }
  static __device__ __forceinline__ double expm1(double x) { return fastmath::expm1(x); }
  static __device__ __forceinline__ double sinh(double x) {   // (e^x - e^-x) / 2 through expm1: no cancellation near 0
    const double em = fastmath::expm1(x);
    const double inv = fastmath::rcp(em + 1.0);                     // e^-x; 0 once e^x has overflowed
    return 0.5 * (em + (inv == 0.0 ? 1.0 : em * inv));              // (e^x - 1) + (1 - e^-x)
  }
  static __device__ __forceinline__ double cosh(double x) {
    const double e = fastmath::exp(x);
    return 0.5 * (e + fastmath::rcp(e));
  }
  static __device__ __forceinline__ double sin(double x) { return bl_sin(x); }
  static __device__ __forceinline__ double cos(double x) { return bl_cos(x); }
};

namespace fastpol {
using fastmath::rcp;

// f, l_i and their spatial derivatives at a point (radiation_geometry.cpp:283-330 in closed form with shared reciprocals)
struct PointGeometry {
  double f;
  double l[3];       // l_1..l_3 (l_0 = 1 covariant, -1 contravariant)
  double df[3];      // d f / d x^a
  double dl[3][3];   // dl[i][a] = d l_{i+1} / d x^{a+1}
};

__device__ __forceinline__ PointGeometry point_geometry(const BlSpacetime &st, double x, double y, double z) {
  BlKerrSchild ks;
  bl_kerr_schild(st, x, y, z, &ks);
  PointGeometry g;
  const double a = st.bh_a, a2 = ks.a2, r = ks.r, r2 = ks.r2;
  g.f = ks.f;
  g.l[0] = ks.l[0]; g.l[1] = ks.l[1]; g.l[2] = ks.l[2];
  const double inv_d = rcp(2.0 * r2 - ks.rr2 + a2), inv_r = rcp(r), inv_ra = rcp(r2 + a2);
  const double q = r2 * r2, az2 = a2 * z * z;
  const double inv_f = rcp(r * (q + az2));
  const double dr[3] = {r * x * inv_d, r * y * inv_d, (r * z + a2 * z * inv_r) * inv_d};
  const double c1 = -(q - 3.0 * az2) * g.f * inv_f;
  g.df[0] = c1 * dr[0];
  g.df[1] = c1 * dr[1];
  g.df[2] = c1 * dr[2] - 2.0 * a2 * r * z * g.f * inv_f;
  const double px = x - 2.0 * r * g.l[0], py = y - 2.0 * r * g.l[1], mz = -z * inv_r * inv_r;
  g.dl[0][0] = (px * dr[0] + r) * inv_ra;
  g.dl[0][1] = (px * dr[1] + a) * inv_ra;
  g.dl[0][2] = px * dr[2] * inv_ra;
  g.dl[1][0] = (py * dr[0] - a) * inv_ra;
  g.dl[1][1] = (py * dr[1] + r) * inv_ra;
  g.dl[1][2] = py * dr[2] * inv_ra;
  g.dl[2][0] = mz * dr[0];
  g.dl[2][1] = mz * dr[1];
  g.dl[2][2] = mz * dr[2] + inv_r;
  return g;
}

// G[mu][beta] = k^alpha Gamma^mu_{alpha beta} at a point is never formed: only y = G^T x is needed, for a handful of covectors
// x. With z = g^{-1} x (z^0 = -x_0 + f (l^up . x), z^i = x_i - f l_i (l^up . x)) and the scalars of the momentum
// (s = k . df, lambda = l . k, v_i = k^a d_a l_i, w_b = k^i d_b l_i):
//   h = s (z.l) + f (z.v) - lambda (z.df) - f (z.w),   y_0 = h / 2,
//   y_i = (l_i h + f (z.l) (v_i + w_i) + lambda (df_i (z.l) + f ((z.dl)_i - (dl z)_i))) / 2.
struct Contracted {
  double s, lambda, v[3], w[3];
};
__device__ __forceinline__ Contracted contract_momentum(const PointGeometry &g, const double k[4]) {
  Contracted c;
  c.s = k[1] * g.df[0] + k[2] * g.df[1] + k[3] * g.df[2];
  c.lambda = k[0] + k[1] * g.l[0] + k[2] * g.l[1] + k[3] * g.l[2];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    c.v[i] = k[1] * g.dl[i][0] + k[2] * g.dl[i][1] + k[3] * g.dl[i][2];
    c.w[i] = k[1] * g.dl[0][i] + k[2] * g.dl[1][i] + k[3] * g.dl[2][i];
  }
  return c;
}
__device__ __forceinline__ void apply_transposed(const PointGeometry &g, const Contracted &c, const double x[4], double y[4]) {
  const double lux = -x[0] + g.l[0] * x[1] + g.l[1] * x[2] + g.l[2] * x[3];
  const double flux = g.f * lux;
  const double z0 = flux - x[0];
  const double z[3] = {x[1] - flux * g.l[0], x[2] - flux * g.l[1], x[3] - flux * g.l[2]};
  const double zl = z0 + z[0] * g.l[0] + z[1] * g.l[1] + z[2] * g.l[2];
  const double zv = z[0] * c.v[0] + z[1] * c.v[1] + z[2] * c.v[2];
  const double zw = z[0] * c.w[0] + z[1] * c.w[1] + z[2] * c.w[2];
  const double zdf = z[0] * g.df[0] + z[1] * g.df[1] + z[2] * g.df[2];
  const double h = c.s * zl + g.f * (zv - zw) - c.lambda * zdf;
  y[0] = 0.5 * h;
  const double fzl = g.f * zl;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const double zdl = z[0] * g.dl[0][i] + z[1] * g.dl[1][i] + z[2] * g.dl[2][i];
    const double dlz = g.dl[i][0] * z[0] + g.dl[i][1] * z[1] + g.dl[i][2] * z[2];
    y[i + 1] = 0.5 * (g.l[i] * h + fzl * (c.v[i] + c.w[i]) + c.lambda * (g.df[i] * zl + g.f * (zdl - dlz)));
  }
}

// P(x) and R(x) = P(G1^T x) of a covector x for the previous sample's tetrad rows e1, e2
struct Projected {
  double p[2], r[2];
};
__device__ __forceinline__ Projected project(const double x[4], const double e1[4], const double e2[4], const PointGeometry &g1,
                                             const Contracted &c1) {
  Projected o;
  o.p[0] = x[0] * e1[0] + x[1] * e1[1] + x[2] * e1[2] + x[3] * e1[3];
  o.p[1] = x[0] * e2[0] + x[1] * e2[1] + x[2] * e2[2] + x[3] * e2[3];
  double gx[4];
  apply_transposed(g1, c1, x, gx);
  o.r[0] = gx[0] * e1[0] + gx[1] * e1[1] + gx[2] * e1[2] + gx[3] * e1[3];
  o.r[1] = gx[0] * e2[0] + gx[1] * e2[1] + gx[2] * e2[2] + gx[3] * e2[3];
  return o;
}
// X(x, y)^{cd}: the coefficients of t_cd in x^T N1 y, N1 = N0 + a D_1(N0)
__device__ __forceinline__ void after_first_transport(const Projected &x, const Projected &y, double a, double X[2][2]) {
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int d = 0; d < 2; d++) X[c][d] = x.p[c] * y.p[d] - a * (x.r[c] * y.p[d] + x.p[c] * y.r[d]);
}
// K_ab^{cd} (coefficients of t_cd in n_ab) -> the 3 x 3 (I, Q, U) block, rows first, and the V number: m[0..8], m[9]
__device__ __forceinline__ void stokes_matrix(const double K[2][2][2][2], double m[10]) {
  double A[2][2], B[2][2], C[2][2], D[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++) {
      A[a][b] = K[a][b][0][0] + K[a][b][1][1];
      B[a][b] = K[a][b][0][0] - K[a][b][1][1];
      C[a][b] = K[a][b][0][1] + K[a][b][1][0];
      D[a][b] = K[a][b][1][0] - K[a][b][0][1];
    }
  m[0] = 0.5 * (A[0][0] + A[1][1]); m[1] = 0.5 * (B[0][0] + B[1][1]); m[2] = 0.5 * (C[0][0] + C[1][1]);
  m[3] = 0.5 * (A[0][0] - A[1][1]); m[4] = 0.5 * (B[0][0] - B[1][1]); m[5] = 0.5 * (C[0][0] - C[1][1]);
  m[6] = 0.5 * (A[0][1] + A[1][0]); m[7] = 0.5 * (B[0][1] + B[1][0]); m[8] = 0.5 * (C[0][1] + C[1][0]);
  m[9] = 0.5 * (D[1][0] - D[0][1]);
}

__device__ __forceinline__ void load_sample(const BlPolSample *s, double x[3], double *delta_lambda, double k[4], double e1[4], double e2[4]) {
  const double2 *q = reinterpret_cast<const double2 *>(s);
  const double2 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4], q5 = q[5], q6 = q[6], q7 = q[7];
  x[0] = q0.x; x[1] = q0.y; x[2] = q1.x; *delta_lambda = q1.y;
  k[0] = q2.x; k[1] = q2.y; k[2] = q3.x; k[3] = q3.y;
  e1[0] = q4.x; e1[1] = q4.y; e1[2] = q5.x; e1[3] = q5.y;
  e2[0] = q6.x; e2[1] = q6.y; e2[2] = q7.x; e2[3] = q7.y;
}

}  // namespace fastpol
}  // namespace

// One lane per sample, ray by ray: a wave takes 64 consecutive samples of one ray (persistent waves over the rays). A lane
// reading its own 128-byte record straight from memory makes every load instruction touch 64 different lines, and two waves
// per SIMD do not hide a trip to memory on their own; so the records of a segment (64 + the first of the next segment: the
// previous sample of lane i is lane i + 1's) travel as coalesced 16-byte units by LDS-DMA (global_load_lds_dwordx4: no
// registers in between) into one of two wave-private LDS tiles while the arithmetic of the segment before runs out of the
// other. The DMA writes LDS lane-linearly, so the tile is swizzled on the source side: unit r of sample i sits in slot
// 8 i + ((r + i) & 7), which spreads the 16-byte LDS reads of consecutive lanes over all banks. The 96-byte results go back
// through the tile they came from (rows padded to 112 bytes) and leave as coalesced 16-byte stores.
constexpr int kMkTileBytes = 9 * 1024;      // nine DMA instructions of 64 x 16 bytes: 65 records and slack
constexpr int kMkOutStride = 112;           // bytes per result row in the tile (96 + 16)

__device__ __forceinline__ void mk_wave_sync() {   // LDS traffic of this wave before / after: ordered and complete
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void mk_read_sample(const unsigned char *tile, int index, double x[3], double *delta_lambda, double k[4],
                                               double e1[4], double e2[4]) {
  double2 q[8];
#pragma unroll
  for (int r = 0; r < 8; r++) q[r] = *reinterpret_cast<const double2 *>(tile + (index * 8 + ((r + index) & 7)) * 16);
  x[0] = q[0].x; x[1] = q[0].y; x[2] = q[1].x; *delta_lambda = q[1].y;
  k[0] = q[2].x; k[1] = q[2].y; k[2] = q[3].x; k[3] = q[3].y;
  e1[0] = q[4].x; e1[1] = q[4].y; e1[2] = q[5].x; e1[3] = q[5].y;
  e2[0] = q[6].x; e2[1] = q[6].y; e2[2] = q[7].x; e2[3] = q[7].y;
}

__global__ void __launch_bounds__(256, 2) bl_transport_matrix_kernel(BlTransferArgs P) {
  using namespace fastpol;
  __shared__ __attribute__((aligned(16))) unsigned char tiles_even[4 * kMkTileBytes];
  __shared__ __attribute__((aligned(16))) unsigned char tiles_odd[4 * kMkTileBytes];
  const BlSpacetime st = P.st;
  const int lane_fixed = threadIdx.x & 63;
  const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  unsigned char *const tile_even = tiles_even + wave_in_block * kMkTileBytes;
  unsigned char *const tile_odd = tiles_odd + wave_in_block * kMkTileBytes;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int n_waves = (gridDim.x * blockDim.x) >> 6;
  int slot = __builtin_amdgcn_readfirstlane(wave), base = 0, num = 0;   // wave-uniform: scalar registers, scalar branches
  const int rays_done = bl_rays_done(P.counters, P.chunk_rays);
  auto settle = [&]() {   // first segment at or after (slot, base) that holds samples; false when the rays are used up
    for (;;) {
      if (slot >= rays_done) return false;
      if (base == 0) num = __builtin_amdgcn_readfirstlane(P.ray_sample_num[slot]);
      if (base < num) return true;
      slot += n_waves;
      base = 0;
    }
  };
  // slot q = lane + 64 j of the tile receives unit 8 i + ((r - i) & 7) of the segment, i = q / 8, r = q % 8 (clamped to the
  // ray's last record: always a valid address, never under a branch)
  auto request = [&](unsigned char *tile, int lane, int slot_r, int base_r, int num_r) {
    const double2 *samples = reinterpret_cast<const double2 *>(P.pol_samples + (size_t)P.ray_offset[slot_r]);
#pragma unroll
    for (int j = 0; j < 9; j++) {
      const int q = lane + 64 * j, i = q >> 3;
      const int unit = min(base_r * 8 + i * 8 + ((q - i) & 7), num_r * 8 - 1);
      // (as assembly: the compiler would otherwise wait for every LDS-DMA in flight before any LDS read it cannot prove
      // disjoint from the DMA's target, which is every read of the other tile; M0 carries the LDS address)
      const unsigned lds_address = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)(tile + j * 1024);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(samples + unit), "s"(lds_address) : "memory");
    }
  };
  if (!settle()) return;
  request(tile_even, lane_fixed, slot, base, num);
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the first tile has landed
  // one segment: arithmetic out of `tile` while the next segment travels into `tile_next`; returns false after the last one
  auto segment = [&](unsigned char *tile, unsigned char *tile_next) {
    const int slot_cur = slot, base_cur = base, num_cur = num;
    // the lane number as the loop sees it: opaque, so that the two dozen tile offsets derived from it are recomputed (three
    // integer instructions each) every segment instead of being hoisted out of the loop into registers that then spill
    int lane = lane_fixed;
    asm volatile("" : "+v"(lane));
    double2 *matrices = reinterpret_cast<double2 *>(P.pol_matrix + (size_t)P.ray_offset[slot_cur] * BL_POL_MATRIX_DOUBLES);
    bool live;
    {
      mk_wave_sync();
      base += 64;
      live = settle();
      // the next segment into the other tile (after the last segment: the current one again, into a tile nobody reads)
      request(tile_next, lane, live ? slot : slot_cur, live ? base : base_cur, live ? num : num_cur);
      const int rec = base_cur + lane;
      double m[10], dn = 0.0;
      if (rec < num_cur) {
        // previous sample in integration order of the transfer: record rec + 1 (the far end is its own predecessor, :150-154)
        const int prev = rec + 1 < num_cur ? lane + 1 : lane;
        double xn[3], xp[3], dp, kn[4], kp[4], e1n[4], e2n[4], e1p[4], e2p[4];
        mk_read_sample(tile, lane, xn, &dn, kn, e1n, e2n);
        mk_read_sample(tile, prev, xp, &dp, kp, e1p, e2p);
        const PointGeometry gn = point_geometry(st, xn[0], xn[1], xn[2]);
        const PointGeometry gp = point_geometry(st, xp[0], xp[1], xp[2]);
        const double kavg[4] = {0.5 * (kp[0] + kn[0]), 0.5 * (kp[1] + kn[1]), 0.5 * (kp[2] + kn[2]), 0.5 * (kp[3] + kn[3])};
        const Contracted cp_own = contract_momentum(gp, kp);     // second half step of the previous sample (:816-822)
        const Contracted cp_avg = contract_momentum(gp, kavg);   // first half step: connection and momentum averaged over the
        const Contracted cn_avg = contract_momentum(gn, kavg);   // two samples (:150-180)
        const double a = (dp + dn) * 0.25, b = (dp + dn) * 0.5;
        // c_a = g e'_a with g = eta + f l l (covariant l_0 = 1); u_a = G_2^T c_a
        double c[2][4], u[2][4];
#pragma unroll
        for (int t = 0; t < 2; t++) {
          const double *e = t == 0 ? e1n : e2n;
          const double le = gn.f * (e[0] + gn.l[0] * e[1] + gn.l[1] * e[2] + gn.l[2] * e[3]);
          c[t][0] = le - e[0];
          c[t][1] = e[1] + le * gn.l[0];
          c[t][2] = e[2] + le * gn.l[1];
          c[t][3] = e[3] + le * gn.l[2];
          double ya[4], yb[4];
          apply_transposed(gp, cp_avg, c[t], ya);
          apply_transposed(gn, cn_avg, c[t], yb);
#pragma unroll
          for (int be = 0; be < 4; be++) u[t][be] = 0.5 * (ya[be] + yb[be]);
        }
        const Projected pc[2] = {project(c[0], e1p, e2p, gp, cp_own), project(c[1], e1p, e2p, gp, cp_own)};
        const Projected pu[2] = {project(u[0], e1p, e2p, gp, cp_own), project(u[1], e1p, e2p, gp, cp_own)};
        double K[2][2][2][2];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < 2; j++) {
            double X1[2][2], X2[2][2];
            after_first_transport(pu[i], pc[j], a, X1);
            after_first_transport(pc[i], pu[j], a, X2);
#pragma unroll
            for (int cc = 0; cc < 2; cc++)
#pragma unroll
              for (int d = 0; d < 2; d++) K[i][j][cc][d] = pc[i].p[cc] * pc[j].p[d] - b * (X1[cc][d] + X2[cc][d]);
          }
        stokes_matrix(K, m);
      }
      mk_wave_sync();   // every lane has read its records: the tile takes the results
      if (rec < num_cur) {
        double2 *out = reinterpret_cast<double2 *>(tile + lane * kMkOutStride);
#pragma unroll
        for (int t = 0; t < 5; t++) out[t] = make_double2(m[2 * t], m[2 * t + 1]);
        out[5] = make_double2(dn, 0.0);   // the sample's length rides along: the sequential kernel reads one stream
      }
      mk_wave_sync();
      // vmcnt(0): the next tile has landed (requested a segment's arithmetic ago), and so have the stores of the segment
      // before; placed here, ahead of this segment's stores, so that those stay in flight during the next segment
      __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
      for (int j = 0; j < 6; j++) {
        const int unit = lane + 64 * j;
        const int sample = unit / 6;
        if (base_cur + sample < num_cur)
          matrices[(size_t)(base_cur + sample) * (BL_POL_MATRIX_DOUBLES / 2) + (unit - sample * 6)] = *reinterpret_cast<const double2 *>(tile + sample * kMkOutStride + (unit - sample * 6) * 16);
      }
    }
    return live;
  };
  for (;;) {
    if (!segment(tile_even, tile_odd)) break;
    if (!segment(tile_odd, tile_even)) break;
  }
}

// One ray per lane, frequencies in sequence: ss = M ss, the coupling, its guards; at the end the last half step and the
// camera's tetrad (:875-939) through the same projection algebra, and nu^3 (:942-949).
__global__ void __launch_bounds__(64, 2) bl_transfer_polarized_matrix_kernel(BlTransferArgs P) {
  using namespace fastpol;
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= bl_rays_done(P.counters, P.chunk_rays)) return;
  const BlSpacetime st = P.st;
  const int num = P.ray_sample_num[slot];
  const long long out_index = P.ray_out_index[slot];
  const double momentum_factor = P.ray_factor[slot];
  const size_t row = (size_t)P.n_rays_total;
  double *img = P.image + out_index;
  const BlPolSample *samples = P.pol_samples + (size_t)P.ray_offset[slot];
  const double *matrices = P.pol_matrix + (size_t)P.ray_offset[slot] * BL_POL_MATRIX_DOUBLES;
  const double2 *pc = P.pol_coeffs + (size_t)P.ray_offset[slot] * P.n_nu * 4;
  // the last sample's second half step and the camera projection do not depend on the frequency
  double m_cam[10];
  if (num > 0) {
    double x0[3], d0, k0[4], e1[4], e2[4];
    load_sample(samples, x0, &d0, k0, e1, e2);
    const PointGeometry g0 = point_geometry(st, x0[0], x0[1], x0[2]);
    const Contracted c0 = contract_momentum(g0, k0);
    const double *cp = P.camera_pos + 4 * out_index, *cd = P.camera_dir + 4 * out_index;
    const double kcov[4] = {cd[0], cd[1], cd[2], cd[3]};
    double gcov[4][4], gcon[4][4], kcon[4], up_con[4], tetrad[4][4];
    bl_pol::geodesic_metric(st, cp[1], cp[2], cp[3], gcov, gcon);
    for (int mu = 0; mu < 4; mu++) {
      double acc = 0.0;
      for (int nu = 0; nu < 4; nu++) acc += gcon[mu][nu] * kcov[nu];
      kcon[mu] = acc;
    }
    const double *u_con = P.cam_u_con, *u_cov = P.cam_u_cov, *vert = P.cam_vert_con_c;
    up_con[0] = u_con[0] * vert[0] - (u_cov[1] * vert[1] + u_cov[2] * vert[2] + u_cov[3] * vert[3]) / u_cov[0];
    up_con[1] = vert[1] + u_con[1] * vert[0];
    up_con[2] = vert[2] + u_con[2] * vert[0];
    up_con[3] = vert[3] + u_con[3] * vert[0];
    bl_pol::tetrad_frame(u_con, u_cov, kcon, kcov, up_con, gcov, gcon, tetrad);
    double c[2][4];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int mu = 0; mu < 4; mu++)
        c[t][mu] = gcov[mu][0] * tetrad[1 + t][0] + gcov[mu][1] * tetrad[1 + t][1] + gcov[mu][2] * tetrad[1 + t][2] + gcov[mu][3] * tetrad[1 + t][3];
    const Projected pcam[2] = {project(c[0], e1, e2, g0, c0), project(c[1], e1, e2, g0, c0)};
    double K[2][2][2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++) after_first_transport(pcam[i], pcam[j], (d0 + d0) * 0.25, K[i][j]);
    stokes_matrix(K, m_cam);
  }
  for (int l = 0; l < P.n_nu; l++) {
    const double freq = P.frequencies[l];
    if (num <= 0) {   // :94-96
      for (int a = 0; a < 4; a++) img[(size_t)(4 * l + a) * row] = 0.0;
      continue;
    }
    double ss_end[4] = {0.0, 0.0, 0.0, 0.0}, tau = 0.0;
    // the next sample's matrix and coefficients are requested before this sample's coupling (a lane walks its own ray: every
    // load is a cache line of its own, and two waves per SIMD do not hide that on their own); clamped at the ray's near end
    // so that the request is never under a branch
    double2 n0, n1, n2, n3, n4, n5, nc0, nc1, nc2, nc3;
    {
      const double2 *mq = reinterpret_cast<const double2 *>(matrices + (size_t)(num - 1) * BL_POL_MATRIX_DOUBLES);
      n0 = mq[0]; n1 = mq[1]; n2 = mq[2]; n3 = mq[3]; n4 = mq[4]; n5 = mq[5];
      const size_t at = (size_t)(num - 1) * P.n_nu + l;
      nc0 = pc[at * 4 + 0]; nc1 = pc[at * 4 + 1]; nc2 = pc[at * 4 + 2]; nc3 = pc[at * 4 + 3];
    }
    for (int rec = num - 1; rec >= 0; rec--) {
      const double2 m0 = n0, m1 = n1, m2 = n2, m3 = n3, m4 = n4;
      const double delta_lambda = n5.x;
      Coupling c;
      c.j_s[0] = nc0.x; c.j_s[1] = nc1.x; c.j_s[2] = 0.0; c.j_s[3] = nc1.y;
      c.alpha_s[0] = nc0.y; c.alpha_s[1] = nc2.x; c.alpha_s[2] = 0.0; c.alpha_s[3] = nc2.y;
      c.rho_s[0] = 0.0; c.rho_s[1] = nc3.x; c.rho_s[2] = 0.0; c.rho_s[3] = nc3.y;
      {
        const int next = rec > 0 ? rec - 1 : 0;
        const double2 *mq = reinterpret_cast<const double2 *>(matrices + (size_t)next * BL_POL_MATRIX_DOUBLES);
        n0 = mq[0]; n1 = mq[1]; n2 = mq[2]; n3 = mq[3]; n4 = mq[4]; n5 = mq[5];
        const size_t at = (size_t)next * P.n_nu + l;
        nc0 = pc[at * 4 + 0]; nc1 = pc[at * 4 + 1]; nc2 = pc[at * 4 + 2]; nc3 = pc[at * 4 + 3];
      }
      double ss_start[4];
      ss_start[0] = m0.x * ss_end[0] + m0.y * ss_end[1] + m1.x * ss_end[2];
      ss_start[1] = m1.y * ss_end[0] + m2.x * ss_end[1] + m2.y * ss_end[2];
      ss_start[2] = m3.x * ss_end[0] + m3.y * ss_end[1] + m4.x * ss_end[2];
      ss_start[3] = m4.y * ss_end[3];
      const double delta_lambda_cgs = delta_lambda * P.x_unit / (freq * momentum_factor);
      ss_end[0] = ss_end[1] = ss_end[2] = ss_end[3] = 0.0;
      couple_sample<CouplingMathTolerant>(&c, P.rotation_split != 0, delta_lambda_cgs, ss_start, ss_end);
      tau += c.delta_tau;   // unpolarized.cpp:139-140 (BlAuxImages::polarized_rows_only: written below)
    }
    const double nu_cu = freq * freq * freq;
    img[(size_t)(4 * l + 0) * row] = (m_cam[0] * ss_end[0] + m_cam[1] * ss_end[1] + m_cam[2] * ss_end[2]) * nu_cu;
    img[(size_t)(4 * l + 1) * row] = (m_cam[3] * ss_end[0] + m_cam[4] * ss_end[1] + m_cam[5] * ss_end[2]) * nu_cu;
    img[(size_t)(4 * l + 2) * row] = (m_cam[6] * ss_end[0] + m_cam[7] * ss_end[1] + m_cam[8] * ss_end[2]) * nu_cu;
    img[(size_t)(4 * l + 3) * row] = (m_cam[9] * ss_end[3]) * nu_cu;
    if (P.aux_images.polarized_rows_only && P.aux_images.image_tau) img[(size_t)(P.aux_images.offset_tau + l) * row] = tau;
  }
}

// (the two halves apart: the matrices need the samples' geometry only and may be built beside the per-frequency coefficient kernel)
extern "C" hipError_t bl_launch_transport_matrices(const BlTransferArgs *args, int num_cus, hipStream_t stream) {
  hipLaunchKernelGGL(bl_transport_matrix_kernel, dim3(num_cus * 2 * 4), dim3(256), 0, stream, *args);
  return hipGetLastError();
}
extern "C" hipError_t bl_launch_transfer_polarized_rays(const BlTransferArgs *args, hipStream_t stream) {
  const int grid = (args->chunk_rays + 63) / 64;
  hipLaunchKernelGGL(bl_transfer_polarized_matrix_kernel, dim3(grid), dim3(64), 0, stream, *args);
  return hipGetLastError();
}
extern "C" hipError_t bl_launch_transfer_polarized_matrix(const BlTransferArgs *args, int num_cus, hipStream_t stream) {
  const hipError_t err = bl_launch_transport_matrices(args, num_cus, stream);
  if (err != hipSuccess) return err;
  return bl_launch_transfer_polarized_rays(args, stream);
}

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

// Emission and absorption over a length dl without rotation (I A14-A17 and its limits; :391-451, :580-654)
__device__ void absorb(const Coupling &c, double dl, double dtau, const double ss_start[4], double ss_end[4]) {
  const double *j_s = c.j_s, *alpha_s = c.alpha_s;
  if (alpha_s[0] == 0.0) {
    for (int a = 0; a < 4; a++) ss_end[a] = ss_start[a] + j_s[a] * dl;
  } else if (c.alpha_p == 0.0) {
    if (c.optically_thin) {
      const double exp_neg = bl_exp(-dtau);
      const double expm1 = bl_expm1(dtau);
      for (int a = 0; a < 4; a++) ss_end[a] = exp_neg * (ss_start[a] + j_s[a] / alpha_s[0] * expm1);
    } else {
      for (int a = 0; a < 4; a++) ss_end[a] = j_s[a] / alpha_s[0];
    }
  } else if (c.optically_thin) {
    const double alpha_p = c.alpha_p, alpha_sq = c.alpha_sq;
    const double exp_neg_i = bl_exp(-dtau);
    const double exp_neg_p = bl_exp(-alpha_p * dl);
    const double sinh_p = bl_sinh(alpha_p * dl);
    const double cosh_p = bl_cosh(alpha_p * dl);
    const double coshm1_p = 0.5 * (bl_expm1(alpha_p * dl) + exp_neg_p - 1.0);
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
__device__ void rotate(const Coupling &c, const double ss_start[4], double ss_end[4]) {
  const double *rho_s = c.rho_s;
  const double cos_rho = bl_cos(c.rho_p * c.delta_lambda_cgs);
  const double sin_rho = bl_sin(c.rho_p * c.delta_lambda_cgs);
  double sin_sq_rho = bl_sin(c.rho_p * c.delta_lambda_cgs / 2.0);
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
    exp_v = bl_exp(-c.delta_tau);
    sin_v = bl_sin(lambda_2 * c.delta_lambda_cgs);
    cos_v = bl_cos(lambda_2 * c.delta_lambda_cgs);
    sinh_v = bl_sinh(lambda_1 * c.delta_lambda_cgs);
    cosh_v = bl_cosh(lambda_1 * c.delta_lambda_cgs);
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

}  // namespace

__global__ void __launch_bounds__(64) bl_transfer_polarized_kernel(BlTransferArgs P) {
  // The previous sample's connection, one column of 64 components per lane ([component][lane]: every access of a
  // wave touches 64 consecutive doubles): 32 KiB per wave, which leaves the registers to N, N_temp and the step
  __shared__ double connection_lds[64 * 64];
  double *connection_old = connection_lds + threadIdx.x;
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= P.chunk_rays) return;
  const BlSpacetime st = P.st;
  const int num = P.ray_sample_num[slot];
  const long long out_index = P.ray_out_index[slot];
  const double momentum_factor = P.ray_factor[slot];
  const size_t row = (size_t)P.n_rays_total;
  double *img = P.image + out_index;
  const BlPolSample *samples = P.pol_samples + (size_t)slot * P.ray_max_steps;
  const double2 *ja = P.transfer + (size_t)slot * P.ray_max_steps * P.n_nu;
  const double2 *pc = P.pol_coeffs + (size_t)slot * P.ray_max_steps * P.n_nu * 3;
  for (int l = 0; l < P.n_nu; l++) {
    const double freq = P.frequencies[l];
    if (num <= 0) {   // :94-96: nothing integrated; rows stay as the auxiliary kernel zeroed them
      for (int a = 0; a < 4; a++) img[(size_t)(4 * l + a) * row] = 0.0;
      continue;
    }
    Cplx nn_con[4][4], nn_con_temp[4][4];
    for (int mu = 0; mu < 4; mu++)
      for (int nu = 0; nu < 4; nu++) nn_con[mu][nu] = nn_con_temp[mu][nu] = Cplx{0.0, 0.0};
    // reference sample order is reversed integration order (geodesics.cpp:832-840): its n = 0 is record num - 1.
    // At n = 0 the "previous" connection and k^mu are the sample's own (:150-154, :167-169); averaging a value
    // with itself returns it, so the loop below needs no first-sample case.
    double delta_lambda_old = 0.0;
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
        const double2 c0 = ja[at], c1 = pc[at * 3 + 0], c2 = pc[at * 3 + 1], c3 = pc[at * 3 + 2];
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
      transport(gk_avg, (delta_lambda_old + delta_lambda) / 2.0, nn_con, nn_con_temp);
      for (int mu = 0; mu < 4; mu++)
        for (int nu = 0; nu < 4; nu++) nn_con[mu][nu] = nn_con_temp[mu][nu];

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

      c.delta_lambda_cgs = delta_lambda_cgs;
      c.delta_tau = c.alpha_s[0] * delta_lambda_cgs;
      c.optically_thin = c.delta_tau <= kDeltaTauMax;
      c.alpha_sq = c.alpha_s[1] * c.alpha_s[1] + c.alpha_s[3] * c.alpha_s[3];
      c.alpha_p = blm_sqrt(c.alpha_sq);
      c.rho_sq = c.rho_s[1] * c.rho_s[1] + c.rho_s[3] * c.rho_s[3];
      c.rho_p = blm_sqrt(c.rho_sq);
      if (P.rotation_split) {   // :388-568
        absorb(c, delta_lambda_cgs / 2.0, c.delta_tau / 2.0, ss_start, ss_end);
        ss_end[0] = (ss_end[0] < 0.0) ? 0.0 : ss_end[0];   // std::max(ss_end[0], 0.0)
        limit_polarization(ss_end);
        for (int a = 0; a < 4; a++) ss_start[a] = ss_end[a];
        if (c.rho_p != 0.0) rotate(c, ss_start, ss_end);
        limit_polarization(ss_end);
        for (int a = 0; a < 4; a++) ss_start[a] = ss_end[a];
        absorb(c, delta_lambda_cgs / 2.0, c.delta_tau / 2.0, ss_start, ss_end);
      } else if (c.alpha_s[0] == 0.0 && c.rho_p == 0.0) {
        for (int a = 0; a < 4; a++) ss_end[a] = ss_start[a] + c.j_s[a] * delta_lambda_cgs;
      } else if (c.alpha_p == 0.0 && c.rho_p == 0.0) {
        absorb(c, delta_lambda_cgs, c.delta_tau, ss_start, ss_end);
      } else if (c.alpha_s[0] == 0.0) {
        rotate(c, ss_start, ss_end);
        for (int a = 0; a < 4; a++) ss_end[a] += c.j_s[a] * delta_lambda_cgs;
      } else if (c.rho_p == 0.0) {
        absorb(c, delta_lambda_cgs, c.delta_tau, ss_start, ss_end);
      } else {
        couple_jointly(c, ss_start, ss_end);
      }
      // std::max(ss_end[0], 0.0) (:781): (a < b) ? b : a, so a NaN intensity stays NaN
      ss_end[0] = (ss_end[0] < 0.0) ? 0.0 : ss_end[0];
      limit_polarization(ss_end);

      // back to coordinates (:793-813), second half step (:816-833)
      from_stokes(s.e1, s.e2, ss_end, nn_con);
      for (int mu = 0; mu < 4; mu++)
        for (int nu = 0; nu < 4; nu++) nn_con_temp[mu][nu] = nn_con[mu][nu];
      transport(gk_new, (delta_lambda + delta_lambda_new) / 4.0, nn_con_temp, nn_con);

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
    }
  }
}

extern "C" hipError_t bl_launch_transfer_polarized(const BlTransferArgs *args, hipStream_t stream) {
  const int grid = (args->chunk_rays + 63) / 64;
  hipLaunchKernelGGL(bl_transfer_polarized_kernel, dim3(grid), dim3(64), 0, stream, *args);
  return hipGetLastError();
}

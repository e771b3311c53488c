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
// polarized coefficient pairs, and one BlPolSample per sample (position, renormalised k_mu, length, sampled
// velocity and field). The other image rows of a polarized run come from bl_transfer_aux_kernel.
//
// Arithmetic: plain IEEE double operations in the reference's order (no contraction: -ffp-contract=off), the
// pinned elementary functions of blmath.h; complex numbers as (re, im) pairs with the component-wise
// semantics libstdc++ gives real x complex and complex + complex. Not the benchmark path: the state of a ray
// (N, the previous connection: ~100 doubles) plus the temporaries of a step live in scratch.
#include <hip/hip_runtime.h>

#include "bl_device.h"

namespace {

constexpr double kPi = 3.141592653589793;
constexpr double kDeltaTauMax = 100.0;   // radiation_integrator.hpp:191

struct Cplx {
  double re, im;
};

__device__ __forceinline__ Cplx cscale(double a, Cplx z) { return Cplx{a * z.re, a * z.im}; }
__device__ __forceinline__ Cplx cadd(Cplx a, Cplx b) { return Cplx{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ Cplx csub(Cplx a, Cplx b) { return Cplx{a.re - b.re, a.im - b.im}; }

// radiation_geometry.cpp:138-262 through the shared Kerr-Schild scalars of bl_geometry.h
__device__ void geodesic_metric(const BlSpacetime &st, double x, double y, double z, double gcov[4][4], double gcon[4][4]) {
  if (st.ray_flat) {
    bl_minkowski(gcov);
    bl_minkowski(gcon);
    return;
  }
  BlKerrSchild ks;
  bl_kerr_schild(st, x, y, z, &ks);
  bl_gcov_ks(ks, gcov);
  bl_gcon_ks(ks, gcon);
}

// radiation_geometry.cpp:274-412
__device__ void geodesic_connection(const BlSpacetime &st, double x, double y, double z, double connection[4][4][4]) {
  if (st.ray_flat) {
    for (int mu = 0; mu < 4; mu++)
      for (int alpha = 0; alpha < 4; alpha++)
        for (int beta = 0; beta < 4; beta++) connection[mu][alpha][beta] = 0.0;
    return;
  }
  const double bh_a = st.bh_a, bh_m = st.bh_m;
  const double a2 = bh_a * bh_a;
  const double rr2 = x * x + y * y + z * z;
  const double r2 = 0.5 * (rr2 - a2 + bl_hypot(rr2 - a2, 2.0 * bh_a * z));
  const double r = blm_sqrt(r2);
  const double f = 2.0 * bh_m * r2 * r / (r2 * r2 + a2 * z * z);
  const double l[4] = {-1.0, (r * x + bh_a * y) / (r2 + a2), (r * y - bh_a * x) / (r2 + a2), z / r};
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
  for (int a = 1; a < 4; a++)
    for (int mu = 0; mu < 4; mu++)
      for (int nu = 0; nu < 4; nu++) {
        const double val = df[a] * l[mu] * l[nu] + f * dl[mu][a] * l[nu] + f * l[mu] * dl[nu][a];
        dgcov[a][mu][nu] = ((mu == 0) != (nu == 0)) ? -val : val;
      }
  for (int mu = 0; mu < 4; mu++)
    for (int alpha = 0; alpha < 4; alpha++)
      for (int beta = 0; beta < 4; beta++) {
        double acc = 0.0;
        for (int nu = 0; nu < 4; nu++)
          acc += 0.5 * gcon[mu][nu] * (dgcov[alpha][beta][nu] + dgcov[beta][alpha][nu] - dgcov[nu][alpha][beta]);
        connection[mu][alpha][beta] = acc;
      }
}

// radiation_geometry.cpp:421-573: metric of the simulation's coordinates at a CKS point
__device__ void simulation_metric(const BlSpacetime &st, int coord, double x, double y, double z, double gcov[4][4],
                                  double gcon[4][4]) {
  const double bh_a = st.bh_a, bh_m = st.bh_m;
  const double a2 = bh_a * bh_a;
  const double rr2 = x * x + y * y + z * z;
  const double r2 = 0.5 * (rr2 - a2 + bl_hypot(rr2 - a2, 2.0 * bh_a * z));
  const double r = blm_sqrt(r2);
  if (coord == BL_COORD_CKS) {
    const double f = 2.0 * bh_m * r2 * r / (r2 * r2 + a2 * z * z);
    const double l1 = (r * x + bh_a * y) / (r2 + a2), l2 = (r * y - bh_a * x) / (r2 + a2), l3 = z / r;
    const double lcov[4] = {1.0, l1, l2, l3}, lcon[4] = {-1.0, l1, l2, l3};
    for (int mu = 0; mu < 4; mu++)
      for (int nu = 0; nu < 4; nu++) {
        gcov[mu][nu] = f * lcov[mu] * lcov[nu];
        gcon[mu][nu] = -f * lcon[mu] * lcon[nu];
      }
    gcov[0][0] = f * lcov[0] * lcov[0] - 1.0;
    gcon[0][0] = -f * lcon[0] * lcon[0] - 1.0;
    for (int a = 1; a < 4; a++) {
      gcov[a][a] = f * lcov[a] * lcov[a] + 1.0;
      gcon[a][a] = -f * lcon[a] * lcon[a] + 1.0;
    }
    return;
  }
  const double cth = z / r;
  const double cth2 = cth * cth;
  const double sth2 = 1.0 - cth2;
  const double delta = r2 - 2.0 * bh_m * r + a2;
  const double sigma = r2 + a2 * cth2;
  for (int mu = 0; mu < 4; mu++)
    for (int nu = 0; nu < 4; nu++) gcov[mu][nu] = gcon[mu][nu] = 0.0;
  gcov[0][0] = -(1.0 - 2.0 * bh_m * r / sigma);
  gcov[0][1] = gcov[1][0] = 2.0 * bh_m * r / sigma;
  gcov[0][3] = gcov[3][0] = -2.0 * bh_m * bh_a * r * sth2 / sigma;
  gcov[1][1] = 1.0 + 2.0 * bh_m * r / sigma;
  gcov[1][3] = gcov[3][1] = -(1.0 + 2.0 * bh_m * r / sigma) * bh_a * sth2;
  gcov[2][2] = sigma;
  gcov[3][3] = (r2 + a2 + 2.0 * bh_m * a2 * r * sth2 / sigma) * sth2;
  gcon[0][0] = -(1.0 + 2.0 * bh_m * r / sigma);
  gcon[0][1] = gcon[1][0] = 2.0 * bh_m * r / sigma;
  gcon[1][1] = delta / sigma;
  gcon[1][3] = gcon[3][1] = bh_a / sigma;
  gcon[2][2] = 1.0 / sigma;
  gcon[3][3] = 1.0 / (sigma * sth2);
}

// radiation_geometry.cpp:69-126
__device__ void coordinate_jacobian(const BlSpacetime &st, int coord, double x, double y, double z, double jacobian[4][4]) {
  for (int mu = 0; mu < 4; mu++)
    for (int nu = 0; nu < 4; nu++) jacobian[mu][nu] = mu == nu ? 1.0 : 0.0;
  if (coord == BL_COORD_CKS) return;
  const double bh_a = st.bh_a;
  const double a2 = bh_a * bh_a;
  const double rr2 = x * x + y * y + z * z;
  const double r2 = 0.5 * (rr2 - a2 + bl_hypot(rr2 - a2, 2.0 * bh_a * z));
  const double r = blm_sqrt(r2);
  const double cth = z / r;
  const double sth = blm_sqrt(1.0 - cth * cth);
  const double ph = bl_atan2(y, x) - bl_atan(bh_a / r);
  const double sph = bl_sin(ph);
  const double cph = bl_cos(ph);
  jacobian[1][1] = sth * cph;
  jacobian[1][2] = cth * (r * cph - bh_a * sph);
  jacobian[1][3] = sth * (-r * sph - bh_a * cph);
  jacobian[2][1] = sth * sph;
  jacobian[2][2] = cth * (r * sph + bh_a * cph);
  jacobian[2][3] = sth * (r * cph - bh_a * sph);
  jacobian[3][1] = cth;
  jacobian[3][2] = -r * sth;
  jacobian[3][3] = 0.0;
}

// radiation_geometry.cpp:597-658
__device__ void tetrad_frame(const double ucon[4], const double ucov[4], const double kcon[4], const double kcov[4],
                             const double up_con[4], const double gcov[4][4], const double gcon[4][4], double tetrad[4][4]) {
  double omega = 0.0;
  for (int mu = 0; mu < 4; mu++) omega -= kcov[mu] * ucon[mu];
  double k_up_over_omega = 0.0;
  for (int mu = 0; mu < 4; mu++) k_up_over_omega += kcov[mu] * up_con[mu];
  k_up_over_omega /= omega;
  double u_up_over_omega = 0.0;
  for (int mu = 0; mu < 4; mu++) u_up_over_omega += ucov[mu] * up_con[mu];
  u_up_over_omega /= omega;
  for (int mu = 0; mu < 4; mu++) tetrad[0][mu] = ucon[mu];
  for (int mu = 0; mu < 4; mu++) tetrad[3][mu] = kcon[mu] / omega - ucon[mu];
  for (int mu = 0; mu < 4; mu++) tetrad[2][mu] = up_con[mu] - k_up_over_omega * tetrad[3][mu] + u_up_over_omega * kcon[mu];
  double norm = 0.0;
  for (int mu = 0; mu < 4; mu++)
    for (int nu = 0; nu < 4; nu++) norm += gcov[mu][nu] * tetrad[2][mu] * tetrad[2][nu];
  norm = blm_sqrt(norm);
  for (int mu = 0; mu < 4; mu++) tetrad[2][mu] /= norm;
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

// N^{mu nu} -> covariant components in a tetrad (:268-285, :904-925): N_{(a)(b)} = e_(a)^mu e_(b)^nu g g N
__device__ void to_tetrad(const double gcov[4][4], const double tetrad[4][4], const Cplx nn_con[4][4], Cplx nn_tet_cov[4][4]) {
  Cplx temp_b[4][4], temp_c[4][4], temp_d[4][4];
  for (int nu = 0; nu < 4; nu++)
    for (int alpha = 0; alpha < 4; alpha++) {
      Cplx acc = {0.0, 0.0};
      for (int beta = 0; beta < 4; beta++) acc = cadd(acc, cscale(gcov[nu][beta], nn_con[alpha][beta]));
      temp_b[nu][alpha] = acc;
    }
  for (int mu = 0; mu < 4; mu++)
    for (int nu = 0; nu < 4; nu++) {
      Cplx acc = {0.0, 0.0};
      for (int alpha = 0; alpha < 4; alpha++) acc = cadd(acc, cscale(gcov[mu][alpha], temp_b[nu][alpha]));
      temp_c[mu][nu] = acc;
    }
  for (int b = 0; b < 4; b++)
    for (int mu = 0; mu < 4; mu++) {
      Cplx acc = {0.0, 0.0};
      for (int nu = 0; nu < 4; nu++) acc = cadd(acc, cscale(tetrad[b][nu], temp_c[mu][nu]));
      temp_d[b][mu] = acc;
    }
  for (int a = 0; a < 4; a++)
    for (int b = 0; b < 4; b++) {
      Cplx acc = {0.0, 0.0};
      for (int mu = 0; mu < 4; mu++) acc = cadd(acc, cscale(tetrad[a][mu], temp_d[b][mu]));
      nn_tet_cov[a][b] = acc;
    }
}

__device__ void stokes_from(const Cplx nn_tet_cov[4][4], double ss[4]) {   // I 14
  ss[0] = 0.5 * cadd(nn_tet_cov[1][1], nn_tet_cov[2][2]).re;
  ss[1] = 0.5 * csub(nn_tet_cov[1][1], nn_tet_cov[2][2]).re;
  ss[2] = 0.5 * cadd(nn_tet_cov[1][2], nn_tet_cov[2][1]).re;
  ss[3] = 0.5 * csub(nn_tet_cov[2][1], nn_tet_cov[1][2]).im;
}

// Half a step of dN/dlambda = -(Gamma^mu_{alpha beta} k^alpha N^{beta nu} + (mu <-> nu)) (:181-198, :816-833):
// target += dN/dlambda(source) * dl
__device__ void transport(const double kcon[4], const double connection[4][4][4], const Cplx source[4][4], double dl,
                          Cplx target[4][4]) {
  double gk[4][4];
  for (int mu = 0; mu < 4; mu++)
    for (int beta = 0; beta < 4; beta++) {
      double acc = 0.0;
      for (int alpha = 0; alpha < 4; alpha++) acc += kcon[alpha] * connection[mu][alpha][beta];
      gk[mu][beta] = acc;
    }
  Cplx delta[4][4];
  for (int mu = 0; mu < 4; mu++)
    for (int nu = 0; nu < 4; nu++) {
      Cplx d = {0.0, 0.0};
      for (int beta = 0; beta < 4; beta++)
        d = csub(d, cadd(cscale(gk[mu][beta], source[beta][nu]), cscale(gk[nu][beta], source[mu][beta])));
      delta[mu][nu] = cscale(dl, d);   // complex * real: component-wise
    }
  for (int mu = 0; mu < 4; mu++)
    for (int nu = 0; nu < 4; nu++) target[mu][nu] = cadd(target[mu][nu], delta[mu][nu]);
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
    double delta_lambda_old = 0.0;
    double kcon_old[4] = {0.0, 0.0, 0.0, 0.0};
    double connection_old[4][4][4];
    Cplx nn_con[4][4], nn_con_temp[4][4];
    for (int mu = 0; mu < 4; mu++)
      for (int nu = 0; nu < 4; nu++) nn_con[mu][nu] = nn_con_temp[mu][nu] = Cplx{0.0, 0.0};
    // reference sample order is reversed integration order (geodesics.cpp:832-840): its n = 0 is record num - 1
    for (int rec = num - 1; rec >= 0; rec--) {
      const bool first = rec == num - 1;
      const BlPolSample s = samples[rec];
      const double delta_lambda = s.delta_lambda;
      const double delta_lambda_new = rec > 0 ? samples[rec - 1].delta_lambda : delta_lambda;
      const double delta_lambda_cgs = delta_lambda * P.x_unit / (freq * momentum_factor);
      const double x1 = s.x[0], x2 = s.x[1], x3 = s.x[2];
      const double kcov[4] = {s.k[0], s.k[1], s.k[2], s.k[3]};
      const double uu1 = s.uu[0], uu2 = s.uu[1], uu3 = s.uu[2], bb1 = s.bb[0], bb2 = s.bb[1], bb3 = s.bb[2];

      double gcov[4][4], gcon[4][4], connection[4][4][4];
      geodesic_metric(st, x1, x2, x3, gcov, gcon);
      geodesic_connection(st, x1, x2, x3, connection);
      for (int mu = 0; mu < 4; mu++)
        for (int alpha = 0; alpha < 4; alpha++)
          for (int beta = 0; beta < 4; beta++)
            connection_old[mu][alpha][beta] = first ? connection[mu][alpha][beta]
                : 0.5 * (connection_old[mu][alpha][beta] + connection[mu][alpha][beta]);
      double kcon[4];
      for (int mu = 0; mu < 4; mu++) {
        double acc = 0.0;
        for (int nu = 0; nu < 4; nu++) acc += gcon[mu][nu] * kcov[nu];
        kcon[mu] = acc;
      }
      for (int mu = 0; mu < 4; mu++) kcon_old[mu] = first ? kcon[mu] : 0.5 * (kcon_old[mu] + kcon[mu]);

      // first half step: N_temp += dN/dlambda(N) dl, N = N_temp
      transport(kcon_old, connection_old, nn_con, (delta_lambda_old + delta_lambda) / 2.0, nn_con_temp);
      for (int mu = 0; mu < 4; mu++)
        for (int nu = 0; nu < 4; nu++) nn_con[mu][nu] = nn_con_temp[mu][nu];

      // fluid frame (:201-265)
      double tetrad[4][4];
      {
        double gcov_sim[4][4], gcon_sim[4][4], jacobian[4][4];
        simulation_metric(st, P.simulation_coord, x1, x2, x3, gcov_sim, gcon_sim);
        const double uu0 = blm_sqrt(1.0 + gcov_sim[1][1] * uu1 * uu1 + 2.0 * gcov_sim[1][2] * uu1 * uu2
            + 2.0 * gcov_sim[1][3] * uu1 * uu3 + gcov_sim[2][2] * uu2 * uu2 + 2.0 * gcov_sim[2][3] * uu2 * uu3
            + gcov_sim[3][3] * uu3 * uu3);
        const double lapse = 1.0 / blm_sqrt(-gcon_sim[0][0]);
        const double shift1 = -gcon_sim[0][1] / gcon_sim[0][0];
        const double shift2 = -gcon_sim[0][2] / gcon_sim[0][0];
        const double shift3 = -gcon_sim[0][3] / gcon_sim[0][0];
        double ucon_sim[4], ucov_sim[4], bcon_sim[4];
        ucon_sim[0] = uu0 / lapse;
        ucon_sim[1] = uu1 - shift1 * uu0 / lapse;
        ucon_sim[2] = uu2 - shift2 * uu0 / lapse;
        ucon_sim[3] = uu3 - shift3 * uu0 / lapse;
        for (int mu = 0; mu < 4; mu++) {
          double acc = 0.0;
          for (int nu = 0; nu < 4; nu++) acc += gcov_sim[mu][nu] * ucon_sim[nu];
          ucov_sim[mu] = acc;
        }
        bcon_sim[0] = ucov_sim[1] * bb1 + ucov_sim[2] * bb2 + ucov_sim[3] * bb3;
        bcon_sim[1] = (bb1 + bcon_sim[0] * ucon_sim[1]) / ucon_sim[0];
        bcon_sim[2] = (bb2 + bcon_sim[0] * ucon_sim[2]) / ucon_sim[0];
        bcon_sim[3] = (bb3 + bcon_sim[0] * ucon_sim[3]) / ucon_sim[0];
        coordinate_jacobian(st, P.simulation_coord, x1, x2, x3, jacobian);
        double ucon[4], bcon[4], ucov[4], upcon[4];
        for (int mu = 0; mu < 4; mu++) {
          double au = 0.0, ab = 0.0;
          for (int nu = 0; nu < 4; nu++) {
            au += jacobian[mu][nu] * ucon_sim[nu];
            ab += jacobian[mu][nu] * bcon_sim[nu];
          }
          ucon[mu] = au;
          bcon[mu] = ab;
        }
        for (int mu = 0; mu < 4; mu++) {
          double acc = 0.0;
          for (int nu = 0; nu < 4; nu++) acc += gcov[mu][nu] * ucon[nu];
          ucov[mu] = acc;
        }
        const bool no_field = bb1 == 0.0 && bb2 == 0.0 && bb3 == 0.0;
        for (int mu = 0; mu < 4; mu++) upcon[mu] = no_field ? (mu == 3 ? 1.0 : 0.0) : bcon[mu];
        tetrad_frame(ucon, ucov, kcon, kcov, upcon, gcov, gcon, tetrad);
      }

      double ss_start[4], ss_end[4] = {0.0, 0.0, 0.0, 0.0};
      {
        Cplx nn_tet_cov[4][4];
        to_tetrad(gcov, tetrad, nn_con, nn_tet_cov);
        stokes_from(nn_tet_cov, ss_start);
      }

      Coupling c;
      {
        const size_t at = (size_t)rec * P.n_nu + l;
        const double2 c0 = ja[at], c1 = pc[at * 3 + 0], c2 = pc[at * 3 + 1], c3 = pc[at * 3 + 2];
        c.j_s[0] = c0.x; c.j_s[1] = c1.x; c.j_s[2] = 0.0; c.j_s[3] = c1.y;
        c.alpha_s[0] = c0.y; c.alpha_s[1] = c2.x; c.alpha_s[2] = 0.0; c.alpha_s[3] = c2.y;
        c.rho_s[0] = 0.0; c.rho_s[1] = c3.x; c.rho_s[2] = 0.0; c.rho_s[3] = c3.y;
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

      // back to coordinates (:793-813)
      {
        Cplx nn_tet_con[4][4];
        for (int mu = 0; mu < 4; mu++)
          for (int nu = 0; nu < 4; nu++) nn_tet_con[mu][nu] = Cplx{0.0, 0.0};
        nn_tet_con[1][1] = Cplx{ss_end[0] + ss_end[1], 0.0};
        nn_tet_con[2][2] = Cplx{ss_end[0] - ss_end[1], 0.0};
        // ss_2 -+ i ss_3 with libstdc++'s real -+ complex: i * ss_3 = (0 * ss_3, 1 * ss_3)
        nn_tet_con[1][2] = Cplx{-(0.0 * ss_end[3]) + ss_end[2], -(1.0 * ss_end[3])};
        nn_tet_con[2][1] = Cplx{0.0 * ss_end[3] + ss_end[2], 1.0 * ss_end[3]};
        Cplx temp_f[4][4];
        for (int nu = 0; nu < 4; nu++)
          for (int a = 0; a < 4; a++) {
            Cplx acc = {0.0, 0.0};
            for (int b = 0; b < 4; b++) acc = cadd(acc, cscale(tetrad[b][nu], nn_tet_con[a][b]));
            temp_f[nu][a] = acc;
          }
        for (int mu = 0; mu < 4; mu++)
          for (int nu = 0; nu < 4; nu++) {
            Cplx acc = {0.0, 0.0};
            for (int a = 0; a < 4; a++) acc = cadd(acc, cscale(tetrad[a][mu], temp_f[nu][a]));
            nn_con[mu][nu] = acc;
          }
      }

      // second half step: N_temp = N, N += dN/dlambda(N_temp) dl
      for (int mu = 0; mu < 4; mu++)
        for (int nu = 0; nu < 4; nu++) nn_con_temp[mu][nu] = nn_con[mu][nu];
      transport(kcon, connection, nn_con_temp, (delta_lambda + delta_lambda_new) / 4.0, nn_con);

      delta_lambda_old = delta_lambda;
      for (int mu = 0; mu < 4; mu++) kcon_old[mu] = kcon[mu];
      for (int mu = 0; mu < 4; mu++)
        for (int alpha = 0; alpha < 4; alpha++)
          for (int beta = 0; beta < 4; beta++) connection_old[mu][alpha][beta] = connection[mu][alpha][beta];
    }

    // camera frame (:875-939) and nu^3 (:942-949)
    {
      const double *cp = P.camera_pos + 4 * out_index, *cd = P.camera_dir + 4 * out_index;
      const double kcov[4] = {cd[0], cd[1], cd[2], cd[3]};
      double gcov[4][4], gcon[4][4], kcon[4], up_con[4], tetrad[4][4];
      geodesic_metric(st, cp[1], cp[2], cp[3], gcov, gcon);
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
      tetrad_frame(u_con, u_cov, kcon, kcov, up_con, gcov, gcon, tetrad);
      Cplx nn_tet_cov[4][4];
      to_tetrad(gcov, tetrad, nn_con, nn_tet_cov);
      double ss[4];
      stokes_from(nn_tet_cov, ss);
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

// bl_pol_frame.h - the fluid frame of one sample as the polarized transfer builds it (reference
// src/radiation_integrator/polarized.cpp:163-265), in the reference's plain arithmetic: geodesic metric, k^mu,
// simulation metric, u^mu and b^mu from the sampled primitives, Jacobian to Cartesian Kerr-Schild, tetrad.
//
// Used in two places: by the extended coefficient kernel for the samples that never reach its own tetrad (cut
// samples, cells without field: the polarized transfer still transports N through them), and by the polarized
// transfer kernel for the camera's frame at the end of a ray. For every other sample the coefficient kernel hands
// over the k^mu and tetrad it has already built for the coefficients (same function of the same inputs in the
// reference: simulation_coefficients.cpp:398-431 and polarized.cpp:201-265).
#ifndef BLACKLIGHT_AMD_BL_POL_FRAME_H_
#define BLACKLIGHT_AMD_BL_POL_FRAME_H_

#include "bl_device.h"

namespace bl_pol {

// radiation_geometry.cpp:138-262 through the shared Kerr-Schild scalars of bl_geometry.h
__device__ inline void geodesic_metric(const BlSpacetime &st, double x, double y, double z, double gcov[4][4], double gcon[4][4]) {
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

// radiation_geometry.cpp:421-573: metric of the simulation's coordinates at a CKS point
__device__ inline void simulation_metric(const BlSpacetime &st, int coord, double x, double y, double z, double gcov[4][4],
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
__device__ inline void coordinate_jacobian(const BlSpacetime &st, int coord, double x, double y, double z, double jacobian[4][4]) {
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
__device__ inline void tetrad_frame(const double ucon[4], const double ucov[4], const double kcon[4], const double kcov[4],
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

// polarized.cpp:163-265 for one sample: k^mu = g^{mu nu} k_nu and rows 1, 2 of the fluid tetrad (bl_polarized_frame_kernel)
static __device__ __forceinline__ void sample_frame(const BlSpacetime &st, int coord, double x1, double x2, double x3, const double kcov[4],
                                          const float uu[3], const float bb[3], BlPolSample *out) {
  double gcov[4][4], gcon[4][4];
  geodesic_metric(st, x1, x2, x3, gcov, gcon);
  double kcon[4];
  for (int mu = 0; mu < 4; mu++) {
    double acc = 0.0;
    for (int nu = 0; nu < 4; nu++) acc += gcon[mu][nu] * kcov[nu];
    kcon[mu] = acc;
  }
  const double uu1 = uu[0], uu2 = uu[1], uu3 = uu[2], bb1 = bb[0], bb2 = bb[1], bb3 = bb[2];
  double gcov_sim[4][4], gcon_sim[4][4], jacobian[4][4];
  simulation_metric(st, coord, x1, x2, x3, gcov_sim, gcon_sim);
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
  coordinate_jacobian(st, coord, x1, x2, x3, jacobian);
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
  double tetrad[4][4];
  tetrad_frame(ucon, ucov, kcon, kcov, upcon, gcov, gcon, tetrad);
  for (int mu = 0; mu < 4; mu++) {
    out->kcon[mu] = kcon[mu];
    out->e1[mu] = tetrad[1][mu];
    out->e2[mu] = tetrad[2][mu];
  }
}

}  // namespace bl_pol

#endif

// bl_oracle.cpp - CPU oracle: restatement of the reference's per-pixel geodesic + radiative
// transfer algorithm. TEST INFRASTRUCTURE ONLY (see bl_oracle.h). Not linked into, imported by or
// executed from the product path.
//
// Each function cites the reference file:line it follows. Expression order is kept exactly as in
// the reference wherever floating-point rounding depends on it; the file must be compiled with
// -ffp-contract=off (oracle/Makefile) so that no product-sum is fused.
//
// Difference in organisation (not in arithmetic): the reference materialises every stage for all
// pixels (geodesics -> sampling -> coefficients -> transfer); this restatement runs the same
// stages for one ray at a time, so memory is O(ray_max_steps) per thread.
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include <omp.h>

#include "bl_oracle.h"
#include "blmath.h"

namespace {

// ---------------------------------------------------------------------------------------------
// Math selection: glibc (tier A, -DBLO_LIBM) or the build's pinned blmath (tier B, default).
#ifdef BLO_LIBM
namespace M {
inline double hypot(double x, double y) { return std::hypot(x, y); }
inline double hypot3(double x, double y, double z) { return std::hypot(x, y, z); }
inline double pow(double x, double y) { return std::pow(x, y); }
inline double exp(double x) { return std::exp(x); }
inline double expm1(double x) { return std::expm1(x); }
inline double log(double x) { return std::log(x); }
inline double cbrt(double x) { return std::cbrt(x); }
inline double sin(double x) { return std::sin(x); }
inline double cos(double x) { return std::cos(x); }
inline double acos(double x) { return std::acos(x); }
inline double atan(double x) { return std::atan(x); }
inline double atan2(double y, double x) { return std::atan2(y, x); }
inline double sinh(double x) { return std::sinh(x); }
inline double cosh(double x) { return std::cosh(x); }
inline double tanh(double x) { return std::tanh(x); }
}  // namespace M
#else
namespace M {
inline double hypot(double x, double y) { return bl_hypot(x, y); }
inline double hypot3(double x, double y, double z) { return bl_hypot3(x, y, z); }
inline double pow(double x, double y) { return bl_pow(x, y); }
inline double exp(double x) { return bl_exp(x); }
inline double expm1(double x) { return bl_expm1(x); }
inline double log(double x) { return bl_log(x); }
inline double cbrt(double x) { return bl_cbrt(x); }
inline double sin(double x) { return bl_sin(x); }
inline double cos(double x) { return bl_cos(x); }
inline double acos(double x) { return bl_acos(x); }
inline double atan(double x) { return bl_atan(x); }
inline double atan2(double y, double x) { return bl_atan2(y, x); }
inline double sinh(double x) { return bl_sinh(x); }
inline double cosh(double x) { return bl_cosh(x); }
inline double tanh(double x) { return bl_tanh(x); }
}  // namespace M
#endif

// Constants, digit-for-digit from reference src/blacklight.hpp:10-27
namespace Math {
constexpr double pi = 3.141592653589793;
constexpr double sqrt2 = 1.4142135623730951;
}  // namespace Math
namespace Physics {
constexpr double c = 2.99792458e10;
constexpr double h = 6.62607015e-27;
constexpr double k_b = 1.380649e-16;
constexpr double m_p = 1.67262192369e-24;
constexpr double m_e = 9.1093837015e-28;
constexpr double e = 4.80320425e-10;
constexpr double gg_msun = 1.32712440018e26;
}  // namespace Physics
constexpr int num_cell_values = 7;  // blacklight.hpp:30-33: rho, n_e, p_gas, theta_e, bb, sigma, beta_inv

// std::pow(2.0, 11.0 / 12.0) of simulation_coefficients.cpp:480. g++ folds this call at compile
// time to the correctly rounded value (checked: the constant below is in the reference binary's
// .rodata and 11.0/12.0 is not), so it never reaches libm / the preloaded math library.
constexpr double pow_2_11_12 = 0x1.e3437e7101343p+0;

struct Oracle {
  const bl_params *p;
  const bl_grid_desc *g;
  // geometry (geodesic_integrator.cpp:107-123, radiation_integrator.cpp:420-431)
  double bh_m, bh_a, r_horizon, r_terminate, mass_msun;
  bool ray_flat;
  // camera (camera.cpp:27-415)
  double cam_x[4], u_con[4], u_cov[4], norm_con[4], norm_con_c[4], hor_con_c[4], vert_con_c[4];
  std::vector<double> image_frequencies;
  // image rows (radiation_integrator.cpp:436-520)
  int image_num_quantities;
  int image_offset_time, image_offset_length, image_offset_lambda, image_offset_emission,
      image_offset_tau, image_offset_lambda_ave, image_offset_emission_ave, image_offset_tau_int,
      image_offset_crossings;
  bool image_polarization;  // forced false in formula mode
  double plasma_thermal_frac;
  double power_jj = 0.0, power_aa = 0.0;   // simulation_coefficients.cpp:54-66
  double power_pol[7] = {};                // :67-80: jj_q, jj_v, aa_q, aa_v, rho, rho_q, rho_v
  // kappa-distribution electrons (simulation_coefficients.cpp:82-193), same names without the prefix
  struct {
    double jj_low, jj_high, jj_x_i, aa_low, aa_high, aa_x_i;
    double jj_low_q, jj_low_v, jj_high_q, jj_high_v, jj_x_q, jj_x_v;
    double aa_low_q, aa_low_v, aa_high_i, aa_high_q, aa_high_v, aa_x_q, aa_x_v;
    double rho_v, rho_frac;
    double rho_q_low_a, rho_q_low_b, rho_q_low_c, rho_q_low_d, rho_q_low_e;
    double rho_q_high_a, rho_q_high_b, rho_q_high_c, rho_q_high_d, rho_q_high_e;
    double rho_v_low_a, rho_v_low_b, rho_v_high_a, rho_v_high_b;
  } kappa = {};
  int render_num_images = 0;               // 0 in formula mode (radiation_integrator.cpp:136-145)
  // slow light: time slices held by the reader, latest first (simulation_reader.cpp:211-303)
  bool define_kappa_aa_high_i = false;   // blo_extra: unpolarized kappa-distribution electrons allowed
  int slow_n = 0;
  const bl_grid_desc *const *slow_grids = nullptr;
  const double *slow_times = nullptr;
  double snapshot_time = 0.0;
};

bool need(const bl_params *p, int index) { return p->has[index] != 0; }

// ---------------------------------------------------------------------------------------------
// geodesic_geometry.cpp:19-26
double RadialGeodesicCoordinate(const Oracle &o, double x, double y, double z) {
  double a2 = o.bh_a * o.bh_a;
  double rr2 = x * x + y * y + z * z;
  double r2 = 0.5 * (rr2 - a2 + M::hypot(rr2 - a2, 2.0 * o.bh_a * z));
  double r = std::sqrt(r2);
  return r;
}

void FlatMetric(double g[4][4]) {
  for (int mu = 0; mu < 4; mu++)
    for (int nu = 0; nu < 4; nu++) g[mu][nu] = 0.0;
  g[0][0] = -1.0;
  g[1][1] = 1.0;
  g[2][2] = 1.0;
  g[3][3] = 1.0;
}

// geodesic_geometry.cpp:38-93 (identical body in radiation_geometry.cpp:138-194)
void CovariantGeodesicMetric(const Oracle &o, double x, double y, double z, double gcov[4][4]) {
  if (o.ray_flat) return FlatMetric(gcov);
  double a2 = o.bh_a * o.bh_a;
  double rr2 = x * x + y * y + z * z;
  double r2 = 0.5 * (rr2 - a2 + M::hypot(rr2 - a2, 2.0 * o.bh_a * z));
  double r = std::sqrt(r2);
  double f = 2.0 * o.bh_m * r2 * r / (r2 * r2 + a2 * z * z);
  double l[4];
  l[0] = 1.0;
  l[1] = (r * x + o.bh_a * y) / (r2 + a2);
  l[2] = (r * y - o.bh_a * x) / (r2 + a2);
  l[3] = z / r;
  for (int mu = 0; mu < 4; mu++)
    for (int nu = 0; nu < 4; nu++) gcov[mu][nu] = f * l[mu] * l[nu];
  gcov[0][0] = f * l[0] * l[0] - 1.0;
  gcov[1][1] = f * l[1] * l[1] + 1.0;
  gcov[2][2] = f * l[2] * l[2] + 1.0;
  gcov[3][3] = f * l[3] * l[3] + 1.0;
}

// geodesic_geometry.cpp:105-161 (identical body in radiation_geometry.cpp:206-262)
void ContravariantGeodesicMetric(const Oracle &o, double x, double y, double z, double gcon[4][4]) {
  if (o.ray_flat) return FlatMetric(gcon);
  double a2 = o.bh_a * o.bh_a;
  double rr2 = x * x + y * y + z * z;
  double r2 = 0.5 * (rr2 - a2 + M::hypot(rr2 - a2, 2.0 * o.bh_a * z));
  double r = std::sqrt(r2);
  double f = 2.0 * o.bh_m * r2 * r / (r2 * r2 + a2 * z * z);
  double l[4];
  l[0] = -1.0;
  l[1] = (r * x + o.bh_a * y) / (r2 + a2);
  l[2] = (r * y - o.bh_a * x) / (r2 + a2);
  l[3] = z / r;
  for (int mu = 0; mu < 4; mu++)
    for (int nu = 0; nu < 4; nu++) gcon[mu][nu] = -f * l[mu] * l[nu];
  gcon[0][0] = -f * l[0] * l[0] - 1.0;
  gcon[1][1] = -f * l[1] * l[1] + 1.0;
  gcon[2][2] = -f * l[2] * l[2] + 1.0;
  gcon[3][3] = -f * l[3] * l[3] + 1.0;
}

// geodesic_geometry.cpp:173-276
void ContravariantGeodesicMetricDerivative(const Oracle &o, double x, double y, double z,
                                           double dgcon[3][4][4]) {
  if (o.ray_flat) {
    for (int a = 0; a < 3; a++)
      for (int mu = 0; mu < 4; mu++)
        for (int nu = 0; nu < 4; nu++) dgcon[a][mu][nu] = 0.0;
    return;
  }
  double bh_a = o.bh_a;
  double a2 = bh_a * bh_a;
  double rr2 = x * x + y * y + z * z;
  double r2 = 0.5 * (rr2 - a2 + M::hypot(rr2 - a2, 2.0 * bh_a * z));
  double r = std::sqrt(r2);
  double f = 2.0 * o.bh_m * r2 * r / (r2 * r2 + a2 * z * z);
  double l[4];
  l[0] = -1.0;
  l[1] = (r * x + bh_a * y) / (r2 + a2);
  l[2] = (r * y - bh_a * x) / (r2 + a2);
  l[3] = z / r;
  double dr[3], df[3], dl[4][3];
  dr[0] = r * x / (2.0 * r2 - rr2 + a2);
  dr[1] = r * y / (2.0 * r2 - rr2 + a2);
  dr[2] = (r * z + a2 * z / r) / (2.0 * r2 - rr2 + a2);
  df[0] = -(r2 * r2 - 3.0 * a2 * z * z) * dr[0] / (r * (r2 * r2 + a2 * z * z)) * f;
  df[1] = -(r2 * r2 - 3.0 * a2 * z * z) * dr[1] / (r * (r2 * r2 + a2 * z * z)) * f;
  df[2] = -((r2 * r2 - 3.0 * a2 * z * z) * dr[2] + 2.0 * a2 * r * z) / (r * (r2 * r2 + a2 * z * z)) * f;
  dl[0][0] = 0.0;
  dl[0][1] = 0.0;
  dl[0][2] = 0.0;
  dl[1][0] = ((x - 2.0 * r * l[1]) * dr[0] + r) / (r2 + a2);
  dl[1][1] = ((x - 2.0 * r * l[1]) * dr[1] + bh_a) / (r2 + a2);
  dl[1][2] = (x - 2.0 * r * l[1]) * dr[2] / (r2 + a2);
  dl[2][0] = ((y - 2.0 * r * l[2]) * dr[0] - bh_a) / (r2 + a2);
  dl[2][1] = ((y - 2.0 * r * l[2]) * dr[1] + r) / (r2 + a2);
  dl[2][2] = (y - 2.0 * r * l[2]) * dr[2] / (r2 + a2);
  dl[3][0] = -z / r2 * dr[0];
  dl[3][1] = -z / r2 * dr[1];
  dl[3][2] = -z / r2 * dr[2] + 1.0 / r;
  // :223-274, one pattern for all 48 components
  for (int a = 0; a < 3; a++)
    for (int mu = 0; mu < 4; mu++)
      for (int nu = 0; nu < 4; nu++)
        dgcon[a][mu][nu] = -(df[a] * l[mu] * l[nu] + f * dl[mu][a] * l[nu] + f * l[mu] * dl[nu][a]);
}

// geodesics.cpp:867-893
void GeodesicSubstepWithDistance(const Oracle &o, const double y[9], double k[9]) {
  double gcov[4][4], gcon[4][4], dgcon[3][4][4];
  CovariantGeodesicMetric(o, y[1], y[2], y[3], gcov);
  ContravariantGeodesicMetric(o, y[1], y[2], y[3], gcon);
  ContravariantGeodesicMetricDerivative(o, y[1], y[2], y[3], dgcon);
  for (int p = 0; p < 9; p++) k[p] = 0.0;
  for (int mu = 0; mu < 4; mu++)
    for (int nu = 0; nu < 4; nu++) k[mu] += gcon[mu][nu] * y[4 + nu];
  for (int a = 1; a < 4; a++)
    for (int mu = 0; mu < 4; mu++)
      for (int nu = 0; nu < 4; nu++) k[4 + a] -= 0.5 * dgcon[a - 1][mu][nu] * y[4 + mu] * y[4 + nu];
  double temp_a[4] = {};
  for (int a = 1; a < 4; a++)
    for (int mu = 0; mu < 4; mu++)
      temp_a[a] += (gcon[a][mu] - gcon[0][a] * gcon[0][mu] / gcon[0][0]) * y[4 + mu];
  for (int a = 1; a < 4; a++)
    for (int b = 1; b < 4; b++) k[8] += gcov[a][b] * temp_a[a] * temp_a[b];
  k[8] = -std::sqrt(k[8]);
}

// geodesics.cpp:909-925
void GeodesicSubstepWithoutDistance(const Oracle &o, const double y[8], double k[8]) {
  double gcon[4][4], dgcon[3][4][4];
  ContravariantGeodesicMetric(o, y[1], y[2], y[3], gcon);
  ContravariantGeodesicMetricDerivative(o, y[1], y[2], y[3], dgcon);
  for (int p = 0; p < 8; p++) k[p] = 0.0;
  for (int mu = 0; mu < 4; mu++)
    for (int nu = 0; nu < 4; nu++) k[mu] += gcon[mu][nu] * y[4 + nu];
  for (int a = 1; a < 4; a++)
    for (int mu = 0; mu < 4; mu++)
      for (int nu = 0; nu < 4; nu++) k[4 + a] -= 0.5 * dgcon[a - 1][mu][nu] * y[4 + mu] * y[4 + nu];
}

// Null-condition renormalisation of the spatial momentum (geodesics.cpp:296-309, :352-371)
void RenormalizeMomentum(const Oracle &o, double x, double y, double z, double k0, double k[3]) {
  double gcon[4][4];
  ContravariantGeodesicMetric(o, x, y, z, gcon);
  double kk[4] = {k0, k[0], k[1], k[2]};
  double temp_a = 0.0;
  for (int a = 1; a < 4; a++)
    for (int b = 1; b < 4; b++) temp_a += gcon[a][b] * kk[a] * kk[b];
  double temp_b = 0.0;
  for (int a = 1; a < 4; a++) temp_b += 2.0 * gcon[0][a] * kk[0] * kk[a];
  double temp_c = gcon[0][0] * kk[0] * kk[0];
  double temp_d = std::sqrt(temp_b * temp_b - 4.0 * temp_a * temp_c);
  double factor = temp_b < 0.0 ? (temp_d - temp_b) / (2.0 * temp_a) : -2.0 * temp_c / (temp_b + temp_d);
  for (int a = 0; a < 3; a++) k[a] *= factor;
}

// ---------------------------------------------------------------------------------------------
// camera.cpp:27-380 (serial part of InitializeCamera)
void InitializeCamera(Oracle &o) {
  const bl_params &p = *o.p;
  int nf = p.image_num_frequencies;
  o.image_frequencies.assign(nf, 0.0);
  if (nf == 1)
    o.image_frequencies[0] = p.image_frequency;
  else {
    o.image_frequencies[0] = p.image_frequency_start;
    o.image_frequencies[nf - 1] = p.image_frequency_end;
    for (int l = 1; l < nf - 1; l++) {
      double frac = static_cast<double>(l) / static_cast<double>(nf - 1);
      if (p.image_frequency_spacing == BL_SPACING_LIN_FREQ)
        o.image_frequencies[l] =
            p.image_frequency_start + frac * (p.image_frequency_end - p.image_frequency_start);
      else if (p.image_frequency_spacing == BL_SPACING_LIN_WAVE)
        o.image_frequencies[l] = 1.0 / (1.0 / p.image_frequency_start
            + frac * (1.0 / p.image_frequency_end - 1.0 / p.image_frequency_start));
      else if (p.image_frequency_spacing == BL_SPACING_LOG)
        o.image_frequencies[l] = M::exp(M::log(p.image_frequency_start)
            + frac * M::log(p.image_frequency_end / p.image_frequency_start));
    }
  }

  double bh_m = o.bh_m, bh_a = o.bh_a;
  double camera_r = p.camera_r;
  bool ray_flat = o.ray_flat;
  bool camera_pole = p.camera_pole != 0;
  double sth = M::sin(p.camera_th);
  double cth = M::cos(p.camera_th);
  double sph = M::sin(p.camera_ph);
  double cph = M::cos(p.camera_ph);
  double srot = M::sin(p.camera_rotation);
  double crot = M::cos(p.camera_rotation);
  double *cam_x = o.cam_x, *u_con = o.u_con, *u_cov = o.u_cov, *norm_con = o.norm_con;
  double *norm_con_c = o.norm_con_c, *hor_con_c = o.hor_con_c, *vert_con_c = o.vert_con_c;

  cam_x[0] = 0.0;
  cam_x[1] = sth * (camera_r * cph - bh_a * sph);
  cam_x[2] = sth * (camera_r * sph + bh_a * cph);
  cam_x[3] = camera_r * cth;
  if (ray_flat) {
    cam_x[1] = camera_r * sth * cph;
    cam_x[2] = camera_r * sth * sph;
  }
  double z_sign = cam_x[3] >= 0.0 ? 1.0 : -1.0;

  double a2 = bh_a * bh_a;
  double r2 = camera_r * camera_r;
  double delta = r2 - 2.0 * bh_m * camera_r + a2;
  double sigma = r2 + a2 * cth * cth;
  double g_cov_r_r = 1.0 + 2.0 * bh_m * camera_r / sigma;
  double g_cov_r_th = 0.0;
  double g_cov_r_ph = -(1.0 + 2.0 * bh_m * camera_r / sigma) * bh_a * sth * sth;
  double g_cov_th_th = sigma;
  double g_cov_th_ph = 0.0;
  double g_cov_ph_ph = (r2 + a2 + 2.0 * bh_m * a2 * camera_r / sigma * sth * sth) * sth * sth;
  double g_con_t_t = -(1.0 + 2.0 * bh_m * camera_r / sigma);
  double g_con_t_r = 2.0 * bh_m * camera_r / sigma;
  double g_con_t_th = 0.0;
  double g_con_t_ph = 0.0;
  double g_con_r_r = delta / sigma;
  double g_con_r_th = 0.0;
  double g_con_r_ph = bh_a / sigma;
  double g_con_th_th = 1.0 / sigma;
  double g_con_th_ph = 0.0;
  double g_con_ph_ph = 1.0 / (sigma * sth * sth);
  if (ray_flat and not camera_pole) {
    g_cov_r_r = 1.0; g_cov_r_th = 0.0; g_cov_r_ph = 0.0; g_cov_th_th = r2; g_cov_th_ph = 0.0;
    g_cov_ph_ph = r2 * sth * sth;
    g_con_t_t = -1.0; g_con_t_r = 0.0; g_con_t_th = 0.0; g_con_t_ph = 0.0; g_con_r_r = 1.0;
    g_con_r_th = 0.0; g_con_r_ph = 0.0; g_con_th_th = 1.0 / r2; g_con_th_ph = 0.0;
    g_con_ph_ph = 1.0 / (r2 * sth * sth);
  }
  if (camera_pole and not ray_flat) {
    double f = 2.0 * bh_m * camera_r / (r2 + a2);
    g_cov_r_r = 1.0 + f; g_cov_r_th = 0.0; g_cov_r_ph = 0.0; g_cov_th_th = 1.0; g_cov_th_ph = 0.0;
    g_cov_ph_ph = 1.0;
    g_con_t_t = -1.0 - f; g_con_t_r = z_sign * f; g_con_t_th = 0.0; g_con_t_ph = 0.0;
    g_con_r_r = 1.0 - f; g_con_r_th = 0.0; g_con_r_ph = 0.0; g_con_th_th = 1.0; g_con_th_ph = 0.0;
    g_con_ph_ph = 1.0;
  }
  if (ray_flat and camera_pole) {
    g_cov_r_r = 1.0; g_cov_r_th = 0.0; g_cov_r_ph = 0.0; g_cov_th_th = 1.0; g_cov_th_ph = 0.0;
    g_cov_ph_ph = 1.0;
    g_con_t_t = -1.0; g_con_t_r = 0.0; g_con_t_th = 0.0; g_con_t_ph = 0.0; g_con_r_r = 1.0;
    g_con_r_th = 0.0; g_con_r_ph = 0.0; g_con_th_th = 1.0; g_con_th_ph = 0.0; g_con_ph_ph = 1.0;
  }

  // camera velocity in spherical coordinates (:152-164)
  double camera_urn = p.camera_urn, camera_uthn = p.camera_uthn, camera_uphn = p.camera_uphn;
  double alpha = 1.0 / std::sqrt(-g_con_t_t);
  double beta_con_r = -g_con_t_r / g_con_t_t;
  double beta_con_th = -g_con_t_th / g_con_t_t;
  double beta_con_ph = -g_con_t_ph / g_con_t_t;
  double utn = std::sqrt(1.0 + g_cov_r_r * camera_urn * camera_urn
      + 2.0 * g_cov_r_th * camera_urn * camera_uthn + 2.0 * g_cov_r_ph * camera_urn * camera_uphn
      + g_cov_th_th * camera_uthn * camera_uthn + 2.0 * g_cov_th_ph * camera_uthn * camera_uphn
      + g_cov_ph_ph * camera_uphn * camera_uphn);
  u_con[0] = utn / alpha;
  double ur = camera_urn - beta_con_r / alpha * utn;
  double uth = camera_uthn - beta_con_th / alpha * utn;
  double uph = camera_uphn - beta_con_ph / alpha * utn;

  // Jacobian (:166-199)
  double dx_dr = sth * cph;
  double dy_dr = sth * sph;
  double dz_dr = cth;
  double dx_dth = cth * (camera_r * cph - bh_a * sph);
  double dy_dth = cth * (camera_r * sph + bh_a * cph);
  double dz_dth = -camera_r * sth;
  double dx_dph = sth * (-camera_r * sph - bh_a * cph);
  double dy_dph = sth * (camera_r * cph - bh_a * sph);
  double dz_dph = 0.0;
  if (ray_flat and not camera_pole) {
    dx_dr = sth * cph; dy_dr = sth * sph; dz_dr = cth;
    dx_dth = camera_r * cth * cph; dy_dth = camera_r * cth * sph; dz_dth = -camera_r * sth;
    dx_dph = -camera_r * sth * sph; dy_dph = camera_r * sth * cph; dz_dph = 0.0;
  }
  if (camera_pole) {
    dx_dr = 0.0; dy_dr = 0.0; dz_dr = z_sign;
    dx_dth = 1.0; dy_dth = 0.0; dz_dth = 0.0;
    dx_dph = 0.0; dy_dph = 1.0; dz_dph = 0.0;
  }

  // camera velocity (:201-212)
  u_con[1] = dx_dr * ur + dx_dth * uth + dx_dph * uph;
  u_con[2] = dy_dr * ur + dy_dth * uth + dy_dph * uph;
  u_con[3] = dz_dr * ur + dz_dth * uth + dz_dph * uph;
  double g_cov[4][4];
  CovariantGeodesicMetric(o, cam_x[1], cam_x[2], cam_x[3], g_cov);
  for (int mu = 0; mu < 4; mu++) {
    u_cov[mu] = 0.0;
    for (int nu = 0; nu < 4; nu++) u_cov[mu] += g_cov[mu][nu] * u_con[nu];
  }

  // photon momentum in spherical coordinates (:214-227)
  double g_con_rn_rn = (g_con_t_t * g_con_r_r - g_con_t_r * g_con_t_r) / g_con_t_t;
  double g_con_rn_thn = (g_con_t_t * g_con_r_th - g_con_t_r * g_con_t_th) / g_con_t_t;
  double g_con_rn_phn = (g_con_t_t * g_con_r_ph - g_con_t_r * g_con_t_ph) / g_con_t_t;
  double g_con_thn_thn = (g_con_t_t * g_con_th_th - g_con_t_th * g_con_t_th) / g_con_t_t;
  double g_con_thn_phn = (g_con_t_t * g_con_th_ph - g_con_t_th * g_con_t_ph) / g_con_t_t;
  double g_con_phn_phn = (g_con_t_t * g_con_ph_ph - g_con_t_ph * g_con_t_ph) / g_con_t_t;
  double k_rn = p.camera_k_r;
  double k_thn = p.camera_k_th;
  double k_phn = p.camera_k_ph;
  double k_tn = -std::sqrt(g_con_rn_rn * k_rn * k_rn + 2.0 * g_con_rn_thn * k_rn * k_thn
      + 2.0 * g_con_rn_phn * k_rn * k_phn + g_con_thn_thn * k_thn * k_thn
      + 2.0 * g_con_thn_phn * k_thn * k_phn + g_con_phn_phn * k_phn * k_phn);
  double k_t = alpha * k_tn + (beta_con_r * k_rn + beta_con_th * k_thn + beta_con_ph * k_phn);

  // Jacobian (:229-264)
  double rr2 = cam_x[1] * cam_x[1] + cam_x[2] * cam_x[2] + cam_x[3] * cam_x[3];
  double dr_dx = camera_r * cam_x[1] / (2.0 * r2 - rr2 + a2);
  double dr_dy = camera_r * cam_x[2] / (2.0 * r2 - rr2 + a2);
  double dr_dz = (camera_r * cam_x[3] + a2 * cam_x[3] / camera_r) / (2.0 * r2 - rr2 + a2);
  double dth_dx = cam_x[3] * dr_dx / (r2 * sth);
  double dth_dy = cam_x[3] * dr_dy / (r2 * sth);
  double dth_dz = (cam_x[3] * dr_dz - camera_r) / (r2 * sth);
  double dph_dx = -cam_x[2] / (cam_x[1] * cam_x[1] + cam_x[2] * cam_x[2]) + bh_a / (r2 + a2) * dr_dx;
  double dph_dy = cam_x[1] / (cam_x[1] * cam_x[1] + cam_x[2] * cam_x[2]) + bh_a / (r2 + a2) * dr_dy;
  double dph_dz = bh_a / (r2 + a2) * dr_dz;
  if (ray_flat and not camera_pole) {
    dr_dx = cam_x[1] / camera_r; dr_dy = cam_x[2] / camera_r; dr_dz = cam_x[3] / camera_r;
    dth_dx = cth * cph / camera_r; dth_dy = cth * sph / camera_r; dth_dz = -sth / camera_r;
    dph_dx = -sph / (camera_r * sth); dph_dy = cph / (camera_r * sth); dph_dz = 0.0;
  }
  if (camera_pole) {
    dr_dx = 0.0; dr_dy = 0.0; dr_dz = z_sign;
    dth_dx = 1.0; dth_dy = 0.0; dth_dz = 0.0;
    dph_dx = 0.0; dph_dy = 1.0; dph_dz = 0.0;
  }

  // photon momentum (:266-270)
  double k_x = dr_dx * p.camera_k_r + dth_dx * p.camera_k_th + dph_dx * p.camera_k_ph;
  double k_y = dr_dy * p.camera_k_r + dth_dy * p.camera_k_th + dph_dy * p.camera_k_ph;
  double k_z = dr_dz * p.camera_k_r + dth_dz * p.camera_k_th + dph_dz * p.camera_k_ph;
  double k_tc = u_con[0] * k_t + u_con[1] * k_x + u_con[2] * k_y + u_con[3] * k_z;

  // contravariant metric in camera frame (:272-280)
  double g_con[4][4];
  ContravariantGeodesicMetric(o, cam_x[1], cam_x[2], cam_x[3], g_con);
  double g_con_xc_xc = g_con[1][1] + u_con[1] * u_con[1];
  double g_con_xc_yc = g_con[1][2] + u_con[1] * u_con[2];
  double g_con_xc_zc = g_con[1][3] + u_con[1] * u_con[3];
  double g_con_yc_yc = g_con[2][2] + u_con[2] * u_con[2];
  double g_con_yc_zc = g_con[2][3] + u_con[2] * u_con[3];
  double g_con_zc_zc = g_con[3][3] + u_con[3] * u_con[3];

  // camera normal (:282-303)
  double norm_cov_xc = k_x - u_cov[1] / u_cov[0] * k_t;
  double norm_cov_yc = k_y - u_cov[2] / u_cov[0] * k_t;
  double norm_cov_zc = k_z - u_cov[3] / u_cov[0] * k_t;
  norm_con_c[0] = -k_tc;
  norm_con_c[1] = g_con_xc_xc * norm_cov_xc + g_con_xc_yc * norm_cov_yc + g_con_xc_zc * norm_cov_zc;
  norm_con_c[2] = g_con_xc_yc * norm_cov_xc + g_con_yc_yc * norm_cov_yc + g_con_yc_zc * norm_cov_zc;
  norm_con_c[3] = g_con_xc_zc * norm_cov_xc + g_con_yc_zc * norm_cov_yc + g_con_zc_zc * norm_cov_zc;
  double norm_norm = std::sqrt(norm_cov_xc * norm_con_c[1] + norm_cov_yc * norm_con_c[2]
      + norm_cov_zc * norm_con_c[3]);
  norm_cov_xc /= norm_norm;
  norm_cov_yc /= norm_norm;
  norm_cov_zc /= norm_norm;
  norm_con_c[0] /= norm_norm;
  norm_con_c[1] /= norm_norm;
  norm_con_c[2] /= norm_norm;
  norm_con_c[3] /= norm_norm;
  norm_con[0] = u_con[0] * norm_con_c[0]
      - (u_cov[1] * norm_con_c[1] + u_cov[2] * norm_con_c[2] + u_cov[3] * norm_con_c[3]) / u_cov[0];
  norm_con[1] = norm_con_c[1] + u_con[1] * norm_con_c[0];
  norm_con[2] = norm_con_c[2] + u_con[2] * norm_con_c[0];
  norm_con[3] = norm_con_c[3] + u_con[3] * norm_con_c[0];

  // up direction (:305-313)
  double up_con_xc = 0.0;
  double up_con_yc = 0.0;
  double up_con_zc = 1.0;
  if (camera_pole) {
    up_con_yc = 1.0;
    up_con_zc = 0.0;
  }

  // covariant metric in camera frame (:315-333)
  double g_cov_xc_xc = g_cov[1][1] - u_cov[1] / u_cov[0] * g_cov[1][0]
      - u_cov[1] / u_cov[0] * g_cov[1][0] + u_cov[1] * u_cov[1] / (u_cov[0] * u_cov[0]) * g_cov[0][0];
  double g_cov_xc_yc = g_cov[1][2] - u_cov[1] / u_cov[0] * g_cov[2][0]
      - u_cov[2] / u_cov[0] * g_cov[1][0] + u_cov[1] * u_cov[2] / (u_cov[0] * u_cov[0]) * g_cov[0][0];
  double g_cov_xc_zc = g_cov[1][3] - u_cov[1] / u_cov[0] * g_cov[3][0]
      - u_cov[3] / u_cov[0] * g_cov[1][0] + u_cov[1] * u_cov[3] / (u_cov[0] * u_cov[0]) * g_cov[0][0];
  double g_cov_yc_yc = g_cov[2][2] - u_cov[2] / u_cov[0] * g_cov[2][0]
      - u_cov[2] / u_cov[0] * g_cov[2][0] + u_cov[2] * u_cov[2] / (u_cov[0] * u_cov[0]) * g_cov[0][0];
  double g_cov_yc_zc = g_cov[2][3] - u_cov[2] / u_cov[0] * g_cov[3][0]
      - u_cov[3] / u_cov[0] * g_cov[2][0] + u_cov[2] * u_cov[3] / (u_cov[0] * u_cov[0]) * g_cov[0][0];
  double g_cov_zc_zc = g_cov[3][3] - u_cov[3] / u_cov[0] * g_cov[3][0]
      - u_cov[3] / u_cov[0] * g_cov[3][0] + u_cov[3] * u_cov[3] / (u_cov[0] * u_cov[0]) * g_cov[0][0];

  // vertical direction (:335-354)
  double up_norm = up_con_xc * norm_cov_xc + up_con_yc * norm_cov_yc + up_con_zc * norm_cov_zc;
  vert_con_c[0] = 0.0;
  vert_con_c[1] = up_con_xc - up_norm * norm_con_c[1];
  vert_con_c[2] = up_con_yc - up_norm * norm_con_c[2];
  vert_con_c[3] = up_con_zc - up_norm * norm_con_c[3];
  double vert_cov_xc = g_cov_xc_xc * vert_con_c[1] + g_cov_xc_yc * vert_con_c[2] + g_cov_xc_zc * vert_con_c[3];
  double vert_cov_yc = g_cov_xc_yc * vert_con_c[1] + g_cov_yc_yc * vert_con_c[2] + g_cov_yc_zc * vert_con_c[3];
  double vert_cov_zc = g_cov_xc_zc * vert_con_c[1] + g_cov_yc_zc * vert_con_c[2] + g_cov_zc_zc * vert_con_c[3];
  double vert_norm = std::sqrt(vert_cov_xc * vert_con_c[1] + vert_cov_yc * vert_con_c[2]
      + vert_cov_zc * vert_con_c[3]);
  vert_cov_xc /= vert_norm;
  vert_cov_yc /= vert_norm;
  vert_cov_zc /= vert_norm;
  vert_con_c[1] /= vert_norm;
  vert_con_c[2] /= vert_norm;
  vert_con_c[3] /= vert_norm;

  // determinant (:356-360)
  double det = g_cov_xc_xc * (g_cov_yc_yc * g_cov_zc_zc - g_cov_yc_zc * g_cov_yc_zc)
      + g_cov_xc_yc * (g_cov_yc_zc * g_cov_xc_zc - g_cov_xc_yc * g_cov_zc_zc)
      + g_cov_xc_zc * (g_cov_xc_yc * g_cov_yc_zc - g_cov_yc_yc * g_cov_xc_zc);
  double det_sqrt = std::sqrt(det);

  // horizontal direction (:362-366)
  hor_con_c[0] = 0.0;
  hor_con_c[1] = (vert_cov_yc * norm_cov_zc - vert_cov_zc * norm_cov_yc) / det_sqrt;
  hor_con_c[2] = (vert_cov_zc * norm_cov_xc - vert_cov_xc * norm_cov_zc) / det_sqrt;
  hor_con_c[3] = (vert_cov_xc * norm_cov_yc - vert_cov_yc * norm_cov_xc) / det_sqrt;

  // rotation (:368-380)
  double temp_hor_con_xc = hor_con_c[1];
  double temp_hor_con_yc = hor_con_c[2];
  double temp_hor_con_zc = hor_con_c[3];
  double temp_vert_con_xc = vert_con_c[1];
  double temp_vert_con_yc = vert_con_c[2];
  double temp_vert_con_zc = vert_con_c[3];
  hor_con_c[1] = temp_hor_con_xc * crot - temp_vert_con_xc * srot;
  hor_con_c[2] = temp_hor_con_yc * crot - temp_vert_con_yc * srot;
  hor_con_c[3] = temp_hor_con_zc * crot - temp_vert_con_zc * srot;
  vert_con_c[1] = temp_vert_con_xc * crot + temp_hor_con_xc * srot;
  vert_con_c[2] = temp_vert_con_yc * crot + temp_hor_con_yc * srot;
  vert_con_c[3] = temp_vert_con_zc * crot + temp_hor_con_zc * srot;
}

// Shared tail of SetPixelPlane / SetPixelPinhole (camera.cpp:553-583, :639-669)
void FinishPixel(const Oracle &o, const double position[4], double pcon[4], double direction[4],
                 double *factor) {
  double gcov[4][4];
  CovariantGeodesicMetric(o, position[1], position[2], position[3], gcov);
  double temp_a = gcov[0][0];
  double temp_b = 0.0;
  for (int a = 1; a < 4; a++) temp_b += 2.0 * gcov[0][a] * pcon[a];
  double temp_c = 0.0;
  for (int a = 1; a < 4; a++)
    for (int b = 1; b < 4; b++) temp_c += gcov[a][b] * pcon[a] * pcon[b];
  double temp_d = std::sqrt(std::max(temp_b * temp_b - 4.0 * temp_a * temp_c, 0.0));
  pcon[0] = temp_a == 0.0 ? -temp_c / (2.0 * temp_b)
      : (temp_b < 0.0 ? 2.0 * temp_c / (temp_d - temp_b) : -(temp_b + temp_d) / (2.0 * temp_a));
  for (int mu = 0; mu < 4; mu++) {
    direction[mu] = 0.0;
    for (int nu = 0; nu < 4; nu++) direction[mu] += gcov[mu][nu] * pcon[nu];
  }
  double nu_local = 0.0;
  if (o.p->image_normalization == BL_NORM_CAMERA)
    for (int mu = 0; mu < 4; mu++) nu_local -= direction[mu] * o.u_con[mu];
  else if (o.p->image_normalization == BL_NORM_INFINITY)
    nu_local = -direction[0];
  *factor = 1.0 / nu_local;
}

// camera.cpp:528-585
void SetPixelPlane(const Oracle &o, double u_ind, double v_ind, double position[4],
                   double direction[4], double *factor) {
  double u = u_ind * o.bh_m * o.p->camera_width;
  double v = v_ind * o.bh_m * o.p->camera_width;
  double dtc = u * o.hor_con_c[0] + v * o.vert_con_c[0];
  double dxc = u * o.hor_con_c[1] + v * o.vert_con_c[1];
  double dyc = u * o.hor_con_c[2] + v * o.vert_con_c[2];
  double dzc = u * o.hor_con_c[3] + v * o.vert_con_c[3];
  double dt = o.u_con[0] * dtc - (o.u_cov[1] * dxc + o.u_cov[2] * dyc + o.u_cov[3] * dzc) / o.u_cov[0];
  double dx = dxc + o.u_con[1] * dtc;
  double dy = dyc + o.u_con[2] * dtc;
  double dz = dzc + o.u_con[3] * dtc;
  position[0] = o.cam_x[0] + dt;
  position[1] = o.cam_x[1] + dx;
  position[2] = o.cam_x[2] + dy;
  position[3] = o.cam_x[3] + dz;
  double pcon[4];
  pcon[1] = o.norm_con[1];
  pcon[2] = o.norm_con[2];
  pcon[3] = o.norm_con[3];
  FinishPixel(o, position, pcon, direction, factor);
}

// camera.cpp:608-671
void SetPixelPinhole(const Oracle &o, double u_ind, double v_ind, double position[4],
                     double direction[4], double *factor) {
  position[0] = o.cam_x[0];
  position[1] = o.cam_x[1];
  position[2] = o.cam_x[2];
  position[3] = o.cam_x[3];
  double u = u_ind * o.bh_m * o.p->camera_width;
  double v = v_ind * o.bh_m * o.p->camera_width;
  double normalization = M::hypot3(u, v, o.p->camera_r);
  double frac_norm = o.p->camera_r / normalization;
  double frac_hor = -u / normalization;
  double frac_vert = -v / normalization;
  double dir_con_tc = o.norm_con_c[0];
  double dir_con_xc = frac_norm * o.norm_con_c[1] + frac_hor * o.hor_con_c[1] + frac_vert * o.vert_con_c[1];
  double dir_con_yc = frac_norm * o.norm_con_c[2] + frac_hor * o.hor_con_c[2] + frac_vert * o.vert_con_c[2];
  double dir_con_zc = frac_norm * o.norm_con_c[3] + frac_hor * o.hor_con_c[3] + frac_vert * o.vert_con_c[3];
  double pcon[4];
  pcon[1] = dir_con_xc + o.u_con[1] * dir_con_tc;
  pcon[2] = dir_con_yc + o.u_con[2] * dir_con_tc;
  pcon[3] = dir_con_zc + o.u_con[3] * dir_con_tc;
  FinishPixel(o, position, pcon, direction, factor);
}

// Pixel index within a level -> fractional image-plane coordinates
// (camera.cpp:393-396 for the root camera, :465-479 for refined blocks)
void PixelIndices(const Oracle &o, const bl_render_desc &d, int64_t pixel, double *u_ind, double *v_ind) {
  const bl_params &p = *o.p;
  if (d.level == 0) {
    int m2 = static_cast<int>(pixel / p.camera_resolution);
    int m1 = static_cast<int>(pixel % p.camera_resolution);
    *u_ind = (m1 - p.camera_resolution / 2.0 + 0.5) / p.camera_resolution;
    *v_ind = (m2 - p.camera_resolution / 2.0 + 0.5) / p.camera_resolution;
  } else {
    int bs = p.adaptive_block_size;
    int block_num_pix = bs * bs;
    int64_t block = pixel / block_num_pix;
    int m = static_cast<int>(pixel % block_num_pix);
    int effective_resolution = p.camera_resolution;
    for (int n = 1; n <= d.level; n++) effective_resolution *= 2;
    int block_v = d.block_locs[2 * block + 0];
    int block_u = d.block_locs[2 * block + 1];
    int m_offset = block_v * bs;
    int l_offset = block_u * bs;
    int m2 = m / bs;
    int m1 = m % bs;
    *u_ind = (m1 + l_offset - effective_resolution / 2.0 + 0.5) / effective_resolution;
    *v_ind = (m2 + m_offset - effective_resolution / 2.0 + 0.5) / effective_resolution;
  }
}

// ---------------------------------------------------------------------------------------------
// Per-ray scratch
struct RayBuffers {
  std::vector<double> geodesic_pos, geodesic_dir, geodesic_len;  // forward order, ray_max_steps
  std::vector<double> sample_pos, sample_dir, sample_len;        // reversed
  std::vector<double> j_i, alpha_i;                              // [n_nu][n]
  std::vector<double> cell_values;                               // [7][n]
  std::vector<unsigned char> sample_cut;
  // polarized transfer: j_Q, j_V, alpha_Q, alpha_V, rho_Q, rho_V [n_nu][max_steps] and the sampled
  // velocity / field of every sample (sample_uu1..bb3, zero where the sample was cut)
  std::vector<double> pol[6];
  std::vector<float> sample_ub;
  void EnablePolarization(int max_steps, int n_nu) {
    for (auto &v : pol) v.assign(static_cast<size_t>(n_nu) * max_steps, 0.0);
    sample_ub.assign(6 * static_cast<size_t>(max_steps), 0.0f);
  }
  explicit RayBuffers(int max_steps, int n_nu)
      : geodesic_pos(4 * static_cast<size_t>(max_steps)), geodesic_dir(4 * static_cast<size_t>(max_steps)),
        geodesic_len(max_steps), sample_pos(4 * static_cast<size_t>(max_steps)),
        sample_dir(4 * static_cast<size_t>(max_steps)), sample_len(max_steps),
        j_i(static_cast<size_t>(n_nu) * max_steps), alpha_i(static_cast<size_t>(n_nu) * max_steps),
        cell_values(static_cast<size_t>(num_cell_values) * max_steps), sample_cut(max_steps) {}
};

// geodesics.cpp:39-324 (per-ray body of IntegrateGeodesicsDP)
void IntegrateRayDP(const Oracle &o, const double camera_pos[4], const double camera_dir[4],
                    RayBuffers &b, int *p_sample_num, bool *p_flag) {
  // coefficients (:42-72)
  double a_vals[7][6] = {};
  a_vals[1][0] = 1.0 / 5.0;
  a_vals[2][0] = 3.0 / 40.0;
  a_vals[2][1] = 9.0 / 40.0;
  a_vals[3][0] = 44.0 / 45.0;
  a_vals[3][1] = -56.0 / 15.0;
  a_vals[3][2] = 32.0 / 9.0;
  a_vals[4][0] = 19372.0 / 6561.0;
  a_vals[4][1] = -25360.0 / 2187.0;
  a_vals[4][2] = 64448.0 / 6561.0;
  a_vals[4][3] = -212.0 / 729.0;
  a_vals[5][0] = 9017.0 / 3168.0;
  a_vals[5][1] = -355.0 / 33.0;
  a_vals[5][2] = 46732.0 / 5247.0;
  a_vals[5][3] = 49.0 / 176.0;
  a_vals[5][4] = -5103.0 / 18656.0;
  a_vals[6][0] = 35.0 / 384.0;
  a_vals[6][2] = 500.0 / 1113.0;
  a_vals[6][3] = 125.0 / 192.0;
  a_vals[6][4] = -2187.0 / 6784.0;
  a_vals[6][5] = 11.0 / 84.0;
  double b_vals_5[7] = {35.0 / 384.0, 0.0, 500.0 / 1113.0, 125.0 / 192.0, -2187.0 / 6784.0, 11.0 / 84.0, 0.0};
  double b_vals_4[7] = {5179.0 / 57600.0, 0.0, 7571.0 / 16695.0, 393.0 / 640.0, -92097.0 / 339200.0,
      187.0 / 2100.0, 1.0 / 40.0};
  double b_vals_4m[7] = {6025192743.0 / 30085553152.0, 0.0, 51252292925.0 / 65400821598.0,
      -2691868925.0 / 45128329728.0, 187940372067.0 / 1594534317056.0,
      -1776094331.0 / 19743644256.0, 11237099.0 / 235043384.0};
  double d_vals[7] = {-12715105075.0 / 11282082432.0, 0.0, 87487479700.0 / 32700410799.0,
      -10690763975.0 / 1880347072.0, 701980252875.0 / 199316789632.0, -1453857185.0 / 822651844.0,
      69997945.0 / 29380423.0};
  // numerical parameters (:75-78)
  double err_power = 0.2;
  double ray_err_factor = 0.9;
  double ray_min_factor = 0.2;
  double ray_max_factor = 10.0;

  const bl_params &p = *o.p;
  int ray_max_steps = p.ray_max_steps;
  double ray_step = p.ray_step;
  double y_vals[9], y_vals_temp[9], y_vals_5[9], y_vals_4[9], y_vals_4m[8], k_vals[7][9], r_vals[4][8];
  double gcon[4][4];
  bool flag = false;
  int sample_num = 0;
  double *gpos = b.geodesic_pos.data(), *gdir = b.geodesic_dir.data(), *glen = b.geodesic_len.data();

  for (int mu = 0; mu < 4; mu++) {
    y_vals[mu] = camera_pos[mu];
    y_vals[4 + mu] = camera_dir[mu];
  }
  y_vals[8] = 0.0;
  for (int q = 0; q < 9; q++) y_vals_5[q] = y_vals[q];
  double r_new = RadialGeodesicCoordinate(o, y_vals[1], y_vals[2], y_vals[3]);
  double h_new = -ray_step * r_new;
  int num_retry = 0;
  bool previous_fail = false;

  for (int n = 0; n < ray_max_steps;) {
    if (num_retry > p.ray_max_retries) {  // :139-143
      flag = true;
      break;
    }
    double h = h_new;
    if (not previous_fail and n > 0)  // :149-154
      for (int q = 0; q < 9; q++) {
        y_vals[q] = y_vals_5[q];
        k_vals[0][q] = k_vals[6][q];
      }
    if (not previous_fail and n == 0) GeodesicSubstepWithDistance(o, y_vals, k_vals[0]);
    double r = r_new;
    if (previous_fail) r = RadialGeodesicCoordinate(o, y_vals[1], y_vals[2], y_vals[3]);

    for (int substep = 1; substep < 7; substep++) {  // :162-170
      for (int q = 0; q < 9; q++) y_vals_temp[q] = y_vals[q];
      for (int s = 0; s < substep; s++)
        for (int q = 0; q < 9; q++) y_vals_temp[q] += a_vals[substep][s] * h * k_vals[s][q];
      GeodesicSubstepWithDistance(o, y_vals_temp, k_vals[substep]);
    }

    for (int q = 0; q < 9; q++) {  // :173-183
      y_vals_5[q] = y_vals[q];
      y_vals_4[q] = y_vals[q];
    }
    for (int s = 0; s < 7; s++)
      for (int q = 0; q < 9; q++) {
        y_vals_5[q] += b_vals_5[s] * h * k_vals[s][q];
        y_vals_4[q] += b_vals_4[s] * h * k_vals[s][q];
      }
    r_new = RadialGeodesicCoordinate(o, y_vals_5[1], y_vals_5[2], y_vals_5[3]);

    double error = 0.0;  // :187-194
    for (int q = 0; q < 8; q++) {
      double y_abs = std::max(std::abs(y_vals[q]), std::abs(y_vals_5[q]));
      double error_scale = p.ray_tol_abs + p.ray_tol_rel * y_abs;
      double delta_y = std::abs(y_vals_5[q] - y_vals_4[q]);
      error = std::max(error, delta_y / error_scale);
    }

    if (not (error <= 1.0)) {  // :197-209
      double h_factor = ray_min_factor;
      if (std::isfinite(error)) {
        double h_factor_ideal = ray_err_factor * M::pow(error, -err_power);
        h_factor = std::max(h_factor_ideal, ray_min_factor);
      }
      h_new = h * h_factor;
      num_retry += 1;
      previous_fail = true;
      continue;
    } else {  // :210-224
      double h_factor = ray_max_factor;
      if (error > 0.0) {
        h_factor = ray_err_factor * M::pow(error, -err_power);
        h_factor = std::max(h_factor, ray_min_factor);
        h_factor = std::min(h_factor, ray_max_factor);
      }
      if (previous_fail) h_factor = std::min(h_factor, 1.0);
      h_new = h * h_factor;
      num_retry = 0;
      previous_fail = false;
    }

    for (int q = 0; q < 8; q++) y_vals_4m[q] = y_vals[q];  // :227-231
    for (int s = 0; s < 7; s++)
      for (int q = 0; q < 8; q++) y_vals_4m[q] += b_vals_4m[s] * h * k_vals[s][q];

    double r_mid = RadialGeodesicCoordinate(o, y_vals_4m[1], y_vals_4m[2], y_vals_4m[3]);  // :234-245
    double delta_s_step = ray_step * r_mid;
    double delta_s_full = y_vals_5[8] - y_vals[8];
    int num_steps_ideal = static_cast<int>(std::ceil(delta_s_full / delta_s_step));
    int num_steps_max = ray_max_steps - n;
    int num_steps = num_steps_ideal;
    if (num_steps > num_steps_max) {
      num_steps = num_steps_max;
      flag = true;
    }

    if (num_steps_ideal == 1) {  // :248-259
      for (int mu = 0; mu < 4; mu++) {
        gpos[4 * n + mu] = y_vals_4m[mu];
        gdir[4 * n + mu] = y_vals_4m[4 + mu];
      }
      glen[n] = h;
    }
    if (num_steps_ideal > 1) {  // :262-274
      for (int q = 0; q < 8; q++) {
        r_vals[0][q] = y_vals_5[q] - y_vals[q];
        r_vals[1][q] = y_vals[q] - y_vals_5[q] + h * k_vals[0][q];
        r_vals[2][q] = 2.0 * (y_vals_5[q] - y_vals[q]) - h * (k_vals[0][q] + k_vals[6][q]);
        r_vals[3][q] = 0.0;
      }
      for (int s = 0; s < 7; s++)
        for (int q = 0; q < 8; q++) r_vals[3][q] += d_vals[s] * h * k_vals[s][q];
    }
    if (num_steps_ideal > 1)  // :277-293
      for (int nn = 0; nn < num_steps; nn++) {
        double frac = (nn + 0.5) / num_steps_ideal;
        for (int q = 0; q < 8; q++)
          y_vals_temp[q] = y_vals[q] + frac * (r_vals[0][q] + (1.0 - frac) * (r_vals[1][q]
              + frac * (r_vals[2][q] + (1.0 - frac) * r_vals[3][q])));
        for (int mu = 0; mu < 4; mu++) {
          gpos[4 * (n + nn) + mu] = y_vals_temp[mu];
          gdir[4 * (n + nn) + mu] = y_vals_temp[4 + mu];
        }
        glen[n + nn] = h / num_steps_ideal;
      }

    // renormalize momentum (:296-309)
    ContravariantGeodesicMetric(o, y_vals_5[1], y_vals_5[2], y_vals_5[3], gcon);
    double temp_a = 0.0;
    for (int a = 1; a < 4; a++)
      for (int bb = 1; bb < 4; bb++) temp_a += gcon[a][bb] * y_vals_5[4 + a] * y_vals_5[4 + bb];
    double temp_b = 0.0;
    for (int a = 1; a < 4; a++) temp_b += 2.0 * gcon[0][a] * y_vals_5[4] * y_vals_5[4 + a];
    double temp_c = gcon[0][0] * y_vals_5[4] * y_vals_5[4];
    double temp_d = std::sqrt(temp_b * temp_b - 4.0 * temp_a * temp_c);
    double factor = temp_b < 0.0 ? (temp_d - temp_b) / (2.0 * temp_a) : -2.0 * temp_c / (temp_b + temp_d);
    for (int a = 1; a < 4; a++) y_vals_5[4 + a] *= factor;

    // termination (:312-322)
    sample_num += num_steps;
    bool terminate_outer = r_new > p.camera_r and r_new > r;
    bool terminate_inner = r_new < o.r_terminate;
    if (terminate_outer or terminate_inner) break;
    bool last_step = n + num_steps >= ray_max_steps;
    if (last_step) flag = true;
    n += num_steps;
  }
  *p_sample_num = sample_num;
  *p_flag = flag;
}

// geodesics.cpp:418-534 (RK4) and :626-723 (RK2), per-ray bodies
void IntegrateRayRK(const Oracle &o, bool rk4, const double camera_pos[4], const double camera_dir[4],
                    RayBuffers &b, int *p_sample_num, bool *p_flag) {
  const bl_params &p = *o.p;
  int ray_max_steps = p.ray_max_steps;
  double gcon[4][4], y_vals[8], y_vals_substep[8], y_vals_accumulate[8], k_vals[8];
  double *gpos = b.geodesic_pos.data(), *gdir = b.geodesic_dir.data(), *glen = b.geodesic_len.data();
  bool flag = false;
  int sample_num = 0;
  for (int mu = 0; mu < 4; mu++) {
    y_vals[mu] = camera_pos[mu];
    y_vals[4 + mu] = camera_dir[mu];
  }
  double r_new = RadialGeodesicCoordinate(o, y_vals[1], y_vals[2], y_vals[3]);
  for (int n = 0; n < ray_max_steps; n++) {
    double r = r_new;
    double h = -p.ray_step * (r - o.r_horizon);
    if (rk4) {
      GeodesicSubstepWithoutDistance(o, y_vals, k_vals);
      for (int q = 0; q < 8; q++) y_vals_accumulate[q] = y_vals[q] + 1.0 / 6.0 * h * k_vals[q];
      for (int q = 0; q < 8; q++) y_vals_substep[q] = y_vals[q] + 0.5 * h * k_vals[q];
      GeodesicSubstepWithoutDistance(o, y_vals_substep, k_vals);
      for (int q = 0; q < 8; q++) y_vals_accumulate[q] += 1.0 / 3.0 * h * k_vals[q];
      for (int q = 0; q < 8; q++) y_vals_substep[q] = y_vals[q] + 0.5 * h * k_vals[q];
      GeodesicSubstepWithoutDistance(o, y_vals_substep, k_vals);
      for (int q = 0; q < 8; q++) y_vals_accumulate[q] += 1.0 / 3.0 * h * k_vals[q];
      for (int q = 0; q < 8; q++) y_vals_substep[q] = y_vals[q] + h * k_vals[q];
      GeodesicSubstepWithoutDistance(o, y_vals_substep, k_vals);
      for (int q = 0; q < 8; q++) y_vals_accumulate[q] += 1.0 / 6.0 * h * k_vals[q];
      for (int mu = 0; mu < 4; mu++) {
        gpos[4 * n + mu] = 0.5 * (y_vals[mu] + y_vals_accumulate[mu]);
        gdir[4 * n + mu] = 0.5 * (y_vals[4 + mu] + y_vals_accumulate[4 + mu]);
      }
      glen[n] = h;
      for (int q = 0; q < 8; q++) y_vals[q] = y_vals_accumulate[q];
    } else {
      GeodesicSubstepWithoutDistance(o, y_vals, k_vals);
      for (int q = 0; q < 8; q++) y_vals_substep[q] = y_vals[q] + h * k_vals[q];
      for (int q = 0; q < 8; q++) y_vals[q] += 1.0 / 2.0 * h * k_vals[q];
      for (int mu = 0; mu < 4; mu++) {
        gpos[4 * n + mu] = y_vals[mu];
        gdir[4 * n + mu] = y_vals[4 + mu];
      }
      glen[n] = h;
      GeodesicSubstepWithoutDistance(o, y_vals_substep, k_vals);
      for (int q = 0; q < 8; q++) y_vals[q] += 1.0 / 2.0 * h * k_vals[q];
    }
    ContravariantGeodesicMetric(o, y_vals[1], y_vals[2], y_vals[3], gcon);
    double temp_a = 0.0;
    for (int a = 1; a < 4; a++)
      for (int bb = 1; bb < 4; bb++) temp_a += gcon[a][bb] * y_vals[4 + a] * y_vals[4 + bb];
    double temp_b = 0.0;
    for (int a = 1; a < 4; a++) temp_b += 2.0 * gcon[0][a] * y_vals[4] * y_vals[4 + a];
    double temp_c = gcon[0][0] * y_vals[4] * y_vals[4];
    double temp_d = std::sqrt(temp_b * temp_b - 4.0 * temp_a * temp_c);
    double factor = temp_b < 0.0 ? (temp_d - temp_b) / (2.0 * temp_a) : -2.0 * temp_c / (temp_b + temp_d);
    for (int a = 1; a < 4; a++) y_vals[4 + a] *= factor;
    sample_num++;
    r_new = RadialGeodesicCoordinate(o, y_vals[1], y_vals[2], y_vals[3]);
    bool terminate_outer = r_new > p.camera_r and r_new > r;
    bool terminate_inner = r_new < o.r_terminate;
    if (terminate_outer or terminate_inner) break;
    bool last_step = n + 1 >= ray_max_steps;
    if (last_step) flag = true;
  }
  *p_sample_num = sample_num;
  *p_flag = flag;
}

// geodesics.cpp:327-371 (truncate, renormalise) and :808-849 (reverse), per ray.
// Returns the final sample_num.
int FinishRay(const Oracle &o, RayBuffers &b, int sample_num) {
  double *gpos = b.geodesic_pos.data(), *gdir = b.geodesic_dir.data(), *glen = b.geodesic_len.data();
  int num_samples = sample_num;
  if (num_samples > 1) {
    double r_new = RadialGeodesicCoordinate(o, gpos[1], gpos[2], gpos[3]);
    for (int n = 1; n < num_samples; n++) {
      double r_old = r_new;
      r_new = RadialGeodesicCoordinate(o, gpos[4 * n + 1], gpos[4 * n + 2], gpos[4 * n + 3]);
      bool terminate_outer = r_new > o.p->camera_r and r_new > r_old;
      bool terminate_inner = r_new < o.r_terminate;
      if (terminate_outer or terminate_inner) {
        sample_num = n;
        break;
      }
    }
  }
  for (int n = 0; n < sample_num; n++)
    RenormalizeMomentum(o, gpos[4 * n + 1], gpos[4 * n + 2], gpos[4 * n + 3], gdir[4 * n + 0], &gdir[4 * n + 1]);
  // reverse (:820-842); sample_len zero-initialised there, samples after a zero length are skipped
  std::fill(b.sample_len.begin(), b.sample_len.begin() + std::max(sample_num, 0), 0.0);
  for (int n = 0; n < sample_num; n++) {
    double len = glen[n];
    if (len == 0.0) break;
    int nr = sample_num - 1 - n;
    for (int mu = 0; mu < 4; mu++) {
      b.sample_pos[4 * nr + mu] = gpos[4 * n + mu];
      b.sample_dir[4 * nr + mu] = gdir[4 * n + mu];
    }
    b.sample_len[nr] = -len;
  }
  return sample_num;
}

// ---------------------------------------------------------------------------------------------
// radiation_geometry.cpp:37-57
void ConvertFromCKS(const Oracle &o, double *p_x1, double *p_x2, double *p_x3) {
  int coord = o.p->simulation_coord;
  if (coord == BL_COORD_SKS or coord == BL_COORD_FMKS) {
    double x = *p_x1, y = *p_x2, z = *p_x3;
    double a2 = o.bh_a * o.bh_a;
    double rr2 = x * x + y * y + z * z;
    double r2 = 0.5 * (rr2 - a2 + M::hypot(rr2 - a2, 2.0 * o.bh_a * z));
    double r = std::sqrt(r2);
    double th = M::acos(z / r);
    double ph = M::atan2(y, x) - M::atan(o.bh_a / r);
    ph += ph < 0.0 ? 2.0 * Math::pi : 0.0;
    ph -= ph >= 2.0 * Math::pi ? 2.0 * Math::pi : 0.0;
    *p_x1 = r;
    *p_x2 = th;
    *p_x3 = ph;
  }
}

// radiation_geometry.cpp:69-126
void CoordinateJacobian(const Oracle &o, double x, double y, double z, double jacobian[4][4]) {
  for (int mu = 0; mu < 4; mu++)
    for (int nu = 0; nu < 4; nu++) jacobian[mu][nu] = mu == nu ? 1.0 : 0.0;
  int coord = o.p->simulation_coord;
  if (coord == BL_COORD_SKS or coord == BL_COORD_FMKS) {
    double bh_a = o.bh_a;
    double a2 = bh_a * bh_a;
    double rr2 = x * x + y * y + z * z;
    double r2 = 0.5 * (rr2 - a2 + M::hypot(rr2 - a2, 2.0 * bh_a * z));
    double r = std::sqrt(r2);
    double cth = z / r;
    double sth = std::sqrt(1.0 - cth * cth);
    double ph = M::atan2(y, x) - M::atan(bh_a / r);
    double sph = M::sin(ph);
    double cph = M::cos(ph);
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
}

// radiation_geometry.cpp:421-491
void CovariantSimulationMetric(const Oracle &o, double x, double y, double z, double gcov[4][4]) {
  double bh_a = o.bh_a, bh_m = o.bh_m;
  if (o.p->simulation_coord == BL_COORD_CKS) {
    double a2 = bh_a * bh_a;
    double rr2 = x * x + y * y + z * z;
    double r2 = 0.5 * (rr2 - a2 + M::hypot(rr2 - a2, 2.0 * bh_a * z));
    double r = std::sqrt(r2);
    double f = 2.0 * bh_m * r2 * r / (r2 * r2 + a2 * z * z);
    double l[4] = {1.0, (r * x + bh_a * y) / (r2 + a2), (r * y - bh_a * x) / (r2 + a2), z / r};
    for (int mu = 0; mu < 4; mu++)
      for (int nu = 0; nu < 4; nu++) gcov[mu][nu] = f * l[mu] * l[nu];
    gcov[0][0] = f * l[0] * l[0] - 1.0;
    gcov[1][1] = f * l[1] * l[1] + 1.0;
    gcov[2][2] = f * l[2] * l[2] + 1.0;
    gcov[3][3] = f * l[3] * l[3] + 1.0;
  } else {
    double a2 = bh_a * bh_a;
    double rr2 = x * x + y * y + z * z;
    double r2 = 0.5 * (rr2 - a2 + M::hypot(rr2 - a2, 2.0 * bh_a * z));
    double r = std::sqrt(r2);
    double cth = z / r;
    double cth2 = cth * cth;
    double sth2 = 1.0 - cth2;
    double sigma = r2 + a2 * cth2;
    gcov[0][0] = -(1.0 - 2.0 * bh_m * r / sigma);
    gcov[0][1] = 2.0 * bh_m * r / sigma;
    gcov[0][2] = 0.0;
    gcov[0][3] = -2.0 * bh_m * bh_a * r * sth2 / sigma;
    gcov[1][0] = 2.0 * bh_m * r / sigma;
    gcov[1][1] = 1.0 + 2.0 * bh_m * r / sigma;
    gcov[1][2] = 0.0;
    gcov[1][3] = -(1.0 + 2.0 * bh_m * r / sigma) * bh_a * sth2;
    gcov[2][0] = 0.0;
    gcov[2][1] = 0.0;
    gcov[2][2] = sigma;
    gcov[2][3] = 0.0;
    gcov[3][0] = -2.0 * bh_m * bh_a * r * sth2 / sigma;
    gcov[3][1] = -(1.0 + 2.0 * bh_m * r / sigma) * bh_a * sth2;
    gcov[3][2] = 0.0;
    gcov[3][3] = (r2 + a2 + 2.0 * bh_m * a2 * r * sth2 / sigma) * sth2;
  }
}

// radiation_geometry.cpp:502-573
void ContravariantSimulationMetric(const Oracle &o, double x, double y, double z, double gcon[4][4]) {
  double bh_a = o.bh_a, bh_m = o.bh_m;
  if (o.p->simulation_coord == BL_COORD_CKS) {
    double a2 = bh_a * bh_a;
    double rr2 = x * x + y * y + z * z;
    double r2 = 0.5 * (rr2 - a2 + M::hypot(rr2 - a2, 2.0 * bh_a * z));
    double r = std::sqrt(r2);
    double f = 2.0 * bh_m * r2 * r / (r2 * r2 + a2 * z * z);
    double l[4] = {-1.0, (r * x + bh_a * y) / (r2 + a2), (r * y - bh_a * x) / (r2 + a2), z / r};
    for (int mu = 0; mu < 4; mu++)
      for (int nu = 0; nu < 4; nu++) gcon[mu][nu] = -f * l[mu] * l[nu];
    gcon[0][0] = -f * l[0] * l[0] - 1.0;
    gcon[1][1] = -f * l[1] * l[1] + 1.0;
    gcon[2][2] = -f * l[2] * l[2] + 1.0;
    gcon[3][3] = -f * l[3] * l[3] + 1.0;
  } else {
    double a2 = bh_a * bh_a;
    double rr2 = x * x + y * y + z * z;
    double r2 = 0.5 * (rr2 - a2 + M::hypot(rr2 - a2, 2.0 * bh_a * z));
    double r = std::sqrt(r2);
    double cth = z / r;
    double cth2 = cth * cth;
    double sth2 = 1.0 - cth2;
    double delta = r2 - 2.0 * bh_m * r + a2;
    double sigma = r2 + a2 * cth2;
    gcon[0][0] = -(1.0 + 2.0 * bh_m * r / sigma);
    gcon[0][1] = 2.0 * bh_m * r / sigma;
    gcon[0][2] = 0.0;
    gcon[0][3] = 0.0;
    gcon[1][0] = 2.0 * bh_m * r / sigma;
    gcon[1][1] = delta / sigma;
    gcon[1][2] = 0.0;
    gcon[1][3] = bh_a / sigma;
    gcon[2][0] = 0.0;
    gcon[2][1] = 0.0;
    gcon[2][2] = 1.0 / sigma;
    gcon[2][3] = 0.0;
    gcon[3][0] = 0.0;
    gcon[3][1] = bh_a / sigma;
    gcon[3][2] = 0.0;
    gcon[3][3] = 1.0 / (sigma * sth2);
  }
}

// radiation_geometry.cpp:597-658
void Tetrad(const double ucon[4], const double ucov[4], const double kcon[4], const double kcov[4],
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
  for (int mu = 0; mu < 4; mu++)
    tetrad[2][mu] = up_con[mu] - k_up_over_omega * tetrad[3][mu] + u_up_over_omega * kcon[mu];
  double norm = 0.0;
  for (int mu = 0; mu < 4; mu++)
    for (int nu = 0; nu < 4; nu++) norm += gcov[mu][nu] * tetrad[2][mu] * tetrad[2][nu];
  norm = std::sqrt(norm);
  for (int mu = 0; mu < 4; mu++) tetrad[2][mu] /= norm;
  double tetrad_1_cov[4];
  tetrad_1_cov[0] = tetrad[0][1] * (tetrad[2][3] * tetrad[3][2] - tetrad[2][2] * tetrad[3][3])
      + tetrad[0][2] * (tetrad[2][1] * tetrad[3][3] - tetrad[2][3] * tetrad[3][1])
      + tetrad[0][3] * (tetrad[2][2] * tetrad[3][1] - tetrad[2][1] * tetrad[3][2]);
  tetrad_1_cov[1] = tetrad[0][0] * (tetrad[2][2] * tetrad[3][3] - tetrad[2][3] * tetrad[3][2])
      + tetrad[0][2] * (tetrad[2][3] * tetrad[3][0] - tetrad[2][0] * tetrad[3][3])
      + tetrad[0][3] * (tetrad[2][0] * tetrad[3][2] - tetrad[2][2] * tetrad[3][0]);
  tetrad_1_cov[2] = tetrad[0][0] * (tetrad[2][3] * tetrad[3][1] - tetrad[2][1] * tetrad[3][3])
      + tetrad[0][1] * (tetrad[2][0] * tetrad[3][3] - tetrad[2][3] * tetrad[3][0])
      + tetrad[0][3] * (tetrad[2][1] * tetrad[3][0] - tetrad[2][0] * tetrad[3][1]);
  tetrad_1_cov[3] = tetrad[0][0] * (tetrad[2][1] * tetrad[3][2] - tetrad[2][2] * tetrad[3][1])
      + tetrad[0][1] * (tetrad[2][2] * tetrad[3][0] - tetrad[2][0] * tetrad[3][2])
      + tetrad[0][2] * (tetrad[2][0] * tetrad[3][1] - tetrad[2][1] * tetrad[3][0]);
  for (int mu = 0; mu < 4; mu++) {
    tetrad[1][mu] = 0.0;
    for (int nu = 0; nu < 4; nu++) tetrad[1][mu] += gcon[mu][nu] * tetrad_1_cov[nu];
  }
}

// Geometric cuts shared by simulation_sampling.cpp:237-292 and formula_coefficients.cpp:73-116.
// Returns true if the sample is cut.
bool GeometricCut(const Oracle &o, double x1, double x2, double x3, double r) {
  const bl_params &p = *o.p;
  if (r > p.camera_r) return true;
  if (p.cut_omit_near or p.cut_omit_far) {
    double dot_product = x1 * o.cam_x[1] + x2 * o.cam_x[2] + x3 * o.cam_x[3];
    if ((p.cut_omit_near and dot_product > 0.0) or (p.cut_omit_far and dot_product < 0.0)) return true;
  }
  if ((p.cut_omit_in >= 0.0 and r < p.cut_omit_in) or (p.cut_omit_out >= 0.0 and r > p.cut_omit_out))
    return true;
  if (p.cut_midplane_theta > 0.0 or p.cut_midplane_theta < 0.0) {
    double th = M::acos(x3 / r);
    if ((p.cut_midplane_theta > 0.0 and std::abs(th - Math::pi / 2.0) > p.cut_midplane_theta)
        or (p.cut_midplane_theta < 0.0 and std::abs(th - Math::pi / 2.0) < -p.cut_midplane_theta))
      return true;
  }
  if ((p.cut_midplane_z > 0.0 and std::abs(x3) > p.cut_midplane_z)
      or (p.cut_midplane_z < 0.0 and std::abs(x3) < -p.cut_midplane_z))
    return true;
  if (p.cut_plane) {
    double dot_product = (x1 - p.cut_plane_origin_x) * p.cut_plane_normal_x
        + (x2 - p.cut_plane_origin_y) * p.cut_plane_normal_y
        + (x3 - p.cut_plane_origin_z) * p.cut_plane_normal_z;
    if (dot_product < 0.0) return true;
  }
  return false;
}

inline float GridVal(const bl_grid_desc &g, int var, int b, int k, int j, int i) {
  // Array<float>(n_var, n_b, n_k, n_j, n_i) (utils/array.cpp:317-325)
  size_t idx = ((static_cast<size_t>(var) * g.n_blocks + b) * g.n_k + k) * g.n_j + j;
  return g.prim[idx * g.n_i + i];
}

// simulation_sampling.cpp:1334-1351
double InterpolateSimple(const bl_grid_desc &g, int var, int b, int k, int j, int i, double f_k, double f_j, double f_i) {
  double val_mmm = static_cast<double>(GridVal(g, var, b, k, j, i));
  double val_mmp = static_cast<double>(GridVal(g, var, b, k, j, i + 1));
  double val_mpm = static_cast<double>(GridVal(g, var, b, k, j + 1, i));
  double val_mpp = static_cast<double>(GridVal(g, var, b, k, j + 1, i + 1));
  double val_pmm = static_cast<double>(GridVal(g, var, b, k + 1, j, i));
  double val_pmp = static_cast<double>(GridVal(g, var, b, k + 1, j, i + 1));
  double val_ppm = static_cast<double>(GridVal(g, var, b, k + 1, j + 1, i));
  double val_ppp = static_cast<double>(GridVal(g, var, b, k + 1, j + 1, i + 1));
  double val = (1.0 - f_k) * (1.0 - f_j) * (1.0 - f_i) * val_mmm
      + (1.0 - f_k) * (1.0 - f_j) * f_i * val_mmp + (1.0 - f_k) * f_j * (1.0 - f_i) * val_mpm
      + (1.0 - f_k) * f_j * f_i * val_mpp + f_k * (1.0 - f_j) * (1.0 - f_i) * val_pmm
      + f_k * (1.0 - f_j) * f_i * val_pmp + f_k * f_j * (1.0 - f_i) * val_ppm
      + f_k * f_j * f_i * val_ppp;
  return val;
}

// simulation_sampling.cpp:84-93: blocks along x^3 at a refinement level
inline int N3Level(const bl_grid_desc &g, int level) { return (g.n_3_root / g.n_k) << level; }

// simulation_sampling.cpp:1068-1321: the cell (block, k, j, i) that stands for cell (k, j, i) of block b when that
// index is one beyond the block: the ghost cell's counterpart in the neighbouring block of the same level, the
// coarse cell containing it, the fine cell nearest the sample, or - where the mesh ends - the edge cell itself.
// Returns false where the reference throws "Grid interpolation failed."
bool FindNearbyInds(const Oracle &o, int b, int k, int j, int i, int k_c, int j_c, int i_c, double x3, double x2, double x1, int inds[4]) {
  const bl_grid_desc &g = *o.g;
  const bool sks = o.p->simulation_coord == BL_COORD_SKS;
  int n_b = g.n_blocks, n_i = g.n_i, n_j = g.n_j, n_k = g.n_k;
  auto levels = [&](int bb) { return g.levels[bb]; };
  auto locations = [&](int bb, int a) { return g.locations[3 * bb + a]; };
  int max_level = 0;
  for (int bb = 0; bb < n_b; bb++) max_level = std::max(max_level, levels(bb));
  int level = levels(b);
  int location_i = locations(b, 0), location_j = locations(b, 1), location_k = locations(b, 2);
  bool upper_i = i > n_i / 2, upper_j = j > n_j / 2, upper_k = k > n_k / 2;
  int i_safe = std::max(std::min(i, n_i - 1), 0);
  int j_safe = std::max(std::min(j, n_j - 1), 0);
  int k_safe = std::max(std::min(k, n_k - 1), 0);
  if (i == i_safe and j == j_safe and k == k_safe) {
    inds[0] = b; inds[1] = k; inds[2] = j; inds[3] = i;
    return true;
  }
  bool x1_off_grid = true, x2_off_grid = true, x3_off_grid = true;
  for (int b_alt = 0; b_alt < n_b; b_alt++) {
    int level_alt = levels(b_alt);
    int li = locations(b_alt, 0), lj = locations(b_alt, 1), lk = locations(b_alt, 2);
    if (x1_off_grid and i != i_safe) {
      bool same = level_alt == level and li == (i == -1 ? location_i - 1 : location_i + 1) and lj == location_j and lk == location_k;
      bool coarser = level_alt == level - 1 and li == (i == -1 ? (location_i - 1) / 2 : (location_i + 1) / 2)
          and lj == location_j / 2 and lk == location_k / 2;
      bool finer = level_alt == level + 1 and li == (i == -1 ? location_i * 2 - 1 : location_i * 2 + 2)
          and lj == (upper_j ? location_j * 2 + 1 : location_j * 2) and lk == (upper_k ? location_k * 2 + 1 : location_k * 2);
      if (same or coarser or finer) x1_off_grid = false;
    }
    if (x2_off_grid and j != j_safe) {
      bool same = level_alt == level and li == location_i and lj == (j == -1 ? location_j - 1 : location_j + 1) and lk == location_k;
      bool coarser = level_alt == level - 1 and li == location_i / 2
          and lj == (j == -1 ? (location_j - 1) / 2 : (location_j + 1) / 2) and lk == location_k / 2;
      bool finer = level_alt == level + 1 and li == (upper_i ? location_i * 2 + 1 : location_i * 2)
          and lj == (j == -1 ? location_j * 2 - 1 : location_j * 2 + 2) and lk == (upper_k ? location_k * 2 + 1 : location_k * 2);
      if (same or coarser or finer) x2_off_grid = false;
    }
    if (x3_off_grid and k != k_safe) {
      bool same = level_alt == level and li == location_i and lj == location_j and lk == (k == -1 ? location_k - 1 : location_k + 1);
      bool coarser = level_alt == level - 1 and li == location_i / 2 and lj == location_j / 2
          and lk == (k == -1 ? (location_k - 1) / 2 : (location_k + 1) / 2);
      bool finer = level_alt == level + 1 and li == (upper_i ? location_i * 2 + 1 : location_i * 2)
          and lj == (upper_j ? location_j * 2 + 1 : location_j * 2) and lk == (k == -1 ? location_k * 2 - 1 : location_k * 2 + 2);
      if (same or coarser or finer) x3_off_grid = false;
    }
    // across the periodic boundary in x^3 (:1181-1219)
    if (x3_off_grid and sks and k == -1 and location_k == 0) {
      bool same = level_alt == level and li == location_i and lj == location_j and lk == N3Level(g, level_alt) - 1;
      bool coarser = level_alt == level - 1 and li == location_i / 2 and lj == location_j / 2 and lk == N3Level(g, level_alt) - 1;
      bool finer = level_alt == level + 1 and li == (upper_i ? location_i * 2 + 1 : location_i * 2)
          and lj == (upper_j ? location_j * 2 + 1 : location_j * 2) and lk == N3Level(g, level_alt) - 1;
      if (same or coarser or finer) x3_off_grid = false;
    }
    if (x3_off_grid and sks and k == n_k and location_k == N3Level(g, level) - 1) {
      bool same = level_alt == level and li == location_i and lj == location_j and lk == 0;
      bool coarser = level_alt == level - 1 and li == location_i / 2 and lj == location_j / 2 and lk == 0;
      bool finer = level_alt == level + 1 and li == (upper_i ? location_i * 2 + 1 : location_i * 2)
          and lj == (upper_j ? location_j * 2 + 1 : location_j * 2) and lk == 0;
      if (same or coarser or finer) x3_off_grid = false;
    }
  }
  if (i == i_safe) x1_off_grid = false;
  if (j == j_safe) x2_off_grid = false;
  if (k == k_safe) x3_off_grid = false;
  if (x1_off_grid) i = i_safe;
  if (x2_off_grid) j = j_safe;
  if (x3_off_grid) k = k_safe;
  auto find = [&](int level_sought, int li, int lj, int lk) {
    for (int b_alt = 0; b_alt < n_b; b_alt++)
      if (levels(b_alt) == level_sought and locations(b_alt, 0) == li and locations(b_alt, 1) == lj and locations(b_alt, 2) == lk)
        return b_alt;
    return -1;
  };
  // same level (:1239-1261)
  int level_sought = level;
  int location_i_sought = i == i_safe ? location_i : i == -1 ? location_i - 1 : location_i + 1;
  int location_j_sought = j == j_safe ? location_j : j == -1 ? location_j - 1 : location_j + 1;
  int location_k_sought = k == k_safe ? location_k : k == -1 ? location_k - 1 : location_k + 1;
  if (sks and k == -1 and location_k == 0) location_k_sought = N3Level(g, level_sought) - 1;
  if (sks and k == n_k and location_k == N3Level(g, level) - 1) location_k_sought = 0;
  int i_sought = i == i_safe ? i : i == -1 ? n_i - 1 : 0;
  int j_sought = j == j_safe ? j : j == -1 ? n_j - 1 : 0;
  int k_sought = k == k_safe ? k : k == -1 ? n_k - 1 : 0;
  int b_alt = find(level_sought, location_i_sought, location_j_sought, location_k_sought);
  if (b_alt >= 0) {
    inds[0] = b_alt; inds[1] = k_sought; inds[2] = j_sought; inds[3] = i_sought;
    return true;
  }
  // coarser level (:1264-1291)
  level_sought = level - 1;
  if (level_sought >= 0) {
    location_i_sought = i == i_safe ? location_i / 2 : i == -1 ? (location_i - 1) / 2 : (location_i + 1) / 2;
    location_j_sought = j == j_safe ? location_j / 2 : j == -1 ? (location_j - 1) / 2 : (location_j + 1) / 2;
    location_k_sought = k == k_safe ? location_k / 2 : k == -1 ? (location_k - 1) / 2 : (location_k + 1) / 2;
    if (sks and k == -1 and location_k == 0) location_k_sought = N3Level(g, level_sought) - 1;
    if (sks and k == n_k and location_k == N3Level(g, level) - 1) location_k_sought = 0;
    i_sought = i == i_safe ? (location_i % 2 * n_i + i) / 2 : i == -1 ? n_i - 1 : 0;
    j_sought = j == j_safe ? (location_j % 2 * n_j + j) / 2 : j == -1 ? n_j - 1 : 0;
    k_sought = k == k_safe ? (location_k % 2 * n_k + k) / 2 : k == -1 ? n_k - 1 : 0;
    b_alt = find(level_sought, location_i_sought, location_j_sought, location_k_sought);
    if (b_alt >= 0) {
      inds[0] = b_alt; inds[1] = k_sought; inds[2] = j_sought; inds[3] = i_sought;
      return true;
    }
  }
  // finer level (:1294-1316)
  level_sought = level + 1;
  location_i_sought = location_i * 2 + (i == i_safe ? 0 : i == -1 ? -1 : 1) + (upper_i ? 1 : 0);
  location_j_sought = location_j * 2 + (j == j_safe ? 0 : j == -1 ? -1 : 1) + (upper_j ? 1 : 0);
  location_k_sought = location_k * 2 + (k == k_safe ? 0 : k == -1 ? -1 : 1) + (upper_k ? 1 : 0);
  if (sks and k == -1 and location_k == 0 and level_sought <= max_level) location_k_sought = N3Level(g, level_sought) - 1;
  if (sks and k == n_k and location_k == N3Level(g, level) - 1) location_k_sought = 0;
  i_sought = i == i_safe ? (upper_i ? (i - n_i / 2) * 2 : i * 2) : i == -1 ? n_i - 2 : 0;
  j_sought = j == j_safe ? (upper_j ? (j - n_j / 2) * 2 : j * 2) : j == -1 ? n_j - 2 : 0;
  k_sought = k == k_safe ? (upper_k ? (k - n_k / 2) * 2 : k * 2) : k == -1 ? n_k - 2 : 0;
  b_alt = find(level_sought, location_i_sought, location_j_sought, location_k_sought);
  if (b_alt >= 0) {
    const double *x1v = g.x1v + static_cast<size_t>(b) * n_i, *x2v = g.x2v + static_cast<size_t>(b) * n_j, *x3v = g.x3v + static_cast<size_t>(b) * n_k;
    inds[0] = b_alt; inds[1] = k_sought; inds[2] = j_sought; inds[3] = i_sought;
    inds[1] += k < k_c or (k == k_c and x3 > x3v[k_c]) ? 1 : 0;
    inds[2] += j < j_c or (j == j_c and x2 > x2v[j_c]) ? 1 : 0;
    inds[3] += i < i_c or (i == i_c and x1 > x1v[i_c]) ? 1 : 0;
    return true;
  }
  return false;
}

struct Prims {
  float rho, pgas, kappa, uu1, uu2, uu3, bb1, bb2, bb3;
};

// The block a thread is looking at (simulation_sampling.cpp:205-214): the reference keeps it from one
// sample to the next and only searches again, from block 0, when the point leaves it.
struct BlockState {
  int b = 0;
  bool valid = false;
  // slow light, per ray (simulation_sampling.cpp:218-226): which extrapolations its samples needed
  bool extrap[4] = {false, false, false, false};   // camera small, camera large, source small, source large
  double extrap_val[4] = {0.0, 0.0, 0.0, 0.0};
};

constexpr double extrapolation_tolerance = 1.0;   // simulation_reader.hpp:99

// simulation_sampling.cpp:201-575 + :666-1033 for one sample.
// Returns: 0 = sampled, 1 = cut, 2 = NaN, 3 = fallback values. *gathered set if the grid was read.
int SampleOne(const Oracle &o, const double pos[4], Prims *out, bool *gathered, BlockState *state) {
  const bl_params &p = *o.p;
  const bl_grid_desc &g = *o.g;
  *gathered = false;
  double x1 = pos[1], x2 = pos[2], x3 = pos[3];
  double r = RadialGeodesicCoordinate(o, x1, x2, x3);
  if (GeometricCut(o, x1, x2, x3, r)) return 1;
  ConvertFromCKS(o, &x1, &x2, &x3);
  // time interpolation (:296-349)
  const bool slow = p.slow_light_on != 0;
  const bool slow_interp = slow and p.slow_interp;
  int t_ind = 0;
  double t_frac = 0.0;
  if (slow) {
    const double *time = o.slow_times;
    const int chunk = o.slow_n;
    double x0 = pos[0] + o.snapshot_time;   // :231
    if (x0 >= time[0]) {
      if (x0 > time[0] + extrapolation_tolerance) {
        state->extrap[1] = true;
        state->extrap_val[1] = std::max(state->extrap_val[1], x0 - time[0]);
      } else if (x0 > time[0]) {
        state->extrap[0] = true;
        state->extrap_val[0] = std::max(state->extrap_val[0], x0 - time[0]);
      }
    } else if (x0 <= time[chunk - 1]) {
      if (x0 < time[chunk - 1] - extrapolation_tolerance) {
        state->extrap[3] = true;
        state->extrap_val[3] = std::max(state->extrap_val[3], time[chunk - 1] - x0);
      } else if (x0 < time[chunk - 1]) {
        state->extrap[2] = true;
        state->extrap_val[2] = std::max(state->extrap_val[2], time[chunk - 1] - x0);
      }
      if (slow_interp) {
        t_ind = chunk - 2;
        t_frac = 1.0;
      } else
        t_ind = chunk - 1;
    } else {
      while (time[t_ind++] > x0);
      t_ind--;
      if (slow_interp) {
        t_ind--;
        t_frac = (x0 - time[t_ind]) / (time[t_ind + 1] - time[t_ind]);
      } else if (time[t_ind - 1] - x0 <= x0 - time[t_ind])
        t_ind--;
    }
  }
  int n_i = g.n_i, n_j = g.n_j, n_k = g.n_k, n_b = g.n_blocks;
  if (p.simulation_coord == BL_COORD_FMKS) {
    // FMKS grids (:190-198, :352-394 with the grid's bounds in r, theta, phi; :396-456): one block in native coordinates,
    // position from the reader's SKS -> FMKS look-up table by plain scaling - no half-cell shift in x^1, x^2, the map's
    // x^2 read at row j + 1 in both terms, and no bounds on i_m + 1 / j_m + 1: exactly as written there. One thing is
    // not restated: the reference's per-thread "current block" bounds (:184-198, :362-392) start as these SKS bounds
    // and turn into the NATIVE coordinate box after the first off-grid sample that lies inside that box, so that its
    // image depends on the OpenMP schedule from then on. Here (and on the GPU) a sample outside the grid's SKS bounds
    // is off the grid, and nothing else changes.
    const double *bounds = g.simulation_bounds;
    if (not (x1 >= bounds[0] and x1 <= bounds[1] and x2 >= bounds[2] and x2 <= bounds[3] and x3 >= bounds[4] and x3 <= bounds[5])) {
      if (p.fallback_nan) return 2;
      return 3;
    }
    const size_t m1 = g.sks_map_n1, m2 = g.sks_map_n2;
    auto sks_map = [&](int v, long j, long i) { return g.sks_map[(static_cast<size_t>(v) * m2 + j) * m1 + i]; };
    double i_ind, j_ind;
    double f_i = std::modf((x1 - g.sks_map_r_in) / g.sks_map_dr, &i_ind);
    double f_j = std::modf(x2 / g.sks_map_dtheta, &j_ind);
    int i = static_cast<int>(i_ind), j = static_cast<int>(j_ind);
    double fmks_x1 = (1.0 - f_i) * sks_map(0, j, i) + f_i * sks_map(0, j, i + 1);
    double fmks_x2 = (1.0 - f_j) * sks_map(1, j + 1, i) + f_j * sks_map(1, j + 1, i);
    double x1_0 = g.x1f[0];
    double dx1 = g.x1f[1] - g.x1f[0];
    double dx2 = g.x2f[1] - g.x2f[0];
    f_i = std::modf((fmks_x1 - x1_0) / dx1, &i_ind);
    f_j = std::modf(fmks_x2 / dx2, &j_ind);
    int i_m = static_cast<int>(i_ind), j_m = static_cast<int>(j_ind);
    int k;
    for (k = 0; k < n_k; k++)
      if (g.x3f[k + 1] >= x3) break;
    int k_m = k == 0 or (k != n_k - 1 and x3 >= g.x3v[k]) ? k : k - 1;
    double f_k = (x3 - g.x3v[k_m]) / (g.x3v[k_m + 1] - g.x3v[k_m]);
    bool code_kappa = p.plasma_model == BL_PLASMA_CODE_KAPPA;
    *gathered = true;
    const int vars[9] = {g.ind_rho, g.ind_pgas, g.ind_kappa, g.ind_uu1, g.ind_uu2, g.ind_uu3, g.ind_bb1, g.ind_bb2, g.ind_bb3};
    float *dst[9] = {&out->rho, &out->pgas, &out->kappa, &out->uu1, &out->uu2, &out->uu3, &out->bb1, &out->bb2, &out->bb3};
    // The reference's Array has no bounds: a cell index beyond a row is the next row's cell, and a cell beyond the block
    // is the next VARIABLE's data (or, for the last variable, memory past the array). The first is reproduced as it is;
    // the second has no value to reproduce (status 4, like the upper edges of the last MeshBlock above).
    const long n_cells = static_cast<long>(n_k) * n_j * n_i;
    auto beyond = [&](long kk, long jj, long ii) { return (kk * n_j + jj) * n_i + ii >= n_cells or (kk * n_j + jj) * n_i + ii < 0; };
    // the values: from the time slice(s) of the sample as everywhere else (:710-786, :809-912 - the FMKS branch only finds
    // indices and fractions, :396-456)
    auto fmks_slice = [&](int t) -> const bl_grid_desc & { return slow ? *o.slow_grids[t] : g; };
    if (not p.simulation_interp) {
      int jn = f_j >= 0.5 ? j_m + 1 : j_m, in = f_i >= 0.5 ? i_m + 1 : i_m;
      if (beyond(k, jn, in)) return 4;
      for (int v = 0; v < 9; v++) {
        if (v == 2 and not code_kappa) {
          *dst[v] = 0.0f;
          continue;
        }
        if (not slow_interp)
          *dst[v] = GridVal(fmks_slice(t_ind), vars[v], 0, k, jn, in);
        else {
          double val_1 = static_cast<double>(GridVal(fmks_slice(t_ind), vars[v], 0, k, jn, in));
          double val_2 = static_cast<double>(GridVal(fmks_slice(t_ind + 1), vars[v], 0, k, jn, in));
          *dst[v] = static_cast<float>((1.0 - t_frac) * val_1 + t_frac * val_2);
        }
      }
      return 0;
    }
    if (beyond(k_m + 1, j_m + 1, i_m + 1) or beyond(k_m, j_m, i_m)) return 4;
    auto fmks_spatial = [&](const bl_grid_desc &gs, int v) {
      double val = InterpolateSimple(gs, vars[v], 0, k_m, j_m, i_m, f_k, f_j, f_i);
      if (v < 3 and val <= 0.0) val = static_cast<double>(GridVal(gs, vars[v], 0, k_m, j_m, i_m));
      return val;
    };
    for (int v = 0; v < 9; v++) {
      if (v == 2 and not code_kappa) {
        *dst[v] = 0.0f;
        continue;
      }
      if (not slow_interp)
        *dst[v] = static_cast<float>(fmks_spatial(fmks_slice(t_ind), v));
      else {
        double val_1 = fmks_spatial(fmks_slice(t_ind), v);
        double val_2 = fmks_spatial(fmks_slice(t_ind + 1), v);
        *dst[v] = static_cast<float>((1.0 - t_frac) * val_1 + t_frac * val_2);
      }
    }
    return 0;
  }
  // block test and search (:352-394)
  if (!state->valid) {   // :205-214: block 0 to start with
    state->b = 0;
    state->valid = true;
  }
  int b = state->b;
  auto inside = [&](int bb) {
    return x1 >= g.x1f[static_cast<size_t>(bb) * (n_i + 1)] and x1 <= g.x1f[static_cast<size_t>(bb) * (n_i + 1) + n_i]
        and x2 >= g.x2f[static_cast<size_t>(bb) * (n_j + 1)] and x2 <= g.x2f[static_cast<size_t>(bb) * (n_j + 1) + n_j]
        and x3 >= g.x3f[static_cast<size_t>(bb) * (n_k + 1)] and x3 <= g.x3f[static_cast<size_t>(bb) * (n_k + 1) + n_k];
  };
  if (!inside(b)) {
    int b_new;
    for (b_new = 0; b_new < n_b; b_new++)
      if (inside(b_new)) break;
    if (b_new == n_b) {
      if (p.fallback_nan) return 2;
      return 3;
    }
    b = b_new;
    state->b = b;
  }
  const double *x1f = g.x1f + static_cast<size_t>(b) * (n_i + 1), *x2f = g.x2f + static_cast<size_t>(b) * (n_j + 1);
  const double *x3f = g.x3f + static_cast<size_t>(b) * (n_k + 1);
  const double *x1v = g.x1v + static_cast<size_t>(b) * n_i, *x2v = g.x2v + static_cast<size_t>(b) * n_j, *x3v = g.x3v + static_cast<size_t>(b) * n_k;
  int i, j, k;  // :458-466
  for (i = 0; i < n_i; i++)
    if (x1f[i + 1] >= x1) break;
  for (j = 0; j < n_j; j++)
    if (x2f[j + 1] >= x2) break;
  for (k = 0; k < n_k; k++)
    if (x3f[k + 1] >= x3) break;
  bool code_kappa = p.plasma_model == BL_PLASMA_CODE_KAPPA;
  *gathered = true;
  // the nine variables in the order of Prims; without code_kappa the entropy stays 0
  const int vars[9] = {g.ind_rho, g.ind_pgas, g.ind_kappa, g.ind_uu1, g.ind_uu2, g.ind_uu3, g.ind_bb1, g.ind_bb2, g.ind_bb3};
  float *dst[9] = {&out->rho, &out->pgas, &out->kappa, &out->uu1, &out->uu2, &out->uu3, &out->bb1, &out->bb2, &out->bb3};
  auto slice = [&](int t) -> const bl_grid_desc & { return slow ? *o.slow_grids[t] : g; };
  if (not p.simulation_interp) {  // :710-786
    for (int v = 0; v < 9; v++) {
      if (v == 2 and not code_kappa) {
        *dst[v] = 0.0f;
        continue;
      }
      if (not slow_interp)
        *dst[v] = GridVal(slice(t_ind), vars[v], b, k, j, i);
      else {
        double val_1 = static_cast<double>(GridVal(slice(t_ind), vars[v], b, k, j, i));
        double val_2 = static_cast<double>(GridVal(slice(t_ind + 1), vars[v], b, k, j, i));
        *dst[v] = static_cast<float>((1.0 - t_frac) * val_1 + t_frac * val_2);
      }
    }
    return 0;
  }
  if (p.simulation_block_interp) {   // inter-block interpolation (:505-546, :912-1033)
    int i_m = x1 >= x1v[i] ? i : i - 1;
    int j_m = x2 >= x2v[j] ? j : j - 1;
    int k_m = x3 >= x3v[k] ? k : k - 1;
    int i_p = i_m + 1, j_p = j_m + 1, k_p = k_m + 1;
    // :520-522 read x1v(b, i + 1) with i = n_i - 1 at a block's upper edge: in the reference's Array that is the
    // first centre of the NEXT block's row; for the last block it is past the allocation - undefined (status 4)
    if ((i_p == n_i or j_p == n_j or k_p == n_k) and b == n_b - 1) return 4;
    double x1_m = i_m == -1 ? 2.0 * x1f[i] - x1v[i] : x1v[i_m];
    double x2_m = j_m == -1 ? 2.0 * x2f[j] - x2v[j] : x2v[j_m];
    double x3_m = k_m == -1 ? 2.0 * x3f[k] - x3v[k] : x3v[k_m];
    double x1_p = i_p == n_i ? 2.0 * x1v[i + 1] - x1v[i] : x1v[i_p];
    double x2_p = j_p == n_j ? 2.0 * x2v[j + 1] - x2v[j] : x2v[j_p];
    double x3_p = k_p == n_k ? 2.0 * x3v[k + 1] - x3v[k] : x3v[k_p];
    double f_i = (x1 - x1_m) / (x1_p - x1_m);
    double f_j = (x2 - x2_m) / (x2_p - x2_m);
    double f_k = (x3 - x3_m) / (x3_p - x3_m);
    int inds[8][4];
    bool found = true;
    for (int corner = 0; corner < 8; corner++)
      found = FindNearbyInds(o, b, (corner & 4) ? k_p : k_m, (corner & 2) ? j_p : j_m, (corner & 1) ? i_p : i_m, k, j, i, x3, x2, x1,
                             inds[corner]) and found;
    if (not found) return 5;   // "Grid interpolation failed."
    auto advanced = [&](const bl_grid_desc &gs, int v) {   // InterpolateAdvanced (:1365-1386) + the <= 0 rule (:936-945)
      double vals[8];
      for (int c = 0; c < 8; c++) vals[c] = static_cast<double>(GridVal(gs, vars[v], inds[c][0], inds[c][1], inds[c][2], inds[c][3]));
      double val = (1.0 - f_k) * (1.0 - f_j) * (1.0 - f_i) * vals[0]
          + (1.0 - f_k) * (1.0 - f_j) * f_i * vals[1] + (1.0 - f_k) * f_j * (1.0 - f_i) * vals[2]
          + (1.0 - f_k) * f_j * f_i * vals[3] + f_k * (1.0 - f_j) * (1.0 - f_i) * vals[4]
          + f_k * (1.0 - f_j) * f_i * vals[5] + f_k * f_j * (1.0 - f_i) * vals[6]
          + f_k * f_j * f_i * vals[7];
      if (v < 3 and val <= 0.0) val = vals[0];
      return val;
    };
    for (int v = 0; v < 9; v++) {
      if (v == 2 and not code_kappa) {
        *dst[v] = 0.0f;
        continue;
      }
      if (not slow_interp)
        *dst[v] = static_cast<float>(advanced(slice(t_ind), v));
      else {
        double val_1 = advanced(slice(t_ind), v);
        double val_2 = advanced(slice(t_ind + 1), v);
        *dst[v] = static_cast<float>((1.0 - t_frac) * val_1 + t_frac * val_2);
      }
    }
    return 0;
  }
  // intrablock interpolation (:485-490, :809-912)
  int i_m = i == 0 or (i != n_i - 1 and x1 >= x1v[i]) ? i : i - 1;
  int j_m = j == 0 or (j != n_j - 1 and x2 >= x2v[j]) ? j : j - 1;
  int k_m = k == 0 or (k != n_k - 1 and x3 >= x3v[k]) ? k : k - 1;
  double f_i = (x1 - x1v[i_m]) / (x1v[i_m + 1] - x1v[i_m]);
  double f_j = (x2 - x2v[j_m]) / (x2v[j_m + 1] - x2v[j_m]);
  double f_k = (x3 - x3v[k_m]) / (x3v[k_m + 1] - x3v[k_m]);
  auto spatial = [&](const bl_grid_desc &gs, int v) {   // InterpolateSimple + the <= 0 rule of rho, pgas, kappa
    double val = InterpolateSimple(gs, vars[v], b, k_m, j_m, i_m, f_k, f_j, f_i);
    if (v < 3 and val <= 0.0) val = static_cast<double>(GridVal(gs, vars[v], b, k_m, j_m, i_m));
    return val;
  };
  for (int v = 0; v < 9; v++) {
    if (v == 2 and not code_kappa) {
      *dst[v] = 0.0f;
      continue;
    }
    if (not slow_interp)
      *dst[v] = static_cast<float>(spatial(slice(t_ind), v));
    else {
      double val_1 = spatial(slice(t_ind), v);
      double val_2 = spatial(slice(t_ind + 1), v);
      *dst[v] = static_cast<float>((1.0 - t_frac) * val_1 + t_frac * val_2);
    }
  }
  return 0;
}

// std::cyl_bessel_k(n, x) for n = 0, 1, 2 as libstdc++ 11 evaluates it (GCC 11.4,
// <tr1/modified_bessel_func.tcc> __bessel_ik:75-258 through __cyl_bessel_k:305-318; the reference calls it at
// simulation_coefficients.cpp:537-539): Temme's series for x < 2, Steed's continued fraction otherwise, then
// upward recurrence. Only the K branch is restated: the continued fraction and recurrences for I_nu, which
// the header evaluates in the same call, do not enter K_nu. For integer order mu = nu - nl = 0, so
// __gamma_temme (bessel_function.tcc:100-119) gives gampl = gammi = 1 / tgamma(1) = 1, gam1 = -Euler's
// constant, gam2 = 1. libm calls (log, sinh, cosh, exp) go through M:: like everywhere else.
// simulation_coefficients.cpp:740-773: 2F1 through its Pfaff transformation, ten terms of the series
double Hypergeometric(double alpha, double beta, double gamma, double z) {
  const int k_max = 10;
  double a = alpha;
  double b = gamma - beta;
  double c = gamma;
  double x = z / (z - 1.0);
  double result = 1.0;
  double a_k = 1.0, b_k = 1.0, c_k = 1.0, xk = 1.0, k_factorial = 1.0;
  for (int k = 1; k <= k_max; k++) {
    a_k *= a + k - 1.0;
    b_k *= b + k - 1.0;
    c_k *= c + k - 1.0;
    xk *= x;
    k_factorial *= k;
    result += a_k * b_k * xk / (c_k * k_factorial);
  }
  result *= M::pow(1.0 - z, -alpha);
  return result;
}

double CylBesselK(int nl, double x) {
  if (std::isnan(x)) return std::numeric_limits<double>::quiet_NaN();
  if (x == 0.0) return std::numeric_limits<double>::infinity();
  const double eps = std::numeric_limits<double>::epsilon();
  const int max_iter = 15000;
  const double mu = 0.0, mu2 = 0.0;
  const double xi = 1.0 / x;
  const double xi2 = 2.0 * xi;
  double kmu, knu1;
  if (x < 2.0) {
    const double x2 = x / 2.0;
    const double pimu = Math::pi * mu;
    const double fact = std::abs(pimu) < eps ? 1.0 : pimu / M::sin(pimu);
    double d = -M::log(x2);
    double e = mu * d;
    const double fact2 = std::abs(e) < eps ? 1.0 : M::sinh(e) / e;
    const double gam1 = -0.5772156649015328606065120900824024L, gam2 = 1.0, gampl = 1.0, gammi = 1.0;
    double ff = fact * (gam1 * M::cosh(e) + gam2 * fact2 * d);
    double sum = ff;
    e = M::exp(e);
    double pp = e / (2.0 * gampl);
    double q = 1.0 / (2.0 * e * gammi);
    double c = 1.0;
    d = x2 * x2;
    double sum1 = pp;
    for (int i = 1; i <= max_iter; ++i) {
      ff = (i * ff + pp + q) / (i * i - mu2);
      c *= d / i;
      pp /= i - mu;
      q /= i + mu;
      const double del = c * ff;
      sum += del;
      const double del1 = c * (pp - i * ff);
      sum1 += del1;
      if (std::abs(del) < eps * std::abs(sum)) break;
    }
    kmu = sum;
    knu1 = sum1 * xi2;
  } else {
    double b = 2.0 * (1.0 + x);
    double d = 1.0 / b;
    double delh = d;
    double h = delh;
    double q1 = 0.0;
    double q2 = 1.0;
    double a1 = 0.25 - mu2;
    double c = a1;
    double q = c;
    double a = -a1;
    double s = 1.0 + q * delh;
    for (int i = 2; i <= max_iter; ++i) {
      a -= 2 * (i - 1);
      c = -a * c / i;
      const double qnew = (q1 - b * q2) / a;
      q1 = q2;
      q2 = qnew;
      q += c * qnew;
      b += 2.0;
      d = 1.0 / (b + a * d);
      delh = (b * d - 1.0) * delh;
      h += delh;
      const double dels = q * delh;
      s += dels;
      if (std::abs(dels / s) < eps) break;
    }
    h = a1 * h;
    kmu = std::sqrt(Math::pi / (2.0 * x)) * M::exp(-x) / s;
    knu1 = kmu * (mu + x + 0.5 - h) * xi;
  }
  for (int i = 1; i <= nl; ++i) {
    const double knutemp = (mu + i) * xi2 * knu1 + kmu;
    kmu = knu1;
    knu1 = knutemp;
  }
  return kmu;
}

// radiation_geometry.cpp:274-412: Christoffel symbols of the Cartesian Kerr-Schild geodesic metric
void GeodesicConnection(const Oracle &o, double x, double y, double z, double connection[4][4][4]) {
  if (o.ray_flat) {
    for (int mu = 0; mu < 4; mu++)
      for (int alpha = 0; alpha < 4; alpha++)
        for (int beta = 0; beta < 4; beta++) connection[mu][alpha][beta] = 0.0;
    return;
  }
  double bh_a = o.bh_a, bh_m = o.bh_m;
  double a2 = bh_a * bh_a;
  double rr2 = x * x + y * y + z * z;
  double r2 = 0.5 * (rr2 - a2 + M::hypot(rr2 - a2, 2.0 * bh_a * z));
  double r = std::sqrt(r2);
  double f = 2.0 * bh_m * r2 * r / (r2 * r2 + a2 * z * z);
  double l[4] = {-1.0, (r * x + bh_a * y) / (r2 + a2), (r * y - bh_a * x) / (r2 + a2), z / r};
  double gcon[4][4];
  for (int mu = 0; mu < 4; mu++)
    for (int nu = 0; nu < 4; nu++) gcon[mu][nu] = -f * l[mu] * l[nu] + (mu == nu ? (mu == 0 ? -1.0 : 1.0) : 0.0);
  // the reference writes the off-diagonal entries without the "+ 0": same bits
  for (int mu = 0; mu < 4; mu++)
    for (int nu = 0; nu < 4; nu++)
      if (mu != nu) gcon[mu][nu] = -f * l[mu] * l[nu];
  gcon[0][0] = -f * l[0] * l[0] - 1.0;
  double dr[4], df[4], dl[4][4];   // dr[a] = d r / d x^a, dl[mu][a] = d l_mu / d x^a (a = 1..3)
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
  // dgcov[a][mu][nu] = d g_{mu nu} / d x^a = +-(df l_mu l_nu + f dl_mu l_nu + f l_mu dl_nu): minus exactly when
  // one of mu, nu is 0 (l_0 of the covariant null vector is +1, the table above holds the contravariant -1)
  double dgcov[4][4][4] = {};
  for (int a = 1; a < 4; a++)
    for (int mu = 0; mu < 4; mu++)
      for (int nu = 0; nu < 4; nu++) {
        double val = df[a] * l[mu] * l[nu] + f * dl[mu][a] * l[nu] + f * l[mu] * dl[nu][a];
        dgcov[a][mu][nu] = ((mu == 0) != (nu == 0)) ? -val : +val;
      }
  for (int mu = 0; mu < 4; mu++)
    for (int alpha = 0; alpha < 4; alpha++)
      for (int beta = 0; beta < 4; beta++) {
        connection[mu][alpha][beta] = 0.0;
        for (int nu = 0; nu < 4; nu++)
          connection[mu][alpha][beta] +=
              0.5 * gcon[mu][nu] * (dgcov[alpha][beta][nu] + dgcov[beta][alpha][nu] - dgcov[nu][alpha][beta]);
      }
}

// simulation_coefficients.cpp:253-700 for one sample (thermal electrons, unpolarized outputs).
// j[l], alpha[l] must be zero on entry (as after Array::Zero(), :226-229); cell[7] NaN on entry.
// pol (polarized transfer only): the six arrays j_Q, j_V, alpha_Q, alpha_V, rho_Q, rho_V of this sample, same
// stride between frequencies, zero on entry.
void SimulationCoefficientsOne(const Oracle &o, const double pos[4], const double kcov_in[4],
                               const Prims &s, double momentum_factor, double *j, double *alpha,
                               int stride, double cell[num_cell_values], double *const pol[6] = nullptr) {
  const bl_params &p = *o.p;
  double d_unit = p.simulation_rho_cgs;  // :237-239
  double e_unit = d_unit * Physics::c * Physics::c;
  double b_unit = std::sqrt(4.0 * Math::pi * e_unit);
  double gcov_sim[4][4], gcon_sim[4][4], gcov[4][4], gcon[4][4], jacobian[4][4], tetrad[4][4];
  double x1 = pos[1], x2 = pos[2], x3 = pos[3];
  double kcov[4] = {kcov_in[0], kcov_in[1], kcov_in[2], kcov_in[3]};
  double rho = s.rho;
  double pgas = s.pgas;
  double kappa = 0.0;
  if (p.plasma_model == BL_PLASMA_CODE_KAPPA) kappa = s.kappa;
  double uu1_sim = s.uu1, uu2_sim = s.uu2, uu3_sim = s.uu3;
  double bb1_sim = s.bb1, bb2_sim = s.bb2, bb3_sim = s.bb3;

  double rho_cgs = rho * d_unit;  // :287-290
  double pgas_cgs = pgas * e_unit;
  double n_cgs = rho_cgs / (p.plasma_mu * Physics::m_p);
  double n_e_cgs = n_cgs / (1.0 + 1.0 / p.plasma_ne_ni);

  CovariantSimulationMetric(o, x1, x2, x3, gcov_sim);
  ContravariantSimulationMetric(o, x1, x2, x3, gcon_sim);

  double uu0_sim = std::sqrt(1.0 + gcov_sim[1][1] * uu1_sim * uu1_sim  // :297-313
      + 2.0 * gcov_sim[1][2] * uu1_sim * uu2_sim + 2.0 * gcov_sim[1][3] * uu1_sim * uu3_sim
      + gcov_sim[2][2] * uu2_sim * uu2_sim + 2.0 * gcov_sim[2][3] * uu2_sim * uu3_sim
      + gcov_sim[3][3] * uu3_sim * uu3_sim);
  double lapse_sim = 1.0 / std::sqrt(-gcon_sim[0][0]);
  double shift1_sim = -gcon_sim[0][1] / gcon_sim[0][0];
  double shift2_sim = -gcon_sim[0][2] / gcon_sim[0][0];
  double shift3_sim = -gcon_sim[0][3] / gcon_sim[0][0];
  double ucon_sim[4];
  ucon_sim[0] = uu0_sim / lapse_sim;
  ucon_sim[1] = uu1_sim - shift1_sim * uu0_sim / lapse_sim;
  ucon_sim[2] = uu2_sim - shift2_sim * uu0_sim / lapse_sim;
  ucon_sim[3] = uu3_sim - shift3_sim * uu0_sim / lapse_sim;
  double ucov_sim[4] = {};
  for (int mu = 0; mu < 4; mu++)
    for (int nu = 0; nu < 4; nu++) ucov_sim[mu] += gcov_sim[mu][nu] * ucon_sim[nu];

  double bcon_sim[4];  // :316-330
  bcon_sim[0] = ucov_sim[1] * bb1_sim + ucov_sim[2] * bb2_sim + ucov_sim[3] * bb3_sim;
  bcon_sim[1] = (bb1_sim + bcon_sim[0] * ucon_sim[1]) / ucon_sim[0];
  bcon_sim[2] = (bb2_sim + bcon_sim[0] * ucon_sim[2]) / ucon_sim[0];
  bcon_sim[3] = (bb3_sim + bcon_sim[0] * ucon_sim[3]) / ucon_sim[0];
  double bcov_sim[4] = {};
  for (int mu = 0; mu < 4; mu++)
    for (int nu = 0; nu < 4; nu++) bcov_sim[mu] += gcov_sim[mu][nu] * bcon_sim[nu];
  double b_sq = 0.0;
  for (int mu = 0; mu < 4; mu++) b_sq += bcov_sim[mu] * bcon_sim[mu];
  double bb_cgs = std::sqrt(b_sq) * b_unit;
  double sigma = b_sq / rho;
  double beta_inv = b_sq / (2.0 * pgas);

  double kb_tt_e_cgs = std::numeric_limits<double>::quiet_NaN();  // :333-358
  double theta_e = std::numeric_limits<double>::quiet_NaN();
  if (o.plasma_thermal_frac != 0.0 and p.plasma_model == BL_PLASMA_TI_TE_BETA) {
    double tti_tte = (p.plasma_rat_high + p.plasma_rat_low * beta_inv * beta_inv) / (1.0 + beta_inv * beta_inv);
    double kb_tt_tot_cgs = p.plasma_mu * Physics::m_p * pgas_cgs / rho_cgs;
    if (p.plasma_use_p)
      kb_tt_e_cgs = (1.0 + p.plasma_ne_ni) / (tti_tte + p.plasma_ne_ni) * kb_tt_tot_cgs;
    else {
      kb_tt_e_cgs = (1.0 + p.plasma_ne_ni) * kb_tt_tot_cgs / (o.g->plasma_gamma - 1.0);
      kb_tt_e_cgs /= tti_tte / (o.g->plasma_gamma_i - 1.0) + p.plasma_ne_ni / (o.g->plasma_gamma_e - 1.0);
    }
    theta_e = kb_tt_e_cgs / (Physics::m_e * Physics::c * Physics::c);
  }
  if (o.plasma_thermal_frac != 0.0 and p.plasma_model == BL_PLASMA_CODE_KAPPA) {
    double mu_e = p.plasma_mu * (1.0 + 1.0 / p.plasma_ne_ni);
    double rho_e = rho * Physics::m_e / (mu_e * Physics::m_p);
    double rho_kappa_e_cbrt = M::cbrt(rho_e * kappa);
    theta_e = 1.0 / 5.0 * (std::sqrt(1.0 + 25.0 * rho_kappa_e_cbrt * rho_kappa_e_cbrt) - 1.0);
    kb_tt_e_cgs = theta_e * Physics::m_e * Physics::c * Physics::c;
  }

  if ((p.cut_rho_min >= 0.0 and rho_cgs < p.cut_rho_min)  // :361-375
      or (p.cut_rho_max >= 0.0 and rho_cgs > p.cut_rho_max)
      or (p.cut_n_e_min >= 0.0 and n_e_cgs < p.cut_n_e_min)
      or (p.cut_n_e_max >= 0.0 and n_e_cgs > p.cut_n_e_max)
      or (p.cut_p_gas_min >= 0.0 and pgas_cgs < p.cut_p_gas_min)
      or (p.cut_p_gas_max >= 0.0 and pgas_cgs > p.cut_p_gas_max)
      or (p.cut_theta_e_min >= 0.0 and theta_e < p.cut_theta_e_min)
      or (p.cut_theta_e_max >= 0.0 and theta_e > p.cut_theta_e_max)
      or (p.cut_b_min >= 0.0 and bb_cgs < p.cut_b_min)
      or (p.cut_b_max >= 0.0 and bb_cgs > p.cut_b_max)
      or (p.cut_sigma_min >= 0.0 and sigma < p.cut_sigma_min)
      or (p.cut_sigma_max >= 0.0 and sigma > p.cut_sigma_max)
      or (p.cut_beta_inverse_min >= 0.0 and beta_inv < p.cut_beta_inverse_min)
      or (p.cut_beta_inverse_max >= 0.0 and beta_inv > p.cut_beta_inverse_max))
    return;

  if (p.image_lambda_ave or p.image_emission_ave or p.image_tau_int or o.render_num_images > 0) {  // :378-387
    cell[0] = rho_cgs;
    cell[1] = n_e_cgs;
    cell[2] = pgas_cgs;
    cell[3] = theta_e;
    cell[4] = bb_cgs;
    cell[5] = sigma;
    cell[6] = beta_inv;
  }
  if (not (p.image_light or p.image_emission or p.image_tau or p.image_emission_ave or p.image_tau_int))
    return;
  if (bb1_sim == 0.0 and bb2_sim == 0.0 and bb3_sim == 0.0) return;  // :394

  CoordinateJacobian(o, x1, x2, x3, jacobian);  // :398-408
  double ucon[4] = {};
  for (int mu = 0; mu < 4; mu++)
    for (int nu = 0; nu < 4; nu++) ucon[mu] += jacobian[mu][nu] * ucon_sim[nu];
  double bcon[4] = {};
  for (int mu = 0; mu < 4; mu++)
    for (int nu = 0; nu < 4; nu++) bcon[mu] += jacobian[mu][nu] * bcon_sim[nu];
  CovariantGeodesicMetric(o, x1, x2, x3, gcov);  // :411-428
  ContravariantGeodesicMetric(o, x1, x2, x3, gcon);
  double kcon[4] = {};
  for (int mu = 0; mu < 4; mu++)
    for (int nu = 0; nu < 4; nu++) kcon[mu] += gcon[mu][nu] * kcov[nu];
  double ucov[4] = {};
  for (int mu = 0; mu < 4; mu++)
    for (int nu = 0; nu < 4; nu++) ucov[mu] += gcov[mu][nu] * ucon[nu];
  double bcov[4] = {};
  for (int mu = 0; mu < 4; mu++)
    for (int nu = 0; nu < 4; nu++) bcov[mu] += gcov[mu][nu] * bcon[nu];
  Tetrad(ucon, ucov, kcon, kcov, bcon, gcov, gcon, tetrad);  // :431

  double k_tet_1 = 0.0, k_tet_2 = 0.0, k_tet_3 = 0.0, b_tet_1 = 0.0, b_tet_2 = 0.0, b_tet_3 = 0.0;  // :434-455
  for (int mu = 0; mu < 4; mu++) {
    k_tet_1 += tetrad[1][mu] * kcov[mu];
    k_tet_2 += tetrad[2][mu] * kcov[mu];
    k_tet_3 += tetrad[3][mu] * kcov[mu];
    b_tet_1 += tetrad[1][mu] * bcov[mu];
    b_tet_2 += tetrad[2][mu] * bcov[mu];
    b_tet_3 += tetrad[3][mu] * bcov[mu];
  }
  double k_sq_tet = k_tet_1 * k_tet_1 + k_tet_2 * k_tet_2 + k_tet_3 * k_tet_3;
  double b_sq_tet = b_tet_1 * b_tet_1 + b_tet_2 * b_tet_2 + b_tet_3 * b_tet_3;
  double k_b_tet = k_tet_1 * b_tet_1 + k_tet_2 * b_tet_2 + k_tet_3 * b_tet_3;
  double cos2_theta_b = std::min(k_b_tet * k_b_tet / (k_sq_tet * b_sq_tet), 1.0);
  double sin2_theta_b = 1.0 - cos2_theta_b;
  double sin_theta_b = std::sqrt(sin2_theta_b);
  double cos_theta_b = std::sqrt(cos2_theta_b) * (k_b_tet >= 0.0 ? 1.0 : -1.0);
  const bool polarized = o.image_polarization and pol != nullptr;   // image_light and image_polarization

  int nf = p.image_num_frequencies;
  for (int l = 0; l < nf; l++) {  // :458-605 (thermal and power-law electrons)
    double nu_cgs = 0.0;
    for (int mu = 0; mu < 4; mu++) nu_cgs -= kcov[mu] * ucon[mu];
    nu_cgs *= o.image_frequencies[l] * momentum_factor;
    double nu_2_cgs = nu_cgs * nu_cgs;
    double nu_c_cgs = Physics::e * bb_cgs / (2.0 * Math::pi * Physics::m_e * Physics::c);
    double nu_s_cgs = 2.0 / 9.0 * nu_c_cgs * theta_e * theta_e * sin_theta_b;
    double j_i_val = 0.0;
    if (o.plasma_thermal_frac != 0.0) {
      double xx = nu_cgs / nu_s_cgs;
      double xx_1_2 = std::sqrt(xx);
      double xx_1_3 = M::cbrt(xx);
      double xx_1_6 = std::sqrt(xx_1_3);
      double coefficient = o.plasma_thermal_frac * n_e_cgs * Physics::e * Physics::e * nu_c_cgs
          / (Physics::c * nu_2_cgs) * M::exp(-xx_1_3);
      double var_a = Math::sqrt2 * Math::pi / 27.0 * sin_theta_b;
      double var_b = pow_2_11_12;
      double var_c = xx_1_2 + var_b * xx_1_6;
      j_i_val = coefficient * var_a * var_c * var_c;
      if (p.image_light or p.image_emission or p.image_emission_ave) j[l * stride] = j_i_val;
      if (polarized) {  // :485-495
        double var_d = (7.0 * M::pow(theta_e, 0.96) + 35.0) / (10.0 * M::pow(theta_e, 0.96) + 75.0) * var_b;
        double var_e = xx_1_2 + var_d * xx_1_6;
        double var_f = cos_theta_b / theta_e;
        double var_g = Math::pi / 3.0 + Math::pi / 3.0 * xx_1_3 + 2.0 / 300.0 * xx_1_2
            + 2.0 / 19.0 * Math::pi * xx_1_3 * xx_1_3;
        pol[0][l * stride] = -coefficient * var_a * var_e * var_e;
        pol[1][l * stride] = coefficient * var_f * var_g;
      }
    }
    if (o.plasma_thermal_frac != 0.0) {
      double b_nu_nu_3_cgs = 2.0 * Physics::h / (Physics::c * Physics::c)
          / M::expm1(Physics::h * nu_cgs / kb_tt_e_cgs);
      if (p.image_light or p.image_tau or p.image_tau_int) alpha[l * stride] = j_i_val / b_nu_nu_3_cgs;
      if (polarized) {
        pol[2][l * stride] = pol[0][l * stride] / b_nu_nu_3_cgs;
        pol[3][l * stride] = pol[1][l * stride] / b_nu_nu_3_cgs;
      }
      if ((p.image_light or p.image_tau or p.image_tau_int)
          and 1.0 / (alpha[l * stride] * alpha[l * stride]) == std::numeric_limits<double>::infinity()) {
        alpha[l * stride] = 0.0;
        if (polarized) {
          pol[2][l * stride] = 0.0;
          pol[3][l * stride] = 0.0;
        }
      }
    }
    if (o.plasma_thermal_frac != 0.0 and polarized) {  // rotativities, :527-553
      double coefficient_q = -o.plasma_thermal_frac * n_e_cgs * Physics::e * Physics::e * nu_c_cgs * nu_c_cgs
          * sin2_theta_b / (Physics::m_e * Physics::c * nu_2_cgs);
      double coefficient_v = o.plasma_thermal_frac * 2.0 * n_e_cgs * Physics::e * Physics::e * nu_c_cgs
          * cos_theta_b / (Physics::m_e * Physics::c * nu_cgs);
      double factor_q = 0.0;
      double factor_v = 1.0;
      if (theta_e >= 0.01) {   // theta_e_zero, radiation_integrator.hpp:190
        double kk_0 = CylBesselK(0, 1.0 / theta_e);
        double kk_1 = CylBesselK(1, 1.0 / theta_e);
        double kk_2 = CylBesselK(2, 1.0 / theta_e);
        double xx = nu_cgs / nu_s_cgs;
        double xx_neg_1_2 = 1.0 / std::sqrt(xx);
        double var_a = 2.011 * M::exp(-19.78 * M::pow(xx, -0.5175));
        double var_b = M::cos(39.89 * xx_neg_1_2) * M::exp(-70.16 * M::pow(xx, -0.6));
        double var_c = 0.011 * M::exp(-1.69 * xx_neg_1_2);
        double var_d = 0.003135 * M::pow(xx, 4.0 / 3.0);
        double var_e = 0.5 * (1.0 + M::tanh(10.0 * M::log(0.6648 * xx_neg_1_2)));
        double f_0 = var_a - var_b - var_c;
        double f_m = f_0 + (var_c - var_d) * var_e;
        double delta_jj_5 = 0.4379 * M::log(1.0 + 1.3414 * M::pow(xx, -0.7515));
        factor_q = f_m * (kk_1 / kk_2 + 6.0 * theta_e);
        factor_v = (kk_0 - delta_jj_5) / kk_2;
        factor_v = factor_v < 0.0 or factor_v > 1.0 ? 1.0 : factor_v;
      }
      pol[4][l * stride] = coefficient_q * factor_q;
      pol[5][l * stride] = coefficient_v * factor_v;
    }
    // power-law electrons (:556-584, unpolarized part)
    if (p.plasma_power_frac != 0.0 and (p.image_light or p.image_emission or p.image_emission_ave)) {
      double var_a = M::pow(nu_cgs / (nu_c_cgs * sin_theta_b), -(p.plasma_p - 1.0) / 2.0);
      double coefficient = p.plasma_power_frac * n_e_cgs * Physics::e * Physics::e * nu_c_cgs
          / (Physics::c * nu_2_cgs) * o.power_jj * sin_theta_b * var_a;
      j[l * stride] += coefficient;
      if (polarized) {
        double var_b = cos_theta_b / sin_theta_b;
        double var_c = 1.0 / std::sqrt(nu_cgs / (3.0 * nu_c_cgs * sin_theta_b));
        pol[0][l * stride] += coefficient * o.power_pol[0];
        pol[1][l * stride] += coefficient * o.power_pol[1] * var_b * var_c;
      }
    }
    if (p.plasma_power_frac != 0.0 and (p.image_light or p.image_tau or p.image_tau_int)) {
      double var_a = M::pow(nu_cgs / (nu_c_cgs * sin_theta_b), -(p.plasma_p + 2.0) / 2.0);
      double coefficient = p.plasma_power_frac * n_e_cgs * Physics::e * Physics::e
          / (Physics::m_e * Physics::c) * o.power_aa * var_a;
      alpha[l * stride] += coefficient;
      if (polarized) {
        double var_b = M::pow(3.1 * M::pow(sin_theta_b, -1.92) - 3.1, 0.512);
        double var_c = 1.0 / std::sqrt(nu_cgs / (nu_c_cgs * sin_theta_b));
        double var_d = cos_theta_b >= 0.0 ? 1.0 : -1.0;
        pol[2][l * stride] += coefficient * o.power_pol[2];
        pol[3][l * stride] += coefficient * o.power_pol[3] * var_b * var_c * var_d;
      }
    }
    if (p.plasma_power_frac != 0.0 and polarized) {  // :587-605
      double var_a = n_e_cgs * Physics::e * Physics::e * nu_cgs / (Physics::m_e * Physics::c * nu_c_cgs * sin_theta_b);
      double var_b = nu_c_cgs * sin_theta_b / nu_cgs;
      double var_c = var_b * var_b;
      double var_d = var_c * var_b;
      double var_e = 1.0 - M::pow(2.0 * nu_c_cgs * p.plasma_gamma_min * p.plasma_gamma_min * sin_theta_b / (3.0 * nu_cgs),
                                  p.plasma_p / 2.0 - 1.0);
      double var_f = cos_theta_b / sin_theta_b;
      double coefficient = p.plasma_power_frac * o.power_pol[4] * var_a;
      pol[4][l * stride] += coefficient * o.power_pol[5] * var_d * var_e;
      pol[5][l * stride] += coefficient * o.power_pol[6] * var_c * var_f;
    }
    // kappa-distribution electrons (:607-698); only reached in polarized runs (image_light is then set)
    if (p.plasma_kappa_frac != 0.0 and (p.image_light or p.image_emission or p.image_emission_ave)) {
      const auto &kk = o.kappa;
      double nu_kappa_cgs = nu_c_cgs * p.plasma_w * p.plasma_w * p.plasma_kappa * p.plasma_kappa * sin_theta_b;
      double xx = nu_cgs / nu_kappa_cgs;
      double var_a = p.plasma_kappa_frac * n_e_cgs * Physics::e * Physics::e * nu_c_cgs / (Physics::c * nu_2_cgs);
      double var_b = M::cbrt(xx) * sin_theta_b;
      double var_c = M::pow(xx, -(p.plasma_kappa - 2.0) / 2.0) * sin_theta_b;
      double coefficient_low = kk.jj_low * var_a * var_b;
      double coefficient_high = kk.jj_high * var_a * var_c;
      j[l * stride] += M::pow(M::pow(coefficient_low, -kk.jj_x_i) + M::pow(coefficient_high, -kk.jj_x_i), -1.0 / kk.jj_x_i);
      if (polarized) {
        double var_d = M::pow(M::pow(sin_theta_b, -2.4) - 1.0, 0.48);
        double var_e = M::pow(xx, -0.35);
        double var_f = M::pow(M::pow(sin_theta_b, -2.5) - 1.0, 0.44);
        double var_g = 1.0 / std::sqrt(xx);
        double var_h = cos_theta_b >= 0.0 ? 1.0 : -1.0;
        double jj_q_low = coefficient_low * kk.jj_low_q;
        double jj_v_low = coefficient_low * kk.jj_low_v * var_d * var_e;
        double jj_q_high = coefficient_high * kk.jj_high_q;
        double jj_v_high = coefficient_high * kk.jj_high_v * var_f * var_g;
        pol[0][l * stride] -= M::pow(M::pow(jj_q_low, -kk.jj_x_q) + M::pow(jj_q_high, -kk.jj_x_q), -1.0 / kk.jj_x_q);
        pol[1][l * stride] += M::pow(M::pow(jj_v_low, -kk.jj_x_v) + M::pow(jj_v_high, -kk.jj_x_v), -1.0 / kk.jj_x_v) * var_h;
      }
    }
    if (p.plasma_kappa_frac != 0.0 and (p.image_light or p.image_tau or p.image_tau_int)) {
      const auto &kk = o.kappa;
      double nu_kappa_cgs = nu_c_cgs * p.plasma_w * p.plasma_w * p.plasma_kappa * p.plasma_kappa * sin_theta_b;
      double xx = nu_cgs / nu_kappa_cgs;
      double var_a = p.plasma_kappa_frac * n_e_cgs * Physics::e * Physics::e / (Physics::m_e * Physics::c);
      double var_b = M::pow(xx, -2.0 / 3.0);
      double var_c = M::pow(xx, -(1.0 + p.plasma_kappa) / 2.0);
      double coefficient_low = kk.aa_low * var_a * var_b;
      double coefficient_high = kk.aa_high * var_a * var_c;
      double aa_i_low = coefficient_low;
      double aa_i_high = coefficient_high * kk.aa_high_i;
      alpha[l * stride] += M::pow(M::pow(aa_i_low, -kk.aa_x_i) + M::pow(aa_i_high, -kk.aa_x_i), -1.0 / kk.aa_x_i);
      if (polarized) {
        double var_d = M::pow(M::pow(sin_theta_b, -2.28) - 1.0, 0.446);
        double var_e = M::pow(xx, -0.35);
        double var_f = std::sqrt(M::pow(sin_theta_b, -2.05) - 1.0);
        double var_g = 1.0 / std::sqrt(xx);
        double var_h = cos_theta_b >= 0.0 ? 1.0 : -1.0;
        double aa_q_low = coefficient_low * kk.aa_low_q;
        double aa_v_low = coefficient_low * kk.aa_low_v * var_d * var_e;
        double aa_q_high = coefficient_high * kk.aa_high_q;
        double aa_v_high = coefficient_high * kk.aa_high_v * var_f * var_g;
        pol[2][l * stride] -= M::pow(M::pow(aa_q_low, -kk.aa_x_q) + M::pow(aa_q_high, -kk.aa_x_q), -1.0 / kk.aa_x_q);
        pol[3][l * stride] += M::pow(M::pow(aa_v_low, -kk.aa_x_v) + M::pow(aa_v_high, -kk.aa_x_v), -1.0 / kk.aa_x_v) * var_h;
      }
    }
    if (p.plasma_kappa_frac != 0.0 and polarized) {  // rotativities, :670-698
      const auto &kk = o.kappa;
      double nu_kappa_cgs = nu_c_cgs * p.plasma_w * p.plasma_w * p.plasma_kappa * p.plasma_kappa * sin_theta_b;
      double xx = nu_cgs / nu_kappa_cgs;
      double var_a = -p.plasma_kappa_frac * n_e_cgs * Physics::e * Physics::e * nu_c_cgs * nu_c_cgs * sin2_theta_b
          / (Physics::m_e * Physics::c * nu_2_cgs);
      double var_b = p.plasma_kappa_frac * 2.0 * n_e_cgs * Physics::e * Physics::e * nu_c_cgs * cos_theta_b
          / (Physics::m_e * Physics::c * nu_cgs);
      double var_c = 1.0 / std::sqrt(xx);
      double rho_q_low = var_a * kk.rho_q_low_a * (1.0 - M::exp(kk.rho_q_low_b * M::pow(xx, 0.84))
          - M::sin(kk.rho_q_low_c * xx) * M::exp(kk.rho_q_low_d * M::pow(xx, kk.rho_q_low_e)));
      double rho_q_high = var_a * kk.rho_q_high_a * (1.0 - M::exp(kk.rho_q_high_b * M::pow(xx, 0.84))
          - M::sin(kk.rho_q_high_c * xx) * M::exp(kk.rho_q_high_d * M::pow(xx, kk.rho_q_high_e)));
      double rho_v_low = kk.rho_v * var_b * kk.rho_v_low_a * (1.0 - 0.17 * M::log(1.0 + kk.rho_v_low_b * var_c));
      double rho_v_high = kk.rho_v * var_b * kk.rho_v_high_a * (1.0 - 0.17 * M::log(1.0 + kk.rho_v_high_b * var_c));
      pol[4][l * stride] += (1.0 - kk.rho_frac) * rho_q_low + kk.rho_frac * rho_q_high;
      pol[5][l * stride] += (1.0 - kk.rho_frac) * rho_v_low + kk.rho_frac * rho_v_high;
    }
  }
}

// formula_coefficients.cpp:62-180 for one sample. j, alpha zero on entry.
void FormulaCoefficientsOne(const Oracle &o, const double pos[4], const double kcov[4],
                            double momentum_factor, double *j, double *alpha, int stride) {
  const bl_params &p = *o.p;
  double bh_a = o.bh_a, bh_m = o.bh_m;
  double x = pos[1], y = pos[2], z = pos[3];
  double k_0 = kcov[0], k_1 = kcov[1], k_2 = kcov[2], k_3 = kcov[3];
  double r = RadialGeodesicCoordinate(o, x, y, z);
  if (GeometricCut(o, x, y, z, r)) return;
  double rr = std::sqrt(r * r - z * z);
  double cth = z / r;
  double sth = std::sqrt(1.0 - cth * cth);
  double ph = M::atan2(y, x) - M::atan(bh_a / r);
  double sph = M::sin(ph);
  double cph = M::cos(ph);
  double delta = r * r - 2.0 * bh_m * r + bh_a * bh_a;
  double sigma = r * r + bh_a * bh_a * cth * cth;
  double gtt_bl = -(1.0 + 2.0 * bh_m * r * (r * r + bh_a * bh_a) / (delta * sigma));
  double gtph_bl = -2.0 * bh_m * bh_a * r / (delta * sigma);
  double grr_bl = delta / sigma;
  double gthth_bl = 1.0 / sigma;
  double gphph_bl = (sigma - 2.0 * bh_m * r) / (delta * sigma * sth * sth);
  double ll = p.formula_l0 / (1.0 + rr) * M::pow(rr, 1.0 + p.formula_q);
  double u_norm = 1.0 / std::sqrt(-gtt_bl + 2.0 * gtph_bl * ll - gphph_bl * ll * ll);
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
  double u0 = ut;
  double u1 = sth * cph * ur + cth * (r * cph - bh_a * sph) * uth + sth * (-r * sph - bh_a * cph) * uph;
  double u2 = sth * sph * ur + cth * (r * sph + bh_a * cph) * uth + sth * (r * cph - bh_a * sph) * uph;
  double u3 = cth * ur - r * sth * uth;
  double n_n0_fluid = M::exp(-0.5 * (r * r / (p.formula_r0 * p.formula_r0) + p.formula_h * p.formula_h * cth * cth));
  for (int l = 0; l < p.image_num_frequencies; l++) {
    double nu_fluid_cgs = -(u0 * k_0 + u1 * k_1 + u2 * k_2 + u3 * k_3) * o.image_frequencies[l] * momentum_factor;
    double j_nu_fluid_cgs = p.formula_cn0 * n_n0_fluid * M::pow(nu_fluid_cgs / p.formula_nup, -p.formula_alpha);
    j[l * stride] = j_nu_fluid_cgs / (nu_fluid_cgs * nu_fluid_cgs);
    double alpha_nu_fluid_cgs = p.formula_a * p.formula_cn0 * n_n0_fluid
        * M::pow(nu_fluid_cgs / p.formula_nup, -p.formula_beta - p.formula_alpha);
    alpha[l * stride] = alpha_nu_fluid_cgs * nu_fluid_cgs;
  }
}

// unpolarized.cpp:53-208 for one pixel. image_col[q] = image(q, m).
void IntegrateUnpolarizedOne(const Oracle &o, const RayBuffers &b, int num_steps, int max_steps,
                             double momentum_factor, double *image_col) {
  const bl_params &p = *o.p;
  constexpr double delta_tau_max = 100.0;  // radiation_integrator.hpp:191
  double x_unit = Physics::gg_msun * o.mass_msun / (Physics::c * Physics::c);
  double t_unit = x_unit / Physics::c;
  int nf = p.image_num_frequencies;
  for (int q = 0; q < o.image_num_quantities; q++) image_col[q] = 0.0;
  for (int l = 0; l < nf; l++) {
    double integrated_lambda = 0.0;
    double integrated_emission = 0.0;
    double x1_init = b.sample_pos[1], x2_init = b.sample_pos[2], x3_init = b.sample_pos[3];
    bool plane_sign = o.cam_x[1] * x1_init + o.cam_x[2] * x2_init + o.cam_x[3] * x3_init > 0.0;
    int crossings_count = 0;
    for (int n = 0; n < num_steps; n++) {
      double delta_lambda = b.sample_len[n];
      double delta_lambda_cgs = delta_lambda * x_unit / (o.image_frequencies[l] * momentum_factor);
      double t_cgs = b.sample_pos[4 * n + 0] * t_unit;
      double x1 = b.sample_pos[4 * n + 1], x2 = b.sample_pos[4 * n + 2], x3 = b.sample_pos[4 * n + 3];
      double kcov[4] = {b.sample_dir[4 * n + 0], b.sample_dir[4 * n + 1], b.sample_dir[4 * n + 2], b.sample_dir[4 * n + 3]};
      double j = std::numeric_limits<double>::quiet_NaN();
      if (p.image_light or p.image_emission or p.image_emission_ave) j = b.j_i[static_cast<size_t>(l) * max_steps + n];
      double alpha = std::numeric_limits<double>::quiet_NaN();
      if (p.image_light or p.image_tau or p.image_tau_int) alpha = b.alpha_i[static_cast<size_t>(l) * max_steps + n];
      double ss = j / alpha;
      double delta_tau = alpha * delta_lambda_cgs;
      double exp_neg = M::exp(-delta_tau);
      double expm1 = M::expm1(delta_tau);
      bool optically_thin = delta_tau <= delta_tau_max;
      if (p.image_light) {
        if (alpha > 0.0) {
          if (optically_thin)
            image_col[l] = exp_neg * (image_col[l] + ss * expm1);
          else
            image_col[l] = ss;
        } else
          image_col[l] += j * delta_lambda_cgs;
      }
      if (p.image_time and l == 0)
        image_col[o.image_offset_time] = std::min(image_col[o.image_offset_time], t_cgs);
      if (p.image_length and l == 0) {
        double gcov[4][4], gcon[4][4];
        CovariantGeodesicMetric(o, x1, x2, x3, gcov);
        ContravariantGeodesicMetric(o, x1, x2, x3, gcon);
        double temp_a[4] = {};
        for (int a = 1; a < 4; a++)
          for (int mu = 0; mu < 4; mu++)
            temp_a[a] += (gcon[a][mu] - gcon[0][a] * gcon[0][mu] / gcon[0][0]) * kcov[mu];
        double dl_dlambda_sq = 0.0;
        for (int a = 1; a < 4; a++)
          for (int bb = 1; bb < 4; bb++) dl_dlambda_sq += gcov[a][bb] * temp_a[a] * temp_a[bb];
        image_col[o.image_offset_length] += std::sqrt(dl_dlambda_sq) * delta_lambda * x_unit;
      }
      if (p.image_lambda or p.image_lambda_ave) integrated_lambda += delta_lambda_cgs;
      if (p.image_emission or p.image_emission_ave) integrated_emission += j * delta_lambda_cgs;
      if (p.image_tau) image_col[o.image_offset_tau + l] += delta_tau;
      const double *cell = &b.cell_values[0];
      bool cell_ok = not std::isnan(cell[0 * static_cast<size_t>(max_steps) + n]);
      if (p.image_lambda_ave and cell_ok)
        for (int a = 0; a < num_cell_values; a++)
          image_col[o.image_offset_lambda_ave + l * num_cell_values + a] +=
              cell[a * static_cast<size_t>(max_steps) + n] * delta_lambda_cgs;
      if (p.image_emission_ave and cell_ok)
        for (int a = 0; a < num_cell_values; a++)
          image_col[o.image_offset_emission_ave + l * num_cell_values + a] +=
              cell[a * static_cast<size_t>(max_steps) + n] * j * delta_lambda_cgs;
      if (p.image_tau_int and cell_ok) {
        if (optically_thin)
          for (int a = 0; a < num_cell_values; a++) {
            int index = o.image_offset_tau_int + l * num_cell_values + a;
            image_col[index] = exp_neg * (image_col[index] + cell[a * static_cast<size_t>(max_steps) + n] * expm1);
          }
        else
          for (int a = 0; a < num_cell_values; a++)
            image_col[o.image_offset_tau_int + l * num_cell_values + a] = cell[a * static_cast<size_t>(max_steps) + n];
      }
      if (p.image_crossings and l == 0) {
        bool plane_sign_new = o.cam_x[1] * x1 + o.cam_x[2] * x2 + o.cam_x[3] * x3 > 0.0;
        if (plane_sign_new != plane_sign) crossings_count++;
        plane_sign = plane_sign_new;
      }
    }
    if (p.image_lambda) image_col[o.image_offset_lambda + l] = integrated_lambda;
    if (p.image_emission) image_col[o.image_offset_emission + l] = integrated_emission;
    if (p.image_crossings and l == 0) image_col[o.image_offset_crossings] = static_cast<double>(crossings_count);
    if (p.image_lambda_ave)
      for (int a = 0; a < num_cell_values; a++)
        image_col[o.image_offset_lambda_ave + l * num_cell_values + a] /= integrated_lambda;
    if (p.image_emission_ave)
      for (int a = 0; a < num_cell_values; a++)
        image_col[o.image_offset_emission_ave + l * num_cell_values + a] /= integrated_emission;
  }
  if (p.image_light)  // :200-208
    for (int l = 0; l < nf; l++) {
      double nu_cu = o.image_frequencies[l] * o.image_frequencies[l] * o.image_frequencies[l];
      image_col[l] *= nu_cu;
    }
}

// polarized.cpp:51-949 for one ray: the coherency tensor N^{mu nu} is parallel-transported along the ray
// (half a step before and after every sample, connection and k^mu averaged with the previous sample's),
// taken into the fluid's orthonormal tetrad, turned into Stokes parameters, coupled to the plasma over the
// sample's length (rotation split from emission / absorption, or the joint analytic solution), put back, and
// finally projected on the camera's tetrad. image_col as in IntegrateUnpolarizedOne; rows 4 l + {0..3} =
// I, Q, U, V of frequency l. camera_pos / camera_dir: the pixel's initial position and momentum.
void IntegratePolarizedOne(const Oracle &o, const RayBuffers &b, int num_steps, int max_steps,
                           double momentum_factor, const double camera_pos[4], const double camera_dir[4],
                           double *image_col) {
  typedef std::complex<double> cplx;
  const bl_params &p = *o.p;
  const cplx imag_unit(0.0, 1.0);
  constexpr double delta_tau_max = 100.0;  // radiation_integrator.hpp:191
  int nf = p.image_num_frequencies;
  double x_unit = Physics::gg_msun * o.mass_msun / (Physics::c * Physics::c);
  double t_unit = x_unit / Physics::c;
  for (int q = 0; q < o.image_num_quantities; q++) image_col[q] = 0.0;
  if (num_steps <= 0) return;   // :94-96 (the image stays zero, also after the camera projection of a zero N)
  for (int l = 0; l < nf; l++) {
    double delta_lambda_old = 0.0;
    double kcon_old[4];
    double gcov[4][4], gcon[4][4], gcov_sim[4][4], gcon_sim[4][4], connection[4][4][4], connection_old[4][4][4];
    double tetrad[4][4], jacobian[4][4];
    cplx nn_con[4][4], nn_con_temp[4][4], nn_tet_cov[4][4], nn_tet_con[4][4];
    for (int mu = 0; mu < 4; mu++)
      for (int nu = 0; nu < 4; nu++) {
        nn_con[mu][nu] = 0.0;
        nn_con_temp[mu][nu] = 0.0;
      }
    double integrated_lambda = 0.0;
    double integrated_emission = 0.0;
    bool plane_sign = o.cam_x[1] * b.sample_pos[1] + o.cam_x[2] * b.sample_pos[2] + o.cam_x[3] * b.sample_pos[3] > 0.0;
    int crossings_count = 0;
    for (int n = 0; n < num_steps; n++) {
      double delta_lambda = b.sample_len[n];
      double delta_lambda_new = delta_lambda;
      if (n < num_steps - 1) delta_lambda_new = b.sample_len[n + 1];
      double delta_lambda_cgs = delta_lambda * x_unit / (o.image_frequencies[l] * momentum_factor);
      double t_cgs = b.sample_pos[4 * n + 0] * t_unit;
      double x1 = b.sample_pos[4 * n + 1], x2 = b.sample_pos[4 * n + 2], x3 = b.sample_pos[4 * n + 3];
      double kcov[4] = {b.sample_dir[4 * n + 0], b.sample_dir[4 * n + 1], b.sample_dir[4 * n + 2], b.sample_dir[4 * n + 3]};
      double uu1_sim = b.sample_ub[6 * n + 0], uu2_sim = b.sample_ub[6 * n + 1], uu3_sim = b.sample_ub[6 * n + 2];
      double bb1_sim = b.sample_ub[6 * n + 3], bb2_sim = b.sample_ub[6 * n + 4], bb3_sim = b.sample_ub[6 * n + 5];

      CovariantGeodesicMetric(o, x1, x2, x3, gcov);  // :152-166
      ContravariantGeodesicMetric(o, x1, x2, x3, gcon);
      GeodesicConnection(o, x1, x2, x3, connection);
      for (int mu = 0; mu < 4; mu++)
        for (int alpha = 0; alpha < 4; alpha++)
          for (int beta = 0; beta < 4; beta++)
            connection_old[mu][alpha][beta] = n == 0 ? connection[mu][alpha][beta]
                : 0.5 * (connection_old[mu][alpha][beta] + connection[mu][alpha][beta]);
      double kcon[4] = {};  // :169-178
      for (int mu = 0; mu < 4; mu++)
        for (int nu = 0; nu < 4; nu++) kcon[mu] += gcon[mu][nu] * kcov[nu];
      for (int mu = 0; mu < 4; mu++) kcon_old[mu] = n == 0 ? kcon[mu] : 0.5 * (kcon_old[mu] + kcon[mu]);

      double temp_a[4][4] = {};  // first half step of the transport, :181-198
      for (int mu = 0; mu < 4; mu++)
        for (int beta = 0; beta < 4; beta++)
          for (int alpha = 0; alpha < 4; alpha++) temp_a[mu][beta] += kcon_old[alpha] * connection_old[mu][alpha][beta];
      double delta_lambda_local = (delta_lambda_old + delta_lambda) / 2.0;
      for (int mu = 0; mu < 4; mu++)
        for (int nu = 0; nu < 4; nu++) {
          cplx dnn_dlambda = 0.0;
          for (int beta = 0; beta < 4; beta++)
            dnn_dlambda -= temp_a[mu][beta] * nn_con[beta][nu] + temp_a[nu][beta] * nn_con[mu][beta];
          nn_con_temp[mu][nu] += dnn_dlambda * delta_lambda_local;
        }
      for (int mu = 0; mu < 4; mu++)
        for (int nu = 0; nu < 4; nu++) nn_con[mu][nu] = nn_con_temp[mu][nu];

      CovariantSimulationMetric(o, x1, x2, x3, gcov_sim);  // :201-229
      ContravariantSimulationMetric(o, x1, x2, x3, gcon_sim);
      double uu0_sim = std::sqrt(1.0 + gcov_sim[1][1] * uu1_sim * uu1_sim
          + 2.0 * gcov_sim[1][2] * uu1_sim * uu2_sim + 2.0 * gcov_sim[1][3] * uu1_sim * uu3_sim
          + gcov_sim[2][2] * uu2_sim * uu2_sim + 2.0 * gcov_sim[2][3] * uu2_sim * uu3_sim
          + gcov_sim[3][3] * uu3_sim * uu3_sim);
      double lapse_sim = 1.0 / std::sqrt(-gcon_sim[0][0]);
      double shift1_sim = -gcon_sim[0][1] / gcon_sim[0][0];
      double shift2_sim = -gcon_sim[0][2] / gcon_sim[0][0];
      double shift3_sim = -gcon_sim[0][3] / gcon_sim[0][0];
      double ucon_sim[4];
      ucon_sim[0] = uu0_sim / lapse_sim;
      ucon_sim[1] = uu1_sim - shift1_sim * uu0_sim / lapse_sim;
      ucon_sim[2] = uu2_sim - shift2_sim * uu0_sim / lapse_sim;
      ucon_sim[3] = uu3_sim - shift3_sim * uu0_sim / lapse_sim;
      double ucov_sim[4] = {};
      for (int mu = 0; mu < 4; mu++)
        for (int nu = 0; nu < 4; nu++) ucov_sim[mu] += gcov_sim[mu][nu] * ucon_sim[nu];
      double bcon_sim[4];
      bcon_sim[0] = ucov_sim[1] * bb1_sim + ucov_sim[2] * bb2_sim + ucov_sim[3] * bb3_sim;
      bcon_sim[1] = (bb1_sim + bcon_sim[0] * ucon_sim[1]) / ucon_sim[0];
      bcon_sim[2] = (bb2_sim + bcon_sim[0] * ucon_sim[2]) / ucon_sim[0];
      bcon_sim[3] = (bb3_sim + bcon_sim[0] * ucon_sim[3]) / ucon_sim[0];

      CoordinateJacobian(o, x1, x2, x3, jacobian);  // :232-256
      double ucon[4] = {}, bcon[4] = {}, ucov[4] = {}, bcov[4] = {};
      for (int mu = 0; mu < 4; mu++)
        for (int nu = 0; nu < 4; nu++) ucon[mu] += jacobian[mu][nu] * ucon_sim[nu];
      for (int mu = 0; mu < 4; mu++)
        for (int nu = 0; nu < 4; nu++) bcon[mu] += jacobian[mu][nu] * bcon_sim[nu];
      for (int mu = 0; mu < 4; mu++)
        for (int nu = 0; nu < 4; nu++) ucov[mu] += gcov[mu][nu] * ucon[nu];
      for (int mu = 0; mu < 4; mu++)
        for (int nu = 0; nu < 4; nu++) bcov[mu] += gcov[mu][nu] * bcon[nu];
      (void)bcov;
      double upcon[4] = {};  // :259-265
      if (bb1_sim == 0.0 and bb2_sim == 0.0 and bb3_sim == 0.0)
        upcon[3] = 1.0;
      else
        for (int mu = 0; mu < 4; mu++) upcon[mu] = bcon[mu];
      Tetrad(ucon, ucov, kcon, kcov, upcon, gcov, gcon, tetrad);

      cplx temp_b[4][4] = {}, temp_c[4][4] = {}, temp_d[4][4] = {};  // N into the tetrad, :268-292
      for (int nu = 0; nu < 4; nu++)
        for (int alpha = 0; alpha < 4; alpha++)
          for (int beta = 0; beta < 4; beta++) temp_b[nu][alpha] += gcov[nu][beta] * nn_con[alpha][beta];
      for (int mu = 0; mu < 4; mu++)
        for (int nu = 0; nu < 4; nu++)
          for (int alpha = 0; alpha < 4; alpha++) temp_c[mu][nu] += gcov[mu][alpha] * temp_b[nu][alpha];
      for (int bb = 0; bb < 4; bb++)
        for (int mu = 0; mu < 4; mu++)
          for (int nu = 0; nu < 4; nu++) temp_d[bb][mu] += tetrad[bb][nu] * temp_c[mu][nu];
      for (int a = 0; a < 4; a++)
        for (int bb = 0; bb < 4; bb++) {
          nn_tet_cov[a][bb] = 0.0;
          for (int mu = 0; mu < 4; mu++) nn_tet_cov[a][bb] += tetrad[a][mu] * temp_d[bb][mu];
        }
      double ss_start[4];
      ss_start[0] = 0.5 * (nn_tet_cov[1][1] + nn_tet_cov[2][2]).real();
      ss_start[1] = 0.5 * (nn_tet_cov[1][1] - nn_tet_cov[2][2]).real();
      ss_start[2] = 0.5 * (nn_tet_cov[1][2] + nn_tet_cov[2][1]).real();
      ss_start[3] = 0.5 * (nn_tet_cov[2][1] - nn_tet_cov[1][2]).imag();

      size_t at = static_cast<size_t>(l) * max_steps + n;  // :295-315
      double j_s[4] = {b.j_i[at], b.pol[0][at], 0.0, b.pol[1][at]};
      double alpha_s[4] = {b.alpha_i[at], b.pol[2][at], 0.0, b.pol[3][at]};
      double rho_s[4] = {0.0, b.pol[4][at], 0.0, b.pol[5][at]};
      double delta_tau = alpha_s[0] * delta_lambda_cgs;
      bool optically_thin = delta_tau <= delta_tau_max;

      // alternative image quantities, :318-381
      if (p.image_time and l == 0) image_col[o.image_offset_time] = std::min(image_col[o.image_offset_time], t_cgs);
      if (p.image_length and l == 0) {
        double temp_e[4] = {};
        for (int a = 1; a < 4; a++)
          for (int mu = 0; mu < 4; mu++) temp_e[a] += (gcon[a][mu] - gcon[0][a] * gcon[0][mu] / gcon[0][0]) * kcov[mu];
        double dl_dlambda_sq = 0.0;
        for (int a = 1; a < 4; a++)
          for (int bb = 1; bb < 4; bb++) dl_dlambda_sq += gcov[a][bb] * temp_e[a] * temp_e[bb];
        image_col[o.image_offset_length] += std::sqrt(dl_dlambda_sq) * delta_lambda * x_unit;
      }
      if (p.image_lambda or p.image_lambda_ave) integrated_lambda += delta_lambda_cgs;
      if (p.image_emission or p.image_emission_ave) integrated_emission += j_s[0] * delta_lambda_cgs;
      if (p.image_tau) image_col[o.image_offset_tau + l] += delta_tau;
      const double *cell = &b.cell_values[n];
      bool have_cell = not std::isnan(cell[0]);
      if (p.image_lambda_ave and have_cell)
        for (int a = 0; a < num_cell_values; a++)
          image_col[o.image_offset_lambda_ave + l * num_cell_values + a] += cell[a * static_cast<size_t>(max_steps)] * delta_lambda_cgs;
      if (p.image_emission_ave and have_cell)
        for (int a = 0; a < num_cell_values; a++)
          image_col[o.image_offset_emission_ave + l * num_cell_values + a] +=
              cell[a * static_cast<size_t>(max_steps)] * j_s[0] * delta_lambda_cgs;
      if (p.image_tau_int and have_cell) {
        if (optically_thin) {
          double exp_neg = M::exp(-delta_tau);
          double expm1 = M::expm1(delta_tau);
          for (int a = 0; a < num_cell_values; a++) {
            int index = o.image_offset_tau_int + l * num_cell_values + a;
            image_col[index] = exp_neg * (image_col[index] + cell[a * static_cast<size_t>(max_steps)] * expm1);
          }
        } else
          for (int a = 0; a < num_cell_values; a++)
            image_col[o.image_offset_tau_int + l * num_cell_values + a] = cell[a * static_cast<size_t>(max_steps)];
      }
      if (p.image_crossings and l == 0) {
        bool plane_sign_new = o.cam_x[1] * x1 + o.cam_x[2] * x2 + o.cam_x[3] * x3 > 0.0;
        if (plane_sign_new != plane_sign) crossings_count++;
        plane_sign = plane_sign_new;
      }

      // coupling to the plasma, :384-790
      double alpha_sq = alpha_s[1] * alpha_s[1] + alpha_s[3] * alpha_s[3];
      double alpha_p = std::sqrt(alpha_sq);
      double rho_sq = rho_s[1] * rho_s[1] + rho_s[3] * rho_s[3];
      double rho_p = std::sqrt(rho_sq);
      double ss_end[4] = {};
      // emission and absorption over a length dl (I A14-A17 and its limits); shared by the split halves
      // (dl = half the step) and the unsplit case without rotation (dl = the step)
      auto absorb = [&](double dl, double dtau) {
        if (alpha_s[0] == 0.0) {
          for (int a = 0; a < 4; a++) ss_end[a] = ss_start[a] + j_s[a] * dl;
        } else if (alpha_p == 0.0) {
          if (optically_thin) {
            double exp_neg = M::exp(-dtau);
            double expm1 = M::expm1(dtau);
            for (int a = 0; a < 4; a++) ss_end[a] = exp_neg * (ss_start[a] + j_s[a] / alpha_s[0] * expm1);
          } else
            for (int a = 0; a < 4; a++) ss_end[a] = j_s[a] / alpha_s[0];
        } else if (optically_thin) {
          double exp_neg_i = M::exp(-dtau);
          double exp_neg_p = M::exp(-alpha_p * dl);
          double sinh_p = M::sinh(alpha_p * dl);
          double cosh_p = M::cosh(alpha_p * dl);
          double coshm1_p = 0.5 * (M::expm1(alpha_p * dl) + exp_neg_p - 1.0);
          double alpha_ss = alpha_s[1] * ss_start[1] + alpha_s[3] * ss_start[3];
          double alpha_j = alpha_s[1] * j_s[1] + alpha_s[3] * j_s[3];
          double alpha_i_p_factor = 1.0 / (alpha_s[0] * alpha_s[0] - alpha_sq);
          ss_end[0] = (ss_start[0] * cosh_p - alpha_ss / alpha_p * sinh_p) * exp_neg_i
              + alpha_j * alpha_i_p_factor * (-1.0 + (alpha_s[0] * sinh_p + alpha_p * cosh_p) / alpha_p * exp_neg_p)
              + alpha_s[0] * j_s[0] * alpha_i_p_factor * (1.0 - (alpha_s[0] * cosh_p + alpha_p * sinh_p) / alpha_s[0] * exp_neg_p);
          for (int a = 1; a < 4; a++) {
            double term_1 = (ss_start[a] + alpha_s[a] * alpha_ss / alpha_sq * coshm1_p
                - ss_start[0] * alpha_s[a] / alpha_p * sinh_p) * exp_neg_i;
            double term_2 = j_s[a] * (1.0 - exp_neg_i) / alpha_s[0];
            double term_3 = alpha_j * alpha_s[a] / alpha_s[0] * alpha_i_p_factor * (1.0 - (1.0
                - alpha_s[0] * alpha_s[0] / alpha_sq - alpha_s[0] / alpha_sq * (alpha_s[0] * cosh_p + alpha_p * sinh_p)) * exp_neg_i);
            double term_4 = j_s[0] * alpha_s[a] / alpha_p * alpha_i_p_factor * (-alpha_p
                + (alpha_p * cosh_p + alpha_s[0] * sinh_p) * exp_neg_i);
            ss_end[a] = term_1 + term_2 + term_3 + term_4;
          }
        } else {
          double alpha_j = alpha_s[1] * j_s[1] + alpha_s[3] * j_s[3];
          ss_end[0] = (alpha_s[0] * j_s[0] - alpha_j) / (alpha_s[0] * alpha_s[0] - alpha_sq);
          for (int a = 1; a < 4; a++) ss_end[a] = (j_s[a] - alpha_s[a] * ss_end[0]) / alpha_s[0];
        }
      };
      // Faraday rotation and conversion over the whole step without absorption (I A2-A5)
      auto rotate = [&]() {
        double cos_rho = M::cos(rho_p * delta_lambda_cgs);
        double sin_rho = M::sin(rho_p * delta_lambda_cgs);
        double sin_sq_rho = M::sin(rho_p * delta_lambda_cgs / 2.0);
        sin_sq_rho = sin_sq_rho * sin_sq_rho;
        double rho_ss = rho_s[1] * ss_start[1] + rho_s[3] * ss_start[3];
        ss_end[0] = ss_start[0];
        ss_end[1] = ss_start[1] * cos_rho + 2.0 * rho_s[1] * rho_ss / rho_sq * sin_sq_rho - rho_s[3] * ss_start[2] / rho_p * sin_rho;
        ss_end[2] = ss_start[2] * cos_rho + (rho_s[3] * ss_start[1] - rho_s[1] * ss_start[3]) / rho_p * sin_rho;
        ss_end[3] = ss_start[3] * cos_rho + 2.0 * rho_s[3] * rho_ss / rho_sq * sin_sq_rho + rho_s[1] * ss_start[2] / rho_p * sin_rho;
      };
      auto limit_polarization = [&]() {
        double ss_pol = ss_end[1] * ss_end[1] + ss_end[2] * ss_end[2] + ss_end[3] * ss_end[3];
        if (ss_pol > ss_end[0] * ss_end[0]) {
          double factor = std::sqrt(ss_end[0] * ss_end[0] / ss_pol);
          ss_end[1] *= factor;
          ss_end[2] *= factor;
          ss_end[3] *= factor;
        }
      };
      if (p.image_rotation_split) {  // :388-568
        absorb(delta_lambda_cgs / 2.0, delta_tau / 2.0);
        ss_end[0] = std::max(ss_end[0], 0.0);
        limit_polarization();
        for (int a = 0; a < 4; a++) ss_start[a] = ss_end[a];
        if (rho_p != 0.0) rotate();
        limit_polarization();
        for (int a = 0; a < 4; a++) ss_start[a] = ss_end[a];
        absorb(delta_lambda_cgs / 2.0, delta_tau / 2.0);
      } else if (alpha_s[0] == 0.0 and rho_p == 0.0) {  // :571-577
        for (int a = 0; a < 4; a++) ss_end[a] = ss_start[a] + j_s[a] * delta_lambda_cgs;
      } else if (alpha_p == 0.0 and rho_p == 0.0) {
        absorb(delta_lambda_cgs, delta_tau);
      } else if (alpha_s[0] == 0.0) {  // :598-615
        rotate();
        for (int a = 0; a < 4; a++) ss_end[a] += j_s[a] * delta_lambda_cgs;
      } else if (rho_p == 0.0) {
        absorb(delta_lambda_cgs, delta_tau);
      } else {  // absorption and rotation together (L 10, I 24), :657-778
        double alpha_rho = alpha_s[1] * rho_s[1] + alpha_s[3] * rho_s[3];
        double alpha_sq_rho_sq = alpha_sq - rho_sq;
        double lambda_a = std::sqrt(alpha_sq_rho_sq * alpha_sq_rho_sq / 4.0 + alpha_rho * alpha_rho);
        double lambda_b = alpha_sq_rho_sq / 2.0;
        double lambda_1 = std::sqrt(lambda_a + lambda_b);
        double lambda_2 = std::sqrt(lambda_a - lambda_b);
        double coefficient_theta = lambda_1 * lambda_1 + lambda_2 * lambda_2;
        double sg = alpha_rho >= 0.0 ? 1.0 : -1.0;
        double mm_1[4][4] = {}, mm_2[4][4] = {}, mm_3[4][4] = {}, mm_4[4][4] = {};
        for (int a = 0; a < 4; a++) mm_1[a][a] = 1.0;
        // as written in the reference: [1][2] is assigned twice and [1][3], [0][2], [2][3] stay zero
        mm_2[0][1] = lambda_2 * alpha_s[1] - sg * lambda_1 * rho_s[1];
        mm_2[0][3] = lambda_2 * alpha_s[3] - sg * lambda_1 * rho_s[3];
        mm_2[1][2] = sg * lambda_1 * alpha_s[3] + lambda_2 * rho_s[3];
        mm_2[1][2] = sg * lambda_1 * alpha_s[1] + lambda_2 * rho_s[1];
        mm_2[1][0] = mm_2[0][1];
        mm_2[2][0] = mm_2[0][2];
        mm_2[3][0] = mm_2[0][3];
        mm_2[2][1] = -mm_2[1][2];
        mm_2[3][1] = -mm_2[1][3];
        mm_2[3][2] = -mm_2[2][3];
        for (int a = 0; a < 4; a++)
          for (int bb = 0; bb < 4; bb++) mm_2[a][bb] *= 1.0 / coefficient_theta;
        mm_3[0][1] = lambda_1 * alpha_s[1] + sg * lambda_2 * rho_s[1];
        mm_3[0][3] = lambda_1 * alpha_s[3] + sg * lambda_2 * rho_s[3];
        mm_3[1][2] = -(sg * lambda_2 * alpha_s[3] - lambda_1 * rho_s[3]);
        mm_3[1][2] = -(sg * lambda_2 * alpha_s[1] - lambda_1 * rho_s[1]);
        mm_3[1][0] = mm_3[0][1];
        mm_3[2][0] = mm_3[0][2];
        mm_3[3][0] = mm_3[0][3];
        mm_3[2][1] = -mm_3[1][2];
        mm_3[3][1] = -mm_3[1][3];
        mm_3[3][2] = -mm_3[2][3];
        for (int a = 0; a < 4; a++)
          for (int bb = 0; bb < 4; bb++) mm_3[a][bb] *= 1.0 / coefficient_theta;
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
          for (int bb = 0; bb < 4; bb++) mm_4[a][bb] *= 2.0 / coefficient_theta;
        double exp_v = 0.0, sin_v = 0.0, cos_v = 0.0, sinh_v = 0.0, cosh_v = 0.0;
        double oo[4][4] = {}, pp[4][4] = {};
        if (optically_thin) {
          exp_v = M::exp(-delta_tau);
          sin_v = M::sin(lambda_2 * delta_lambda_cgs);
          cos_v = M::cos(lambda_2 * delta_lambda_cgs);
          sinh_v = M::sinh(lambda_1 * delta_lambda_cgs);
          cosh_v = M::cosh(lambda_1 * delta_lambda_cgs);
          for (int a = 0; a < 4; a++)
            for (int bb = 0; bb < 4; bb++)
              oo[a][bb] = exp_v * (0.5 * (mm_1[a][bb] + mm_4[a][bb]) * cosh_v + 0.5 * (mm_1[a][bb] - mm_4[a][bb]) * cos_v
                  - mm_2[a][bb] * sin_v - mm_3[a][bb] * sinh_v);
        }
        double f_1 = 1.0 / (alpha_s[0] * alpha_s[0] - lambda_1 * lambda_1);
        double f_2 = 1.0 / (alpha_s[0] * alpha_s[0] + lambda_2 * lambda_2);
        for (int a = 0; a < 4; a++)
          for (int bb = 0; bb < 4; bb++) {
            double cosh_term = -lambda_1 * f_1 * mm_3[a][bb] + 0.5 * alpha_s[0] * f_1 * (mm_1[a][bb] + mm_4[a][bb]);
            double cos_term = -lambda_2 * f_2 * mm_2[a][bb] + 0.5 * alpha_s[0] * f_2 * (mm_1[a][bb] - mm_4[a][bb]);
            pp[a][bb] = cosh_term + cos_term;
            if (optically_thin) {
              double sin_term = -alpha_s[0] * f_2 * mm_2[a][bb] - 0.5 * lambda_2 * f_2 * (mm_1[a][bb] - mm_4[a][bb]);
              double sinh_term = -alpha_s[0] * f_1 * mm_3[a][bb] + 0.5 * lambda_1 * f_1 * (mm_1[a][bb] + mm_4[a][bb]);
              pp[a][bb] -= exp_v * (cosh_term * cosh_v + cos_term * cos_v + sin_term * sin_v + sinh_term * sinh_v);
            }
          }
        if (optically_thin) {
          for (int a = 0; a < 4; a++)
            for (int bb = 0; bb < 4; bb++) ss_end[a] += pp[a][bb] * j_s[bb] + oo[a][bb] * ss_start[bb];
        } else
          for (int a = 0; a < 4; a++)
            for (int bb = 0; bb < 4; bb++) ss_end[a] += pp[a][bb] * j_s[bb];
      }
      ss_end[0] = std::max(ss_end[0], 0.0);  // :781-790
      limit_polarization();

      for (int mu = 0; mu < 4; mu++)  // back to coordinates, :793-813
        for (int nu = 0; nu < 4; nu++) nn_tet_con[mu][nu] = 0.0;
      nn_tet_con[1][1] = ss_end[0] + ss_end[1];
      nn_tet_con[2][2] = ss_end[0] - ss_end[1];
      nn_tet_con[1][2] = ss_end[2] - imag_unit * ss_end[3];
      nn_tet_con[2][1] = ss_end[2] + imag_unit * ss_end[3];
      cplx temp_f[4][4] = {};
      for (int nu = 0; nu < 4; nu++)
        for (int a = 0; a < 4; a++)
          for (int bb = 0; bb < 4; bb++) temp_f[nu][a] += tetrad[bb][nu] * nn_tet_con[a][bb];
      for (int mu = 0; mu < 4; mu++)
        for (int nu = 0; nu < 4; nu++) {
          nn_con[mu][nu] = 0.0;
          for (int a = 0; a < 4; a++) nn_con[mu][nu] += tetrad[a][mu] * temp_f[nu][a];
        }

      for (int mu = 0; mu < 4; mu++)  // second half step, :816-833
        for (int nu = 0; nu < 4; nu++) nn_con_temp[mu][nu] = nn_con[mu][nu];
      double temp_g[4][4] = {};
      for (int mu = 0; mu < 4; mu++)
        for (int beta = 0; beta < 4; beta++)
          for (int alpha = 0; alpha < 4; alpha++) temp_g[mu][beta] += kcon[alpha] * connection[mu][alpha][beta];
      delta_lambda_local = (delta_lambda + delta_lambda_new) / 4.0;
      for (int mu = 0; mu < 4; mu++)
        for (int nu = 0; nu < 4; nu++) {
          cplx dnn_dlambda = 0.0;
          for (int beta = 0; beta < 4; beta++)
            dnn_dlambda -= temp_g[mu][beta] * nn_con_temp[beta][nu] + temp_g[nu][beta] * nn_con_temp[mu][beta];
          nn_con[mu][nu] += dnn_dlambda * delta_lambda_local;
        }
      delta_lambda_old = delta_lambda;  // :836-843
      for (int mu = 0; mu < 4; mu++) kcon_old[mu] = kcon[mu];
      for (int mu = 0; mu < 4; mu++)
        for (int alpha = 0; alpha < 4; alpha++)
          for (int beta = 0; beta < 4; beta++) connection_old[mu][alpha][beta] = connection[mu][alpha][beta];
    }

    if (p.image_lambda) image_col[o.image_offset_lambda + l] = integrated_lambda;  // :852-872
    if (p.image_emission) image_col[o.image_offset_emission + l] = integrated_emission;
    if (p.image_crossings and l == 0) image_col[o.image_offset_crossings] = static_cast<double>(crossings_count);
    if (p.image_lambda_ave)
      for (int a = 0; a < num_cell_values; a++) image_col[o.image_offset_lambda_ave + l * num_cell_values + a] /= integrated_lambda;
    if (p.image_emission_ave)
      for (int a = 0; a < num_cell_values; a++) image_col[o.image_offset_emission_ave + l * num_cell_values + a] /= integrated_emission;

    // projection on the camera's tetrad, :875-949
    double x = camera_pos[1], y = camera_pos[2], z = camera_pos[3];
    double kcov[4] = {camera_dir[0], camera_dir[1], camera_dir[2], camera_dir[3]};
    CovariantGeodesicMetric(o, x, y, z, gcov);
    ContravariantGeodesicMetric(o, x, y, z, gcon);
    double kcon[4] = {};
    for (int mu = 0; mu < 4; mu++)
      for (int nu = 0; nu < 4; nu++) kcon[mu] += gcon[mu][nu] * kcov[nu];
    double up_con[4];
    up_con[0] = o.u_con[0] * o.vert_con_c[0]
        - (o.u_cov[1] * o.vert_con_c[1] + o.u_cov[2] * o.vert_con_c[2] + o.u_cov[3] * o.vert_con_c[3]) / o.u_cov[0];
    up_con[1] = o.vert_con_c[1] + o.u_con[1] * o.vert_con_c[0];
    up_con[2] = o.vert_con_c[2] + o.u_con[2] * o.vert_con_c[0];
    up_con[3] = o.vert_con_c[3] + o.u_con[3] * o.vert_con_c[0];
    Tetrad(o.u_con, o.u_cov, kcon, kcov, up_con, gcov, gcon, tetrad);
    cplx temp_a[4][4] = {}, temp_b[4][4] = {}, temp_c[4][4] = {};
    for (int nu = 0; nu < 4; nu++)
      for (int alpha = 0; alpha < 4; alpha++)
        for (int beta = 0; beta < 4; beta++) temp_a[nu][alpha] += gcov[nu][beta] * nn_con[alpha][beta];
    for (int mu = 0; mu < 4; mu++)
      for (int nu = 0; nu < 4; nu++)
        for (int alpha = 0; alpha < 4; alpha++) temp_b[mu][nu] += gcov[mu][alpha] * temp_a[nu][alpha];
    for (int bb = 0; bb < 4; bb++)
      for (int mu = 0; mu < 4; mu++)
        for (int nu = 0; nu < 4; nu++) temp_c[bb][mu] += tetrad[bb][nu] * temp_b[mu][nu];
    for (int a = 0; a < 4; a++)
      for (int bb = 0; bb < 4; bb++) {
        nn_tet_cov[a][bb] = 0.0;
        for (int mu = 0; mu < 4; mu++) nn_tet_cov[a][bb] += tetrad[a][mu] * temp_c[bb][mu];
      }
    double nu_cu = o.image_frequencies[l] * o.image_frequencies[l] * o.image_frequencies[l];
    image_col[l * 4 + 0] = 0.5 * (nn_tet_cov[1][1] + nn_tet_cov[2][2]).real() * nu_cu;
    image_col[l * 4 + 1] = 0.5 * (nn_tet_cov[1][1] - nn_tet_cov[2][2]).real() * nu_cu;
    image_col[l * 4 + 2] = 0.5 * (nn_tet_cov[1][2] + nn_tet_cov[2][1]).real() * nu_cu;
    image_col[l * 4 + 3] = 0.5 * (nn_tet_cov[2][1] - nn_tet_cov[1][2]).imag() * nu_cu;
  }
}


// rendering.cpp:25-179 for one pixel. render_col[3 * n_i + c] = render(n_i, c, m).
void RenderOne(const Oracle &o, const RayBuffers &b, int num_steps, int max_steps, double *render_col) {
  const bl_params &p = *o.p;
  constexpr double delta_tau_max = 100.0;
  bool fill_present = false;
  for (int n_i = 0; n_i < o.render_num_images; n_i++)
    for (int n_f = 0; n_f < p.render_num_features[n_i]; n_f++)
      if (p.render_type[n_i][n_f] == BL_RENDER_FILL) fill_present = true;
  double x_unit = Physics::gg_msun * o.mass_msun / (Physics::c * Physics::c);
  for (int q = 0; q < 3 * o.render_num_images; q++) render_col[q] = 0.0;
  double previous_values[num_cell_values], current_values[num_cell_values];
  for (int n_v = 0; n_v < num_cell_values; n_v++) previous_values[n_v] = std::numeric_limits<double>::quiet_NaN();
  for (int n = 0; n < num_steps; n++) {
    double delta_lambda = b.sample_len[n];
    double x1 = b.sample_pos[4 * n + 1], x2 = b.sample_pos[4 * n + 2], x3 = b.sample_pos[4 * n + 3];
    double kcov[4] = {b.sample_dir[4 * n + 0], b.sample_dir[4 * n + 1], b.sample_dir[4 * n + 2], b.sample_dir[4 * n + 3]};
    for (int n_v = 0; n_v < num_cell_values; n_v++) current_values[n_v] = b.cell_values[n_v * static_cast<size_t>(max_steps) + n];
    double delta_length = 0.0;
    if (fill_present) {
      double gcov[4][4], gcon[4][4];
      CovariantGeodesicMetric(o, x1, x2, x3, gcov);
      ContravariantGeodesicMetric(o, x1, x2, x3, gcon);
      double temp_a[4] = {};
      for (int a = 1; a < 4; a++)
        for (int mu = 0; mu < 4; mu++) temp_a[a] += (gcon[a][mu] - gcon[0][a] * gcon[0][mu] / gcon[0][0]) * kcov[mu];
      double dl_dlambda_sq = 0.0;
      for (int a = 1; a < 4; a++)
        for (int bb = 1; bb < 4; bb++) dl_dlambda_sq += gcov[a][bb] * temp_a[a] * temp_a[bb];
      delta_length = std::sqrt(dl_dlambda_sq) * delta_lambda * x_unit;
    }
    for (int n_i = 0; n_i < o.render_num_images; n_i++) {
      double *rgb = render_col + 3 * n_i;
      for (int n_f = 0; n_f < p.render_num_features[n_i]; n_f++) {
        int n_v = p.render_quantity[n_i][n_f];
        int type = p.render_type[n_i][n_f];
        double previous_value = previous_values[n_v];
        double current_value = current_values[n_v];
        const double xyz[3] = {p.render_x[n_i][n_f], p.render_y[n_i][n_f], p.render_z[n_i][n_f]};
        if (type == BL_RENDER_FILL and current_value >= p.render_min[n_i][n_f] and current_value <= p.render_max[n_i][n_f]) {
          double delta_tau = delta_length / p.render_tau_scale[n_i][n_f];
          if (delta_tau <= delta_tau_max) {
            double exp_neg = M::exp(-delta_tau);
            double expm1 = M::expm1(delta_tau);
            for (int c = 0; c < 3; c++) rgb[c] = exp_neg * (rgb[c] + xyz[c] * expm1);
          } else {
            for (int c = 0; c < 3; c++) rgb[c] = xyz[c];
          }
        }
        bool threshold_crossed = false;
        bool rise_search = type == BL_RENDER_THRESH or type == BL_RENDER_RISE;
        if (rise_search and previous_value < p.render_thresh[n_i][n_f] and current_value >= p.render_thresh[n_i][n_f])
          threshold_crossed = true;
        bool fall_search = type == BL_RENDER_THRESH or type == BL_RENDER_FALL;
        if (fall_search and previous_value > p.render_thresh[n_i][n_f] and current_value <= p.render_thresh[n_i][n_f])
          threshold_crossed = true;
        if (threshold_crossed) {
          double opacity = p.render_opacity[n_i][n_f];
          for (int c = 0; c < 3; c++) rgb[c] = (1.0 - opacity) * rgb[c] + opacity * xyz[c];
        }
      }
    }
    for (int n_v = 0; n_v < num_cell_values; n_v++) previous_values[n_v] = current_values[n_v];
  }
}

// radiation_integrator.cpp:436-520
void ImageOffsets(Oracle &o) {
  const bl_params &p = *o.p;
  int nf = p.image_num_frequencies;
  int n = 0;
  int *offs[9] = {&o.image_offset_time, &o.image_offset_length, &o.image_offset_lambda,
                  &o.image_offset_emission, &o.image_offset_tau, &o.image_offset_lambda_ave,
                  &o.image_offset_emission_ave, &o.image_offset_tau_int, &o.image_offset_crossings};
  auto set_from = [&](int first) { for (int q = first; q < 9; q++) *offs[q] = n; };
  set_from(0);
  if (p.image_light) { n += nf * (p.model_type == BL_MODEL_SIMULATION and o.image_polarization ? 4 : 1); set_from(0); }
  if (p.image_time) { n++; set_from(1); }
  if (p.image_length) { n++; set_from(2); }
  if (p.image_lambda) { n += nf; set_from(3); }
  if (p.image_emission) { n += nf; set_from(4); }
  if (p.image_tau) { n += nf; set_from(5); }
  bool sim = p.model_type == BL_MODEL_SIMULATION;
  if (sim and p.image_lambda_ave) { n += nf * num_cell_values; set_from(6); }
  if (sim and p.image_emission_ave) { n += nf * num_cell_values; set_from(7); }
  if (sim and p.image_tau_int) { n += nf * num_cell_values; set_from(8); }
  if (p.image_crossings) n++;
  o.image_num_quantities = n;
}

int Fail(char *err, size_t err_len, const char *message, int code) {
  if (err != nullptr && err_len > 0) std::snprintf(err, err_len, "Error: %s\n", message);
  return code;
}

int Setup(Oracle &o, const bl_params *p, const bl_grid_desc *g, char *err, size_t err_len) {
  o.p = p;
  o.g = g;
  o.bh_m = 1.0;
  if (p->model_type == BL_MODEL_SIMULATION) {
    o.bh_a = p->simulation_a;
    o.mass_msun = p->simulation_m_msun;
  } else {
    o.bh_a = p->formula_spin;
    o.mass_msun = p->formula_mass * Physics::c * Physics::c / Physics::gg_msun;
  }
  o.ray_flat = p->ray_flat != 0;
  o.r_horizon = o.bh_m + std::sqrt(o.bh_m * o.bh_m - o.bh_a * o.bh_a);  // geodesic_integrator.cpp:117-123
  if (p->ray_terminate == BL_TERMINATE_PHOTON)
    o.r_terminate = 2.0 * o.bh_m * (1.0 + M::cos(2.0 / 3.0 * M::acos(-std::abs(o.bh_a) / o.bh_m)));
  else if (p->ray_terminate == BL_TERMINATE_MULTIPLICATIVE)
    o.r_terminate = o.r_horizon * p->ray_factor;
  else
    o.r_terminate = o.r_horizon + p->ray_factor;
  o.image_polarization = p->model_type == BL_MODEL_SIMULATION and p->image_light and p->image_polarization;
  if (p->model_type == BL_MODEL_SIMULATION) {
    if (g == nullptr) return Fail(err, err_len, "oracle: simulation mode needs a grid", BL_E_ARG);
    if (p->simulation_block_interp and p->simulation_interp and (g->levels == nullptr or g->locations == nullptr or (g->n_3_root <= 0 and p->simulation_coord == BL_COORD_SKS)))
      return Fail(err, err_len, "oracle: inter-block interpolation needs the MeshBlock table (levels, locations, n_3_root)", BL_E_ARG);
    if (p->slow_light_on and (o.slow_n < 2 or o.slow_n != p->slow_chunk_size or o.slow_grids == nullptr or o.slow_times == nullptr))
      return Fail(err, err_len, "oracle: slow light needs slow_chunk_size time slices in blo_extra", BL_E_ARG);
    if (p->simulation_coord == BL_COORD_FMKS and (g->sks_map == nullptr or g->n_blocks != 1))
      return Fail(err, err_len, "oracle: fmks needs the reader's sks_map and one block", BL_E_UNSUPPORTED);
    if (p->plasma_kappa_frac != 0.0 and not o.image_polarization and not o.define_kappa_aa_high_i)
      return Fail(err, err_len, "oracle: in unpolarized runs with kappa-distribution electrons the reference reads the uninitialised kappa_aa_high_i (blo_extra::define_kappa_aa_high_i gives it its polarized definition)", BL_E_UNSUPPORTED);
    o.plasma_thermal_frac = 1.0 - (p->plasma_power_frac + p->plasma_kappa_frac);
    o.render_num_images = p->render_num_images;   // radiation_integrator.cpp:136-139
    if (p->plasma_power_frac != 0.0) {  // simulation_coefficients.cpp:54-66 (unpolarized part)
      double plasma_p = p->plasma_p;
      double var_a = M::pow(3.0, plasma_p / 2.0) * (plasma_p - 1.0);
      double var_b = 2.0 * (plasma_p + 1.0);
      double var_c = M::pow(p->plasma_gamma_min, 1.0 - plasma_p) - M::pow(p->plasma_gamma_max, 1.0 - plasma_p);
      double var_d = std::tgamma((3.0 * plasma_p - 1.0) / 12.0);
      double var_e = std::tgamma((3.0 * plasma_p + 19.0) / 12.0);
      double var_f = M::pow(3.0, (plasma_p + 1.0) / 2.0) * (plasma_p - 1.0) / 4.0;
      double var_g = std::tgamma((3.0 * plasma_p + 2.0) / 12.0);
      double var_h = std::tgamma((3.0 * plasma_p + 22.0) / 12.0);
      o.power_jj = var_a / var_b / var_c * var_d * var_e;
      o.power_aa = var_f / var_c * var_g * var_h;
      if (o.image_polarization) {
        double var_i = 2.0 * (plasma_p + 2.0) / (plasma_p + 1.0);
        double var_j = M::pow(p->plasma_gamma_min, -(plasma_p + 1.0));
        double var_k = M::log(p->plasma_gamma_min);
        o.power_pol[0] = -(plasma_p + 1.0) / (plasma_p + 7.0 / 3.0);
        o.power_pol[1] = 0.684 * M::pow(plasma_p, 0.49);
        o.power_pol[2] = -M::pow(0.034 * plasma_p - 0.0344, 0.086);
        o.power_pol[3] = M::pow(0.71 * plasma_p + 0.0352, 0.394);
        o.power_pol[4] = (plasma_p - 1.0) / var_c;
        o.power_pol[5] = -M::pow(p->plasma_gamma_min, 2.0 - plasma_p) / (plasma_p / 2.0 - 1.0);
        o.power_pol[6] = var_i * var_j * var_k;
      }
    }
    if (p->plasma_kappa_frac != 0.0) {  // simulation_coefficients.cpp:82-193 (a polarized run: see above)
      auto &kk = o.kappa;
      double plasma_kappa = p->plasma_kappa, plasma_w = p->plasma_w;
      double var_a = 4.0 * Math::pi * std::tgamma(plasma_kappa - 4.0 / 3.0);
      double var_b = M::pow(3.0, 7.0 / 3.0) * std::tgamma(plasma_kappa - 2.0);
      double var_c = M::pow(3.0, (plasma_kappa - 1.0) / 2.0);
      double var_d = (plasma_kappa - 2.0) * (plasma_kappa - 1.0) / 4.0;
      double var_e = std::tgamma(plasma_kappa / 4.0 - 1.0 / 3.0);
      double var_f = std::tgamma(plasma_kappa / 4.0 + 4.0 / 3.0);
      double var_g = M::pow(3.0, 1.0 / 6.0) * 10.0 / 41.0;
      double var_h = plasma_w * plasma_kappa;
      double var_i = 2.0 * Math::pi * M::pow(var_h, plasma_kappa - 10.0 / 3.0);
      double var_j = (plasma_kappa - 2.0) * (plasma_kappa - 1.0) * plasma_kappa;
      double var_k = 3.0 * plasma_kappa - 1.0;
      double var_l = std::tgamma(5.0 / 3.0);
      double var_m = Hypergeometric(plasma_kappa - 1.0 / 3.0, plasma_kappa + 1.0, plasma_kappa + 2.0 / 3.0, -var_h);
      double var_n = M::pow(Math::pi, 1.5) / 3.0;
      double var_o = var_j / (var_h * var_h * var_h);
      double var_p = 2.0 * std::tgamma(2.0 + plasma_kappa / 2.0) / (2.0 + plasma_kappa) - 1.0;
      kk.jj_low = var_a / var_b;
      kk.jj_high = var_c * var_d * var_e * var_f;
      kk.jj_x_i = 3.0 * M::pow(plasma_kappa, -1.5);
      kk.aa_low = var_g * var_i * var_j / var_k * var_l * var_m;
      kk.aa_high = var_n * var_o * var_p;
      kk.aa_x_i = M::pow(-1.75 + 1.6 * plasma_kappa, -0.86);
      double var_q = 14.3 * M::pow(plasma_w, -0.928);
      double var_r = 169.0 * M::pow(plasma_kappa, -8.0) + 0.0052 * plasma_kappa - 0.0526 + 47.0 / (200.0 * plasma_kappa);
      kk.jj_low_q = 0.5;
      kk.jj_low_v = 0.5625 * M::pow(plasma_kappa, -0.528) / plasma_w;
      kk.jj_high_q = 0.64 + 0.02 * plasma_kappa;
      kk.jj_high_v = 0.765625 * M::pow(plasma_kappa, -0.44) / plasma_w;
      kk.jj_x_q = 3.7 * M::pow(plasma_kappa, -1.6);
      kk.jj_x_v = kk.jj_x_i;
      kk.aa_low_q = 25.0 / 48.0;
      kk.aa_low_v = 77.0 / (100.0 * plasma_w) * M::pow(plasma_kappa, -0.7);
      kk.aa_high_i = M::pow(3.0 / plasma_kappa, 4.75) + 0.6;
      kk.aa_high_q = 441.0 * M::pow(plasma_kappa, -5.76) + 0.55;
      kk.aa_high_v = var_q * var_r;
      kk.aa_x_q = 1.4 * M::pow(plasma_kappa, -1.15);
      kk.aa_x_v = 1.22 * M::pow(plasma_kappa, -1.136) + 0.007;
      kk.rho_v = CylBesselK(0, 1.0 / plasma_w) / CylBesselK(2, 1.0 / plasma_w);
      // the three brackets of kappa (:128-192): each interpolates between the fits for its two ends
      auto rho_q_a = [&](double c_w, double c_0, double c_1) {
        return c_w * plasma_w + std::sqrt(plasma_w) * (c_0 + c_1 * M::exp(-5.0 * plasma_w));
      };
      auto rho_v_a = [&](int which) {
        switch (which) {
          case 0: return (plasma_w * plasma_w + 2.0 * plasma_w + 1.0) / (3.125 * plasma_w * plasma_w + 4.0 * plasma_w + 1.0);
          case 1: return (plasma_w * plasma_w + 54.0 * plasma_w + 50.0) / (30.0 / 11.0 * plasma_w * plasma_w + 134.0 * plasma_w + 50.0);
          case 2: return (plasma_w * plasma_w + 43.0 * plasma_w + 38.0) / (7.0 / 3.0 * plasma_w * plasma_w + 92.5 * plasma_w + 38.0);
          default: return (plasma_w + 13.0 / 14.0) / (2.0 * plasma_w + 13.0 / 14.0);
        }
      };
      // fits at kappa = 3.5, 4, 4.5, 5 (rho_q: a, b, c, d, e; rho_v: a, b)
      const double fit_q[4][5] = {
          {rho_q_a(17.0, -3.0, 7.0), -1.0 / 30.0, 0.1, -1.5, 0.471},
          {rho_q_a(46.0 / 3.0, -5.0 / 3.0, 17.0 / 3.0), -1.0 / 18.0, 1.0 / 6.0, -1.75, 0.5},
          {rho_q_a(14.0, -1.625, 4.5), -1.0 / 12.0, 0.25, -2.0, 0.525},
          {rho_q_a(12.5, -1.0, 5.0), -0.125, 0.375, -2.25, 0.541}};
      const double fit_v_b[4] = {0.447, 0.391, 0.348, 0.313};
      int lo = plasma_kappa < 4.0 ? 0 : (plasma_kappa < 4.5 ? 1 : 2);
      kk.rho_frac = (plasma_kappa - (3.5 + 0.5 * lo)) / ((4.0 + 0.5 * lo) - (3.5 + 0.5 * lo));
      kk.rho_q_low_a = fit_q[lo][0]; kk.rho_q_low_b = fit_q[lo][1]; kk.rho_q_low_c = fit_q[lo][2];
      kk.rho_q_low_d = fit_q[lo][3]; kk.rho_q_low_e = fit_q[lo][4];
      kk.rho_q_high_a = fit_q[lo + 1][0]; kk.rho_q_high_b = fit_q[lo + 1][1]; kk.rho_q_high_c = fit_q[lo + 1][2];
      kk.rho_q_high_d = fit_q[lo + 1][3]; kk.rho_q_high_e = fit_q[lo + 1][4];
      kk.rho_v_low_a = rho_v_a(lo); kk.rho_v_low_b = fit_v_b[lo];
      kk.rho_v_high_a = rho_v_a(lo + 1); kk.rho_v_high_b = fit_v_b[lo + 1];
    }
  } else
    o.plasma_thermal_frac = 0.0;
  ImageOffsets(o);
  InitializeCamera(o);
  (void)need;
  return BL_OK;
}

}  // namespace

extern "C" {

int blo_image_num_quantities(const bl_params *p) {
  Oracle o{};
  o.p = p;
  o.image_polarization = p->model_type == BL_MODEL_SIMULATION and p->image_light and p->image_polarization;
  ImageOffsets(o);
  return o.image_num_quantities;
}

// Length of the frequency list blo_render() writes to its `frequencies` argument
int blo_image_num_frequencies(const bl_params *p) { return p->image_num_frequencies; }

const char *blo_build_info(void) {
#ifdef BLO_LIBM
  return "oracle;libm";
#else
  return "oracle;blmath";
#endif
}

int blo_render(const bl_params *p, const bl_grid_desc *g, const bl_render_desc *d,
               bl_camera_frame *frame, double *frequencies, blo_extra *extra, char *err,
               size_t err_len) {
  if (p == nullptr || d == nullptr) return BL_E_ARG;
  if (d->outputs_on_device) return Fail(err, err_len, "oracle works on host memory only", BL_E_ARG);
  Oracle o{};
  o.define_kappa_aa_high_i = extra != nullptr and extra->define_kappa_aa_high_i != 0;
  if (extra != nullptr and p->model_type == BL_MODEL_SIMULATION and p->slow_light_on) {
    o.slow_n = extra->slow_n;
    o.slow_grids = extra->slow_grids;
    o.slow_times = extra->slow_times;
    o.snapshot_time = extra->slow_snapshot_time;
  }
  int rc = Setup(o, p, g, err, err_len);
  if (rc != BL_OK) return rc;
  if (frame != nullptr) {
    for (int mu = 0; mu < 4; mu++) {
      frame->cam_x[mu] = o.cam_x[mu];
      frame->u_con[mu] = o.u_con[mu];
      frame->u_cov[mu] = o.u_cov[mu];
      frame->norm_con[mu] = o.norm_con[mu];
      frame->norm_con_c[mu] = o.norm_con_c[mu];
      frame->hor_con_c[mu] = o.hor_con_c[mu];
      frame->vert_con_c[mu] = o.vert_con_c[mu];
    }
    frame->bh_m = o.bh_m;
    frame->bh_a = o.bh_a;
    frame->r_horizon = o.r_horizon;
    frame->r_terminate = o.r_terminate;
    frame->mass_msun = o.mass_msun;
  }
  if (frequencies != nullptr)
    for (int l = 0; l < p->image_num_frequencies; l++) frequencies[l] = o.image_frequencies[l];

  int64_t n_rays = d->n_rays;
  int max_steps = p->ray_max_steps;
  int nf = p->image_num_frequencies;
  int n_q = o.image_num_quantities;
  bool simulation = p->model_type == BL_MODEL_SIMULATION;
  int num_threads = (extra != nullptr && extra->num_threads > 0) ? extra->num_threads : omp_get_max_threads();
  int64_t total_samples = 0, total_gathers = 0, total_flagged = 0;
  int max_sample_num = 0;
  if (extra != nullptr) extra->dump_num = 0;
  double time_start = omp_get_wtime();
  int64_t count_0 = 0, count_1 = 0, count_2 = 0, count_3 = 0;   // simulation_sampling.cpp:160-168
  int64_t count_undefined = 0, count_failed = 0;                // inter-block interpolation: SampleOne status 4, 5
  double val_0 = 0.0, val_1 = 0.0, val_2 = 0.0, val_3 = 0.0;

  #pragma omp parallel num_threads(num_threads) reduction(+: total_samples, total_gathers, total_flagged, count_0, count_1, count_2, count_3, count_undefined, count_failed) reduction(max: max_sample_num, val_0, val_1, val_2, val_3)
  {
    RayBuffers b(max_steps, nf);
    if (o.image_polarization) b.EnablePolarization(max_steps, nf);
    BlockState block_state;   // per thread, kept across rays like the reference's (which rays share a thread
                              // differs, and only matters for a sample exactly on a face shared by two blocks)
    std::vector<double> image_col(std::max(n_q, 1));
    #pragma omp for schedule(dynamic, 16)
    for (int64_t ray = 0; ray < n_rays; ray++) {
      int64_t pixel = d->pixel_map != nullptr ? d->pixel_map[ray] : ray;
      double u_ind, v_ind;
      PixelIndices(o, *d, pixel, &u_ind, &v_ind);
      double cpos[4], cdir[4], factor;
      if (p->camera_type == BL_CAMERA_PLANE)
        SetPixelPlane(o, u_ind, v_ind, cpos, cdir, &factor);
      else
        SetPixelPinhole(o, u_ind, v_ind, cpos, cdir, &factor);
      if (d->camera_pos != nullptr)
        for (int mu = 0; mu < 4; mu++) d->camera_pos[4 * ray + mu] = cpos[mu];
      if (d->camera_dir != nullptr)
        for (int mu = 0; mu < 4; mu++) d->camera_dir[4 * ray + mu] = cdir[mu];

      int sample_num = 0;
      bool flag = false;
      if (p->ray_integrator == BL_INTEGRATOR_DP)
        IntegrateRayDP(o, cpos, cdir, b, &sample_num, &flag);
      else
        IntegrateRayRK(o, p->ray_integrator == BL_INTEGRATOR_RK4, cpos, cdir, b, &sample_num, &flag);
      sample_num = FinishRay(o, b, sample_num);
      if (d->sample_num != nullptr) d->sample_num[ray] = sample_num;
      if (d->sample_flags != nullptr) d->sample_flags[ray] = flag ? 1 : 0;
      total_samples += sample_num;
      total_flagged += flag ? 1 : 0;
      max_sample_num = std::max(max_sample_num, sample_num);
      if (extra != nullptr && extra->dump_ray == ray && extra->dump_pos != nullptr) {
        std::memcpy(extra->dump_pos, b.sample_pos.data(), sizeof(double) * 4 * sample_num);
        std::memcpy(extra->dump_dir, b.sample_dir.data(), sizeof(double) * 4 * sample_num);
        std::memcpy(extra->dump_len, b.sample_len.data(), sizeof(double) * sample_num);
        extra->dump_num = sample_num;
      }

      // coefficients: zero-initialised like j_i.Zero() / alpha_i.Zero(), cell_values NaN
      for (int l = 0; l < nf; l++)
        for (int n = 0; n < sample_num; n++) {
          b.j_i[static_cast<size_t>(l) * max_steps + n] = 0.0;
          b.alpha_i[static_cast<size_t>(l) * max_steps + n] = 0.0;
        }
      for (int a = 0; a < num_cell_values; a++)
        for (int n = 0; n < sample_num; n++)
          b.cell_values[a * static_cast<size_t>(max_steps) + n] = std::numeric_limits<double>::quiet_NaN();
      const double nan = std::numeric_limits<double>::quiet_NaN();
      if (o.image_polarization) {   // Zero() of the polarized coefficient and sample arrays
        for (auto &v : b.pol)
          for (int l = 0; l < nf; l++)
            for (int n = 0; n < sample_num; n++) v[static_cast<size_t>(l) * max_steps + n] = 0.0;
        for (int n = 0; n < 6 * sample_num; n++) b.sample_ub[n] = 0.0f;
      }
      if (simulation) {
        bool nan_ray = p->fallback_nan and flag;  // simulation_sampling.cpp:211-216
        for (int e = 0; e < 4; e++) {
          block_state.extrap[e] = false;
          block_state.extrap_val[e] = 0.0;
        }
        for (int n = 0; n < sample_num; n++) {
          Prims s;
          bool gathered = false;
          int status;
          if (nan_ray)
            status = 2;
          else
            status = SampleOne(o, &b.sample_pos[4 * n], &s, &gathered, &block_state);
          if (gathered) total_gathers++;
          if (status == 4) { count_undefined++; continue; }
          if (status == 5) { count_failed++; continue; }
          if (status == 1) continue;  // cut: simulation_coefficients.cpp:260-261
          if (status == 2) {
            float fnan = std::numeric_limits<float>::quiet_NaN();
            s = Prims{fnan, fnan, fnan, fnan, fnan, fnan, fnan, fnan, fnan};
          } else if (status == 3) {
            // fallback values (simulation_sampling.cpp:695-707); velocities and fields default to
            // zero there (radiation_integrator.hpp fallback_uu*, fallback_bb*)
            s = Prims{p->fallback_rho, p->fallback_pgas, p->fallback_kappa, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
          }
          double cell[num_cell_values];
          for (int a = 0; a < num_cell_values; a++) cell[a] = nan;
          double *pol[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
          if (o.image_polarization) {
            for (int c = 0; c < 6; c++) pol[c] = &b.pol[c][n];
            const float ub[6] = {s.uu1, s.uu2, s.uu3, s.bb1, s.bb2, s.bb3};
            for (int c = 0; c < 6; c++) b.sample_ub[6 * n + c] = ub[c];
          }
          SimulationCoefficientsOne(o, &b.sample_pos[4 * n], &b.sample_dir[4 * n], s, factor,
                                    &b.j_i[n], &b.alpha_i[n], max_steps, cell, o.image_polarization ? pol : nullptr);
          for (int a = 0; a < num_cell_values; a++) b.cell_values[a * static_cast<size_t>(max_steps) + n] = cell[a];
        }
        if (block_state.extrap[0]) { count_0++; val_0 = std::max(val_0, block_state.extrap_val[0]); }   // :553-575
        if (block_state.extrap[1]) { count_1++; val_1 = std::max(val_1, block_state.extrap_val[1]); }
        if (block_state.extrap[2]) { count_2++; val_2 = std::max(val_2, block_state.extrap_val[2]); }
        if (block_state.extrap[3]) { count_3++; val_3 = std::max(val_3, block_state.extrap_val[3]); }
      } else if (sample_num > 0) {
        if (p->fallback_nan and flag) {  // formula_coefficients.cpp:51-59 (only frequency 0 is filled)
          for (int n = 0; n < sample_num; n++) {
            b.j_i[n] = nan;
            b.alpha_i[n] = nan;
          }
        } else
          for (int n = 0; n < sample_num; n++)
            FormulaCoefficientsOne(o, &b.sample_pos[4 * n], &b.sample_dir[4 * n], factor, &b.j_i[n],
                                   &b.alpha_i[n], max_steps);
      }
      if (o.image_polarization)
        IntegratePolarizedOne(o, b, sample_num, max_steps, factor, cpos, cdir, image_col.data());
      else
        IntegrateUnpolarizedOne(o, b, sample_num, max_steps, factor, image_col.data());
      if (d->image != nullptr)
        for (int q = 0; q < n_q; q++) d->image[static_cast<size_t>(q) * n_rays + ray] = image_col[q];
      if (o.render_num_images > 0 && d->render != nullptr) {
        double render_col[3 * BL_MAX_RENDER_IMAGES];
        RenderOne(o, b, sample_num, max_steps, render_col);
        for (int q = 0; q < 3 * o.render_num_images; q++) d->render[static_cast<size_t>(q) * n_rays + ray] = render_col[q];
      }
    }
  }
  if (count_failed > 0) return Fail(err, err_len, "Error: Grid interpolation failed.\n", BL_E_INPUT);
  if (count_undefined > 0)
    return Fail(err, err_len, "oracle: inter-block interpolation reached the upper edge of the last MeshBlock, where the reference reads past its cell-centre arrays", BL_E_UNSUPPORTED);
  if (extra != nullptr) {
    extra->n_samples = total_samples;
    extra->n_gathers = total_gathers;
    extra->n_flagged = total_flagged;
    extra->max_sample_num = max_sample_num;
    extra->seconds = omp_get_wtime() - time_start;
    extra->slow_count[0] = count_0; extra->slow_count[1] = count_1; extra->slow_count[2] = count_2; extra->slow_count[3] = count_3;
    extra->slow_val[0] = val_0; extra->slow_val[1] = val_1; extra->slow_val[2] = val_2; extra->slow_val[3] = val_3;
  }
  return BL_OK;
}

}  // extern "C"

// bl_camera.h - camera frame (host, once per run) and per-pixel ray initial conditions
// (host + device, fused into the geodesic kernel's prologue).
//
// Replaces GeodesicIntegrator::InitializeCamera / AugmentCamera / SetPixelPlane / SetPixelPinhole
// (reference src/geodesic_integrator/camera.cpp:27-671). The reference fills camera_pos/dir(n_pix,4)
// and momentum_factors(n_pix) arrays; here a pixel's (x^mu, k_mu, 1/nu_local) is produced in
// registers from the seven frame vectors, and only written out if output_camera asks for it.
#ifndef BLACKLIGHT_AMD_BL_CAMERA_H_
#define BLACKLIGHT_AMD_BL_CAMERA_H_

#include "../../include/blacklight_amd.h"
#include "bl_geometry.h"

// What the per-pixel set-up needs, by value in kernel arguments
struct BlCameraDevice {
  double cam_x[4], u_con[4], u_cov[4], norm_con[4], norm_con_c[4], hor_con_c[4], vert_con_c[4];
  double camera_width, camera_r;
  int camera_type;          // BL_CAMERA_*
  int image_normalization;  // BL_NORM_*
  int camera_resolution;    // root resolution
  int level;                // adaptive level of this render
  int block_size;           // adaptive_block_size (level > 0)
  int effective_resolution; // camera_resolution * 2^level
};

// Fractional image-plane coordinates of a pixel (camera.cpp:393-396 root; :465-479 refined blocks).
// block_locs: [n_blocks][2] = (block_v, block_u), only read when level > 0.
BL_HD void bl_pixel_indices(const BlCameraDevice &cam, long long pixel, const int *block_locs,
                            double *u_ind, double *v_ind) {
  if (cam.level == 0) {
    int m2 = (int)(pixel / cam.camera_resolution);
    int m1 = (int)(pixel % cam.camera_resolution);
    *u_ind = (m1 - cam.camera_resolution / 2.0 + 0.5) / cam.camera_resolution;
    *v_ind = (m2 - cam.camera_resolution / 2.0 + 0.5) / cam.camera_resolution;
  } else {
    int block_num_pix = cam.block_size * cam.block_size;
    long long block = pixel / block_num_pix;
    int m = (int)(pixel % block_num_pix);
    int m_offset = block_locs[2 * block + 0] * cam.block_size;
    int l_offset = block_locs[2 * block + 1] * cam.block_size;
    int m2 = m / cam.block_size;
    int m1 = m % cam.block_size;
    *u_ind = (m1 + l_offset - cam.effective_resolution / 2.0 + 0.5) / cam.effective_resolution;
    *v_ind = (m2 + m_offset - cam.effective_resolution / 2.0 + 0.5) / cam.effective_resolution;
  }
}

// Position x^mu, covariant momentum k_mu and momentum factor 1/nu_local of one pixel
// (camera.cpp:528-585 plane, :608-671 pinhole).
BL_HD void bl_pixel_ray(const BlSpacetime &st, const BlCameraDevice &cam, double u_ind, double v_ind,
                        double position[4], double direction[4], double *factor) {
  double u = u_ind * st.bh_m * cam.camera_width;
  double v = v_ind * st.bh_m * cam.camera_width;
  double p[4];
  if (cam.camera_type == BL_CAMERA_PLANE) {
    double dtc = u * cam.hor_con_c[0] + v * cam.vert_con_c[0];
    double dxc = u * cam.hor_con_c[1] + v * cam.vert_con_c[1];
    double dyc = u * cam.hor_con_c[2] + v * cam.vert_con_c[2];
    double dzc = u * cam.hor_con_c[3] + v * cam.vert_con_c[3];
    double dt = cam.u_con[0] * dtc - (cam.u_cov[1] * dxc + cam.u_cov[2] * dyc + cam.u_cov[3] * dzc) / cam.u_cov[0];
    double dx = dxc + cam.u_con[1] * dtc;
    double dy = dyc + cam.u_con[2] * dtc;
    double dz = dzc + cam.u_con[3] * dtc;
    position[0] = cam.cam_x[0] + dt;
    position[1] = cam.cam_x[1] + dx;
    position[2] = cam.cam_x[2] + dy;
    position[3] = cam.cam_x[3] + dz;
    p[1] = cam.norm_con[1];
    p[2] = cam.norm_con[2];
    p[3] = cam.norm_con[3];
  } else {
    for (int mu = 0; mu < 4; mu++) position[mu] = cam.cam_x[mu];
    double normalization = bl_hypot3(u, v, cam.camera_r);
    double frac_norm = cam.camera_r / normalization;
    double frac_hor = -u / normalization;
    double frac_vert = -v / normalization;
    double dir_con_tc = cam.norm_con_c[0];
    for (int i = 1; i < 4; i++) {
      double dir_c = frac_norm * cam.norm_con_c[i] + frac_hor * cam.hor_con_c[i] + frac_vert * cam.vert_con_c[i];
      p[i] = dir_c + cam.u_con[i] * dir_con_tc;
    }
  }
  // time component from the null condition, then lower the index (camera.cpp:553-574)
  double gcov[4][4];
  bl_gcov(st, position[1], position[2], position[3], gcov);
  double temp_a = gcov[0][0];
  double temp_b = 0.0;
  for (int a = 1; a < 4; a++) temp_b += 2.0 * gcov[0][a] * p[a];
  double temp_c = 0.0;
  for (int a = 1; a < 4; a++)
    for (int b = 1; b < 4; b++) temp_c += gcov[a][b] * p[a] * p[b];
  double disc = temp_b * temp_b - 4.0 * temp_a * temp_c;
  double temp_d = blm_sqrt(disc > 0.0 ? disc : 0.0);  // std::max(disc, 0.0): NaN -> 0 as well
  p[0] = temp_a == 0.0 ? -temp_c / (2.0 * temp_b)
      : (temp_b < 0.0 ? 2.0 * temp_c / (temp_d - temp_b) : -(temp_b + temp_d) / (2.0 * temp_a));
  for (int mu = 0; mu < 4; mu++) {
    double acc = 0.0;
    for (int nu = 0; nu < 4; nu++) acc += gcov[mu][nu] * p[nu];
    direction[mu] = acc;
  }
  double nu_local = 0.0;
  if (cam.image_normalization == BL_NORM_CAMERA) {
    for (int mu = 0; mu < 4; mu++) nu_local -= direction[mu] * cam.u_con[mu];
  } else {
    nu_local = -direction[0];
  }
  *factor = 1.0 / nu_local;
}

// Host only: the serial frame construction of InitializeCamera (camera.cpp:52-380).
// The orthonormal triad (normal, horizontal, vertical) is built in the camera's rest frame from the
// Kerr-Schild metric at the camera, with the flat / polar special cases of the reference.
struct BlSphericalMetric {
  double cov_r_r, cov_r_th, cov_r_ph, cov_th_th, cov_th_ph, cov_ph_ph;
  double con_t_t, con_t_r, con_t_th, con_t_ph, con_r_r, con_r_th, con_r_ph, con_th_th, con_th_ph, con_ph_ph;
};

inline void bl_camera_frame_build(const bl_params &p, const BlSpacetime &st, bl_camera_frame *out) {
  const double bh_m = st.bh_m, bh_a = st.bh_a, camera_r = p.camera_r;
  const bool ray_flat = st.ray_flat != 0, camera_pole = p.camera_pole != 0;
  const double sth = bl_sin(p.camera_th), cth = bl_cos(p.camera_th);
  const double sph = bl_sin(p.camera_ph), cph = bl_cos(p.camera_ph);
  const double srot = bl_sin(p.camera_rotation), crot = bl_cos(p.camera_rotation);
  double *cam_x = out->cam_x, *u_con = out->u_con, *u_cov = out->u_cov, *norm_con = out->norm_con;
  double *norm_con_c = out->norm_con_c, *hor_con_c = out->hor_con_c, *vert_con_c = out->vert_con_c;

  // Position (:61-70)
  cam_x[0] = 0.0;
  cam_x[1] = sth * (camera_r * cph - bh_a * sph);
  cam_x[2] = sth * (camera_r * sph + bh_a * cph);
  cam_x[3] = camera_r * cth;
  if (ray_flat) {
    cam_x[1] = camera_r * sth * cph;
    cam_x[2] = camera_r * sth * sph;
  }
  const double z_sign = cam_x[3] >= 0.0 ? 1.0 : -1.0;

  // Spherical Kerr-Schild metric at the camera (:72-150)
  const double a2 = bh_a * bh_a;
  const double r2 = camera_r * camera_r;
  const double delta = r2 - 2.0 * bh_m * camera_r + a2;
  const double sigma = r2 + a2 * cth * cth;
  BlSphericalMetric g;
  g.cov_r_r = 1.0 + 2.0 * bh_m * camera_r / sigma;
  g.cov_r_th = 0.0;
  g.cov_r_ph = -(1.0 + 2.0 * bh_m * camera_r / sigma) * bh_a * sth * sth;
  g.cov_th_th = sigma;
  g.cov_th_ph = 0.0;
  g.cov_ph_ph = (r2 + a2 + 2.0 * bh_m * a2 * camera_r / sigma * sth * sth) * sth * sth;
  g.con_t_t = -(1.0 + 2.0 * bh_m * camera_r / sigma);
  g.con_t_r = 2.0 * bh_m * camera_r / sigma;
  g.con_t_th = 0.0;
  g.con_t_ph = 0.0;
  g.con_r_r = delta / sigma;
  g.con_r_th = 0.0;
  g.con_r_ph = bh_a / sigma;
  g.con_th_th = 1.0 / sigma;
  g.con_th_ph = 0.0;
  g.con_ph_ph = 1.0 / (sigma * sth * sth);
  if (ray_flat || camera_pole) {
    g.cov_r_th = g.cov_r_ph = g.cov_th_ph = 0.0;
    g.con_t_th = g.con_t_ph = g.con_r_th = g.con_r_ph = g.con_th_ph = 0.0;
  }
  if (ray_flat && !camera_pole) {
    g.cov_r_r = 1.0; g.cov_th_th = r2; g.cov_ph_ph = r2 * sth * sth;
    g.con_t_t = -1.0; g.con_t_r = 0.0; g.con_r_r = 1.0; g.con_th_th = 1.0 / r2;
    g.con_ph_ph = 1.0 / (r2 * sth * sth);
  }
  if (camera_pole && !ray_flat) {
    const double f = 2.0 * bh_m * camera_r / (r2 + a2);
    g.cov_r_r = 1.0 + f; g.cov_th_th = 1.0; g.cov_ph_ph = 1.0;
    g.con_t_t = -1.0 - f; g.con_t_r = z_sign * f; g.con_r_r = 1.0 - f; g.con_th_th = 1.0; g.con_ph_ph = 1.0;
  }
  if (ray_flat && camera_pole) {
    g.cov_r_r = 1.0; g.cov_th_th = 1.0; g.cov_ph_ph = 1.0;
    g.con_t_t = -1.0; g.con_t_r = 0.0; g.con_r_r = 1.0; g.con_th_th = 1.0; g.con_ph_ph = 1.0;
  }

  // Camera 4-velocity from its normal-frame components (:152-164)
  const double urn = p.camera_urn, uthn = p.camera_uthn, uphn = p.camera_uphn;
  const double alpha = 1.0 / blm_sqrt(-g.con_t_t);
  const double beta_con_r = -g.con_t_r / g.con_t_t;
  const double beta_con_th = -g.con_t_th / g.con_t_t;
  const double beta_con_ph = -g.con_t_ph / g.con_t_t;
  const double utn = blm_sqrt(1.0 + g.cov_r_r * urn * urn + 2.0 * g.cov_r_th * urn * uthn
      + 2.0 * g.cov_r_ph * urn * uphn + g.cov_th_th * uthn * uthn + 2.0 * g.cov_th_ph * uthn * uphn
      + g.cov_ph_ph * uphn * uphn);
  u_con[0] = utn / alpha;
  const double ur = urn - beta_con_r / alpha * utn;
  const double uth = uthn - beta_con_th / alpha * utn;
  const double uph = uphn - beta_con_ph / alpha * utn;

  // d(x,y,z)/d(r,th,ph) (:166-199)
  double jac[3][3];  // jac[i][c]: i in (x,y,z), c in (r,th,ph)
  jac[0][0] = sth * cph; jac[1][0] = sth * sph; jac[2][0] = cth;
  jac[0][1] = cth * (camera_r * cph - bh_a * sph);
  jac[1][1] = cth * (camera_r * sph + bh_a * cph);
  jac[2][1] = -camera_r * sth;
  jac[0][2] = sth * (-camera_r * sph - bh_a * cph);
  jac[1][2] = sth * (camera_r * cph - bh_a * sph);
  jac[2][2] = 0.0;
  if (ray_flat && !camera_pole) {
    jac[0][1] = camera_r * cth * cph; jac[1][1] = camera_r * cth * sph;
    jac[0][2] = -camera_r * sth * sph; jac[1][2] = camera_r * sth * cph;
  }
  if (camera_pole) {
    jac[0][0] = 0.0; jac[1][0] = 0.0; jac[2][0] = z_sign;
    jac[0][1] = 1.0; jac[1][1] = 0.0; jac[2][1] = 0.0;
    jac[0][2] = 0.0; jac[1][2] = 1.0; jac[2][2] = 0.0;
  }
  for (int i = 0; i < 3; i++) u_con[i + 1] = jac[i][0] * ur + jac[i][1] * uth + jac[i][2] * uph;
  double g_cov[4][4];
  bl_gcov(st, cam_x[1], cam_x[2], cam_x[3], g_cov);
  for (int mu = 0; mu < 4; mu++) {
    double acc = 0.0;
    for (int nu = 0; nu < 4; nu++) acc += g_cov[mu][nu] * u_con[nu];
    u_cov[mu] = acc;
  }

  // Photon momentum in the normal frame (:214-227)
  const double tt = g.con_t_t;
  const double g_rn_rn = (tt * g.con_r_r - g.con_t_r * g.con_t_r) / tt;
  const double g_rn_thn = (tt * g.con_r_th - g.con_t_r * g.con_t_th) / tt;
  const double g_rn_phn = (tt * g.con_r_ph - g.con_t_r * g.con_t_ph) / tt;
  const double g_thn_thn = (tt * g.con_th_th - g.con_t_th * g.con_t_th) / tt;
  const double g_thn_phn = (tt * g.con_th_ph - g.con_t_th * g.con_t_ph) / tt;
  const double g_phn_phn = (tt * g.con_ph_ph - g.con_t_ph * g.con_t_ph) / tt;
  const double k_rn = p.camera_k_r, k_thn = p.camera_k_th, k_phn = p.camera_k_ph;
  const double k_tn = -blm_sqrt(g_rn_rn * k_rn * k_rn + 2.0 * g_rn_thn * k_rn * k_thn
      + 2.0 * g_rn_phn * k_rn * k_phn + g_thn_thn * k_thn * k_thn + 2.0 * g_thn_phn * k_thn * k_phn
      + g_phn_phn * k_phn * k_phn);
  const double k_t = alpha * k_tn + (beta_con_r * k_rn + beta_con_th * k_thn + beta_con_ph * k_phn);

  // d(r,th,ph)/d(x,y,z) (:229-264)
  const double rr2 = cam_x[1] * cam_x[1] + cam_x[2] * cam_x[2] + cam_x[3] * cam_x[3];
  double dr_d[3], dth_d[3], dph_d[3];
  dr_d[0] = camera_r * cam_x[1] / (2.0 * r2 - rr2 + a2);
  dr_d[1] = camera_r * cam_x[2] / (2.0 * r2 - rr2 + a2);
  dr_d[2] = (camera_r * cam_x[3] + a2 * cam_x[3] / camera_r) / (2.0 * r2 - rr2 + a2);
  dth_d[0] = cam_x[3] * dr_d[0] / (r2 * sth);
  dth_d[1] = cam_x[3] * dr_d[1] / (r2 * sth);
  dth_d[2] = (cam_x[3] * dr_d[2] - camera_r) / (r2 * sth);
  dph_d[0] = -cam_x[2] / (cam_x[1] * cam_x[1] + cam_x[2] * cam_x[2]) + bh_a / (r2 + a2) * dr_d[0];
  dph_d[1] = cam_x[1] / (cam_x[1] * cam_x[1] + cam_x[2] * cam_x[2]) + bh_a / (r2 + a2) * dr_d[1];
  dph_d[2] = bh_a / (r2 + a2) * dr_d[2];
  if (ray_flat && !camera_pole) {
    dr_d[0] = cam_x[1] / camera_r; dr_d[1] = cam_x[2] / camera_r; dr_d[2] = cam_x[3] / camera_r;
    dth_d[0] = cth * cph / camera_r; dth_d[1] = cth * sph / camera_r; dth_d[2] = -sth / camera_r;
    dph_d[0] = -sph / (camera_r * sth); dph_d[1] = cph / (camera_r * sth); dph_d[2] = 0.0;
  }
  if (camera_pole) {
    dr_d[0] = 0.0; dr_d[1] = 0.0; dr_d[2] = z_sign;
    dth_d[0] = 1.0; dth_d[1] = 0.0; dth_d[2] = 0.0;
    dph_d[0] = 0.0; dph_d[1] = 1.0; dph_d[2] = 0.0;
  }
  double k_c[3];  // k_x, k_y, k_z (:266-270)
  for (int i = 0; i < 3; i++) k_c[i] = dr_d[i] * p.camera_k_r + dth_d[i] * p.camera_k_th + dph_d[i] * p.camera_k_ph;
  const double k_tc = u_con[0] * k_t + u_con[1] * k_c[0] + u_con[2] * k_c[1] + u_con[3] * k_c[2];

  // Spatial metric of the camera frame, contravariant (:272-280): gc_con[i][j], i <= j used
  double g_con[4][4];
  bl_gcon(st, cam_x[1], cam_x[2], cam_x[3], g_con);
  double gc_con[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = i; j < 3; j++) {
      gc_con[i][j] = g_con[i + 1][j + 1] + u_con[i + 1] * u_con[j + 1];
      gc_con[j][i] = gc_con[i][j];
    }

  // Unit normal (:282-303)
  double norm_cov_c[3];
  for (int i = 0; i < 3; i++) norm_cov_c[i] = k_c[i] - u_cov[i + 1] / u_cov[0] * k_t;
  norm_con_c[0] = -k_tc;
  for (int i = 0; i < 3; i++)
    norm_con_c[i + 1] = gc_con[0][i] * norm_cov_c[0] + gc_con[1][i] * norm_cov_c[1] + gc_con[2][i] * norm_cov_c[2];
  const double norm_norm = blm_sqrt(norm_cov_c[0] * norm_con_c[1] + norm_cov_c[1] * norm_con_c[2]
      + norm_cov_c[2] * norm_con_c[3]);
  for (int i = 0; i < 3; i++) norm_cov_c[i] /= norm_norm;
  for (int mu = 0; mu < 4; mu++) norm_con_c[mu] /= norm_norm;
  norm_con[0] = u_con[0] * norm_con_c[0]
      - (u_cov[1] * norm_con_c[1] + u_cov[2] * norm_con_c[2] + u_cov[3] * norm_con_c[3]) / u_cov[0];
  for (int i = 1; i < 4; i++) norm_con[i] = norm_con_c[i] + u_con[i] * norm_con_c[0];

  // Up direction and covariant camera-frame metric (:305-333)
  double up_con_c[3] = {0.0, 0.0, 1.0};
  if (camera_pole) {
    up_con_c[1] = 1.0;
    up_con_c[2] = 0.0;
  }
  double gc_cov[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = i; j < 3; j++) {
      gc_cov[i][j] = g_cov[i + 1][j + 1] - u_cov[i + 1] / u_cov[0] * g_cov[j + 1][0]
          - u_cov[j + 1] / u_cov[0] * g_cov[i + 1][0]
          + u_cov[i + 1] * u_cov[j + 1] / (u_cov[0] * u_cov[0]) * g_cov[0][0];
      gc_cov[j][i] = gc_cov[i][j];
    }

  // Unit vertical (:335-354)
  const double up_norm = up_con_c[0] * norm_cov_c[0] + up_con_c[1] * norm_cov_c[1] + up_con_c[2] * norm_cov_c[2];
  vert_con_c[0] = 0.0;
  for (int i = 0; i < 3; i++) vert_con_c[i + 1] = up_con_c[i] - up_norm * norm_con_c[i + 1];
  double vert_cov_c[3];
  for (int i = 0; i < 3; i++)
    vert_cov_c[i] = gc_cov[0][i] * vert_con_c[1] + gc_cov[1][i] * vert_con_c[2] + gc_cov[2][i] * vert_con_c[3];
  const double vert_norm = blm_sqrt(vert_cov_c[0] * vert_con_c[1] + vert_cov_c[1] * vert_con_c[2]
      + vert_cov_c[2] * vert_con_c[3]);
  for (int i = 0; i < 3; i++) vert_cov_c[i] /= vert_norm;
  for (int i = 1; i < 4; i++) vert_con_c[i] /= vert_norm;

  // Horizontal = vertical x normal / sqrt(det) (:356-366)
  const double det = gc_cov[0][0] * (gc_cov[1][1] * gc_cov[2][2] - gc_cov[1][2] * gc_cov[1][2])
      + gc_cov[0][1] * (gc_cov[1][2] * gc_cov[0][2] - gc_cov[0][1] * gc_cov[2][2])
      + gc_cov[0][2] * (gc_cov[0][1] * gc_cov[1][2] - gc_cov[1][1] * gc_cov[0][2]);
  const double det_sqrt = blm_sqrt(det);
  hor_con_c[0] = 0.0;
  hor_con_c[1] = (vert_cov_c[1] * norm_cov_c[2] - vert_cov_c[2] * norm_cov_c[1]) / det_sqrt;
  hor_con_c[2] = (vert_cov_c[2] * norm_cov_c[0] - vert_cov_c[0] * norm_cov_c[2]) / det_sqrt;
  hor_con_c[3] = (vert_cov_c[0] * norm_cov_c[1] - vert_cov_c[1] * norm_cov_c[0]) / det_sqrt;

  // Rotation about the normal (:368-380)
  for (int i = 1; i < 4; i++) {
    const double hor = hor_con_c[i], vert = vert_con_c[i];
    hor_con_c[i] = hor * crot - vert * srot;
    vert_con_c[i] = vert * crot + hor * srot;
  }
}
#endif  // BLACKLIGHT_AMD_BL_CAMERA_H_

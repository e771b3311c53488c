// bl_shade_fast.hip - the tolerant arithmetic tier's coefficient kernels (gfx950) other than the benchmark's (bl_shade_fused.hip):
// bl_shade_fast_kernel (behind a locate kernel), bl_shade_formula_fast_kernel. See bl_sampling_fast.h
// for what the tier is and what it keeps of the exact tier.
#include "bl_sampling_fast.h"

#pragma clang fp contract(fast)

// One redo-list entry per sample whose cut decision the tolerant tier leaves to the exact kernel
__device__ __forceinline__ void fast_defer(const BlShadeArgs &P, unsigned long long idx) {
  const unsigned long long at = atomicAdd(&P.counters[BL_CNT_REDO], 1ull);
  if (at < P.redo_capacity) P.redo_list[at] = idx;
}

// One sample of the tolerant tier: from its primitives (exact tier's trilinear read) and its record to the (a, c)
// transfer records of every frequency. Returns false when a cut decision is left to the exact kernel (nothing written).
// `table` (LDS): the 3 x 14 cut thresholds and guard bands of BlShadeCold, then the frequencies. Read from LDS so that
// nothing in here waits on the vector-memory counter, behind which the next sample's corner cells are in flight.
template <bool kSpinZero, bool kGeneral = false>
__device__ __forceinline__ bool fast_shade_sample(const BlShadeArgs &P, const double *table, const float pr[8], int status, size_t row,
                                                  double x, double y, double z, double kx, double ky, double kz, double kt,
                                                  double momentum_factor, double delta_lambda) {
  const BlSpacetime &st = P.st;
  const BlPlasmaDevice &pl = P.plasma;
  const double bh_m = st.bh_m;
  const double bh_a = kSpinZero ? 0.0 : st.bh_a;
  const double a2 = bh_a * bh_a;
  const double nan = __longlong_as_double(0x7ff8000000000000ll);
  double2 *out = P.transfer + row * P.n_nu;
  // what the loop over frequencies needs
  bool have = false;
  double nu_ratio = 0.0, n_e_cgs = 0.0, kb_tt_e_cgs = 0.0, k_u_inv = 0.0, b_sin = 0.0, b_sin_inv = 0.0;
  const double rho = pr[0], pgas = pr[1], uu1 = pr[2], uu2 = pr[3], uu3 = pr[4], bb1 = pr[5], bb2 = pr[6], bb3 = pr[7];
  if (status != kSampleCut) {
    // ---- Kerr-Schild scalars (radiation_geometry.cpp:18-25, :138-262)
    const double pp2 = x * x + y * y;
    const double rr2 = pp2 + z * z;
    double r2 = rr2;
    if (!kSpinZero) {
      const double u = rr2 - a2, v = 2.0 * bh_a * z;
      r2 = 0.5 * (u + bl_sqrt_g(u * u + v * v));
    }
    const double r_inv = fastmath::rsqrt(r2);
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
      const double td = fastmath::sqrt(tb * tb - 4.0 * ta * tc);
      // (the root that avoids cancellation, one reciprocal for either)
      double factor = (tb < 0.0 ? td - tb : -2.0 * tc) * fastmath::rcp(tb < 0.0 ? 2.0 * ta : tb + td);
      if (P.samples_renormalised) factor = 1.0;   // geodesic checkpoint: done before the samples were saved
      kx *= factor;
      ky *= factor;
      kz *= factor;
      lk *= factor;
    }
    double ut_inv, b_sq, k_u, k_b;
    const bool cartesian = kGeneral && pl.simulation_coord == BL_COORD_CKS;
    if (cartesian) {
      // ---- Cartesian Kerr-Schild simulation (the geodesic's own coordinates): g_ij = delta_ij + f l_i l_j, lapse^-2 = 1 + f,
      // shift^i = f l_i / (1 + f) (radiation_geometry.cpp:138-262 for both metrics); k_i needs no Jacobian
      const double lu = lx * uu1 + ly * uu2 + lz * uu3;
      const double u0n2 = 1.0 + (uu1 * uu1 + uu2 * uu2 + uu3 * uu3) + f * lu * lu;
      const double ut2 = u0n2 * (1.0 + f);
      ut_inv = fastmath::rsqrt(ut2);
      const double ut = ut2 * ut_inv;
      const double sh = f * fastmath::rcp(1.0 + f) * ut;
      const double ux = uu1 - sh * lx, uy = uu2 - sh * ly, uz = uu3 - sh * lz;
      const double flu = f * (ut + (lx * ux + ly * uy + lz * uz));
      const double u_x = ux + flu * lx, u_y = uy + flu * ly, u_z = uz + flu * lz;
      const double bt = u_x * bb1 + u_y * bb2 + u_z * bb3;
      const double lb = lx * bb1 + ly * bb2 + lz * bb3;
      b_sq = ((bb1 * bb1 + bb2 * bb2 + bb3 * bb3) + f * lb * lb + bt * bt) * ut_inv * ut_inv;
      k_u = kt * ut + kx * ux + ky * uy + kz * uz;
      k_b = kt * bt + (kx * (bb1 + bt * ux) + ky * (bb2 + bt * uy) + kz * (bb3 + bt * uz)) * ut_inv;
    } else {
      // ---- simulation metric, spherical Kerr-Schild (radiation_geometry.cpp:421-573); x^2 + y^2 = (r^2 + a^2) sin^2
      const double sth2 = pp2 * ra_inv;
      const double g_rr = 1.0 + hh;
      const double g_thth = sigma;
      const double g_tph = kSpinZero ? 0.0 : -hh * bh_a * sth2;
      const double g_rph = kSpinZero ? 0.0 : -g_rr * bh_a * sth2;
      const double g_phph = kSpinZero ? pp2 : (ra2 + hh * a2 * sth2) * sth2;
      // ---- u^mu from the normal-frame velocities (simulation_coefficients.cpp:297-313). u^t = u0n / lapse = sqrt(S (1 + 2 m r /
      // Sigma)): one reciprocal square root gives u^t and 1 / u^t. (Square roots and reciprocals are a fifth of this function's
      // issue time - a dozen instructions each, one of them at quarter rate - so quantities that share one are taken from one.)
      const double u0n2 = 1.0 + g_rr * uu1 * uu1 + 2.0 * g_rph * uu1 * uu3 + g_thth * uu2 * uu2 + g_phph * uu3 * uu3;
      const double ut2 = u0n2 * g_rr;
      ut_inv = fastmath::rsqrt(ut2);
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
      b_sq = (bb_sq_lab + bt * bt) * ut_inv * ut_inv;
      // ---- k_i in the simulation's coordinates: k'_a = k_i d x^i / d x'^a with the Jacobian of radiation_geometry.cpp:
      // 69-126, whose columns are (l_x, l_y, l_z), (cot(theta) x, cot(theta) y, -r sin(theta)) and (-y, x, 0)
      const double sth_inv = fastmath::rsqrt(sth2);
      const double k_r = lk;
      const double k_th = (lz * (x * kx + y * ky) - r * sth2 * kz) * sth_inv;
      const double k_ph = x * ky - y * kx;
      k_u = kt * ut + k_r * ur + k_th * uu2 + k_ph * uu3;
      k_b = kt * bt + k_r * br + k_th * bth + k_ph * bph;
    }
    // ---- plasma state (:274-358). Everything that is a product of units and parameters is one constant from the host
    // (BlShadeArgs::fast_k), and the cut thresholds come scaled to code units (BuildShadeArgs): no cgs value of rho, p, n_e or
    // Theta_e is formed per sample.
    // 1 / rho and 1 / p from one reciprocal where both are positive (single-precision values: the product is an ordinary double)
    double rho_inv, pgas_inv;
    {
      const double rp = rho * pgas;
      const bool both = rho > 0.0 && rp > 0.0 && rp < __builtin_inf();
      const double t = fastmath::rcp(both ? rp : rho);
      rho_inv = both ? t * pgas : t;
      pgas_inv = both ? t * rho : fastmath::rcp(pgas);
    }
    const double sigma_cut = b_sq * rho_inv;
    const double beta_inv = 0.5 * b_sq * pgas_inv;
    {
      // T_i / T_e = N / D, N = rat_high + rat_low / beta^2, D = 1 + 1 / beta^2: k T_e = (1 + c) k T_tot D / (N + c D), one reciprocal
      // (with plasma_use_p = false the three 1 / (gamma - 1) ride in the constants)
      const double bi2 = beta_inv * beta_inv;
      const double dd = 1.0 + bi2;
      kb_tt_e_cgs = P.fast_k[0] * (pgas * rho_inv) * (dd * fastmath::rcp(P.fast_k[1] + P.fast_k[2] * bi2 + P.fast_k[3] * dd));
    }
    // ---- cell cuts (:361-375): decided here unless a value sits within the guard band of an active threshold
    bool cell_cut = false, undecided = !cartesian && pp2 == 0.0;   // (on the polar axis of the spherical coordinates: the exact kernel's business)
    if (pl.cut_mask != 0) {
      const double bb = (pl.cut_mask & 0x300) ? fastmath::sqrt(b_sq) : 0.0;   // only the field-strength cuts need |b| itself
      const double value[7] = {rho, rho, pgas, kb_tt_e_cgs, bb, sigma_cut, beta_inv};   // against thresholds in these units
      // (one scalar test per quantity, then per bound; the three table values of a bound are read together and combined without
      // short-circuit branches)
#pragma unroll
      for (int v = 0; v < 7; v++)
        if ((pl.cut_mask >> (2 * v)) & 3) {
          const double q = value[v];
#pragma unroll
          for (int upper = 0; upper < 2; upper++) {
            const int c = 2 * v + upper;
            if ((pl.cut_mask >> c) & 1) {
              const double threshold = table[c], band_lo = table[14 + c], band_hi = table[28 + c];
              cell_cut = cell_cut | (upper ? q > threshold : q < threshold);
              undecided = undecided | ((q >= band_lo) & (q <= band_hi));
            }
          }
        }
    }
    if (undecided) return false;   // bl_shade_kernel<..., kRedo> writes this sample's records
    const bool no_field = bb1 == 0.0 && bb2 == 0.0 && bb3 == 0.0;   // :394
    if (!cell_cut && !no_field) {
      // cos^2 = (k.b)^2 / ((k.u)^2 b.b) (:434-455 in invariant form) and 1 / (k.u) from one reciprocal
      const double t = fastmath::rcp(k_u * b_sq);
      k_u_inv = t * b_sq;
      double cos2 = k_b * k_b * (t * k_u_inv);
      cos2 = cos2 < 1.0 ? cos2 : 1.0;
      have = true;
      nu_ratio = -k_u;                                                // :461-463
      // |b| sin(theta_B) and its reciprocal from one reciprocal square root: nu_c sin(theta_B) and nu_s carry nothing else of the field
      const double bs2 = b_sq * (1.0 - cos2);
      b_sin_inv = fastmath::rsqrt(bs2);                               // (inf along the field: nu / nu_s = inf there, as from 1 / 0)
      b_sin = bs2 > 0.0 ? bs2 * b_sin_inv : 0.0;
    }
  }
  if (status == kSampleOffGrid && pl.fallback_nan) {
    // primitives are NaN (simulation_sampling.cpp:377-384): j and alpha are NaN at every frequency, I <- I + NaN
    if (P.freq_split) {
      reinterpret_cast<double2 *>(P.freq_inputs + row)[0] = make_double2(2.0, 0.0);
      return true;
    }
    for (int l = 0; l < P.n_nu; l++) out[l] = make_double2(1.0, nan);
    if (kGeneral && P.tau_inc != nullptr)
      for (int l = 0; l < P.n_nu; l++) P.tau_inc[row * P.n_nu + l] = nan;
    return true;
  }
  // ---- per-frequency coefficients (simulation_coefficients.cpp:464-523) and transfer records (unpolarized.cpp:74-110)
  // Every frequency-dependent quantity factors into a part of the sample and a part of the frequency: with nu = s_nu f_l
  // (s_nu = -k.u x momentum factor), x = nu / nu_s = s_x f_l, so x^(1/2), x^(1/3), x^(1/6) are products of one square / cube
  // root per sample with the frequency's roots from the table (table[44 + n_nu ...], filled once per workgroup) - what is
  // left per sample AND frequency is one exp, two expm1, one reciprocal and two dozen multiplications.
  // nu_c = e |b| b_unit / (2 pi m_e c), nu_s = 2/9 nu_c Theta_e^2 sin(theta_B): nu / nu_s without another reciprocal
  const double power_frac = pl.power_frac;
  const double nu_c_over_b = P.fast_k[7];
  const double momentum_factor_inv = fastmath::rcp(momentum_factor);
  const double kb_tt_e_inv = have ? fastmath::rcp(kb_tt_e_cgs) : 0.0;
  const double s_nu = nu_ratio * momentum_factor;
  const double s_x = have ? s_nu * b_sin_inv * (kb_tt_e_inv * kb_tt_e_inv) * P.fast_k[4] : 0.0;
  // x^(1/3) by the cube root, x^(1/6) as its square root, x^(1/2) as the cube of that
  const double s_1_3 = fastmath::cbrt(s_x);
  const double s_1_6 = fastmath::sqrt(s_1_3);
  const double s_1_2 = s_1_6 * s_1_3;
  const double s_planck = have ? kH * s_nu * kb_tt_e_inv : 0.0;                                   // h nu / (k T_e) = s_planck f_l
  const double s_nu_inv = have ? -k_u_inv * momentum_factor_inv : 0.0;
  const double s_j = P.fast_k[5] * (rho * b_sin) * (s_nu_inv * s_nu_inv);
  if (kGeneral) n_e_cgs = P.fast_k[6] * rho;   // (the power-law terms below)
  const double s_length = delta_lambda * P.x_unit * momentum_factor_inv;                          // unpolarized.cpp:75-76
  if (P.freq_split) {   // several frequencies: the factors go to bl_transfer_freq_kernel, one lane per ray and frequency
    double2 *dst = reinterpret_cast<double2 *>(P.freq_inputs + row);
    dst[0] = make_double2(have ? 1.0 : 0.0, s_1_2);
    dst[1] = make_double2(s_1_3, s_1_6);
    dst[2] = make_double2(s_planck, s_j);
    dst[3] = make_double2(s_length, 0.0);
    return true;
  }
  const int n_nu = P.n_nu;
  for (int l = 0; l < n_nu; l++) {
    double2 rec = make_double2(1.0, 0.0);
    double delta_tau_out = 0.0;   // what the sample adds to an optical-depth image (unpolarized.cpp:150-151)
    if (have) {
      const double f = table[44 + l], f_1_2 = table[44 + n_nu + l], f_1_3 = table[44 + 2 * n_nu + l], f_1_6 = table[44 + 3 * n_nu + l];
      const double f_inv = table[44 + 4 * n_nu + l];
      const double xx_1_3 = s_1_3 * f_1_3;
      const double var_c = s_1_2 * f_1_2 + kPow2_11_12 * (s_1_6 * f_1_6);
      const double j_val = s_j * (f_inv * f_inv) * fastmath::exp(-xx_1_3) * var_c * var_c;
      // (thin steps in Rayleigh-Jeans plasma - nearly every sample - need neither expm1 nor a division: bl_transfer_freq_kernel)
      const double xp = s_planck * f;
      const double planck = xp < 0x1p-10 ? xp * (1.0 + 0.5 * xp * (1.0 + (1.0 / 3.0) * xp * (1.0 + 0.25 * xp))) : fastmath::expm1(xp);
      const double inv_b_nu = planck * (kC * kC / (2.0 * kH));   // 1 / (B_nu / nu^3)
      double alpha_val = j_val * inv_b_nu;
      if (alpha_val * alpha_val <= 0x1p-1024) alpha_val = 0.0;                                // :513-523
      double j_total = j_val;
      if (kGeneral && power_frac != 0.0) {
        // power-law electrons (simulation_coefficients.cpp:556-584): nu / (nu_c sin theta_B) to two powers, one logarithm; the
        // field enters through |b| sin theta_B alone here as well
        const double nu_cgs = s_nu * f;
        const fastmath::PowBase ratio = fastmath::pow_base(nu_cgs * b_sin_inv * (1.0 / nu_c_over_b));
        const double common = power_frac * n_e_cgs * kE * kE;
        j_total += common * (nu_c_over_b * b_sin) * (1.0 / kC) * (s_nu_inv * f_inv) * (s_nu_inv * f_inv) * pl.power_jj
            * fastmath::pow_of(ratio, -(pl.plasma_p - 1.0) * 0.5);
        alpha_val += common * (1.0 / (kMe * kC)) * pl.power_aa * fastmath::pow_of(ratio, -(pl.plasma_p + 2.0) * 0.5);
      }
      const double delta_lambda_cgs = s_length * f_inv;
      delta_tau_out = alpha_val * delta_lambda_cgs;
      if (alpha_val > 0.0) {
        const double delta_tau = alpha_val * delta_lambda_cgs;
        if (delta_tau < 0x1p-10) {
          const double p = 1.0 - 0.5 * delta_tau * (1.0 - (1.0 / 3.0) * delta_tau * (1.0 - 0.25 * delta_tau));
          rec = make_double2(1.0 - delta_tau * p, j_total * delta_lambda_cgs * p);
        } else if (delta_tau <= kDeltaTauMax) {
          const double e1 = fastmath::expm1(-delta_tau);
          rec = make_double2(1.0 + e1, -(j_total * fastmath::rcp(alpha_val)) * e1);
        } else {
          rec = make_double2(BL_AFFINE_THICK, j_total * fastmath::rcp(alpha_val));
        }
      } else {
        rec = make_double2(1.0, j_total * delta_lambda_cgs);
      }
    }
    out[l] = rec;
    if (kGeneral && P.tau_inc != nullptr) P.tau_inc[row * n_nu + l] = delta_tau_out;
  }
  return true;
}

// The tolerant tier's form of gather_finish(): the same weights, the eight products summed with fused multiply-adds (56 additions
// fewer per sample). The sums differ from the reference's by a few units in the last place of a double - and, in the fused kernel,
// by what the tier's own angles move the fractions: up to ~2e-15 / cell width, times whatever the neighbouring cells' spread is
// against the interpolated value (large next to empty cells) - which the conversion to float hides, unless a sum lies that close to
// the midpoint of two floats, where the two could round apart and move a primitive by 6e-8: then the function returns true and the
// sample is left to the exact kernel. The midpoint is where the 29 bits below a float's precision read 2^28; the window around it
// is 2 048 units (2e-13 of the value) for the positive sums of density and pressure, 4 096 for the components of velocity and field,
// whose terms may cancel (a sum that is a 500th of its terms or less is a component that small beside the others: a unit of its float
// precision is 1e-10 of the vector); 1e-4 of the samples go to the exact pass. A randomised sweep at 256^2 ... 512^2
// (tools/gpu_fuzz_tiers.py) found one pixel in 6.5 million at 1.3e-10 with the earlier 64 units for density and pressure; 8 192 units
// for everything cost 0.35 ms per frame in the exact pass and found nothing more. What remains possible beyond the window is that
// same effect: one float's last place in one sample.
__device__ __forceinline__ bool gather_finish_tolerant(const BlShadeArgs &P, float fallback_rho, float fallback_pgas, int status, const float4 (&lo)[8],
                                                       const float4 (&hi)[8], double f_i, double f_j, double f_k, float pr[8]) {
  const BlPlasmaDevice &pl = P.plasma;
  bool near_midpoint = false;
  if (status == kSampleInterp || status == kSampleAdvanced) {   // (advanced: InterpolateAdvanced's weights are these, its cells the eight anchors)
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
          val[q] = __builtin_fma(w, (double)v[q], val[q]);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const uint32_t below = (uint32_t)__double_as_longlong(val[q]) & 0x1fffffffu;
      const uint32_t window = q < 2 ? 2048u : 4096u;
      near_midpoint = near_midpoint || (below - (0x10000000u - window)) <= 2u * window;
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
  return near_midpoint;
}

// Tolerant tier's coefficient kernel, one sample per lane, software-pipelined over the samples of a lane: the corner
// cells of the NEXT sample are requested before the arithmetic of this one (64 registers in flight), its record with
// them, its located sample one sample earlier still. With two waves per SIMD the gather's latency then lies behind
// ~1 700 instructions of arithmetic instead of in front of them (the unpipelined version waited for memory in 54 % of its
// wave cycles). Nothing between the requests and the end of the arithmetic reads global memory: thresholds, fallback
// values and frequencies sit in LDS, the per-ray constants are requested before the cells.
// kMode: 0 thermal electrons on a spherical Kerr-Schild grid; 1 the general arithmetic (power laws, Cartesian grids, an optical-depth
// image); 2 the general arithmetic behind inter-block interpolation (anchor cells); 3 behind slow light (time slices)
template <bool kSpinZero, int kMode>
__global__ void __launch_bounds__(256, BL_FAST_WAVES) bl_shade_fast_kernel(const BlShadeArgs P_at_entry) {
  constexpr bool kGeneral = kMode != 0, kAnchors = kMode == 2, kSlices = kMode == 3;
  // (the instantiation for anchor cells and time slices reads its arguments where it uses them - bl_kernel_util.h: held in scalar
  // registers from the entry on, a hundred of them spill into vector lanes it has no room for)
  const BlShadeArgs &P = (kSlices || kAnchors) ? kernel_arguments_in_place<BlShadeArgs>() : P_at_entry;
  const unsigned long long n_records = P.counters_in[BL_CNT_RECORDS];
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  unsigned long long idx = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;   // record of `next`
  // LDS: 3 x 14 cut thresholds / guard bands, the two fallback primitives, the frequencies and their roots / reciprocals
  extern __shared__ double fast_table[];
  // (slow light: behind the table, the cell array of every time slice of the window)
  const unsigned long long *slice_cells = reinterpret_cast<const unsigned long long *>(fast_table + 44 + 5 * P.n_nu);
  if (kSlices && P.slow.n > 0)
    for (int i = threadIdx.x; i < P.slow.n; i += blockDim.x)
      reinterpret_cast<unsigned long long *>(fast_table + 44 + 5 * P.n_nu)[i] = reinterpret_cast<unsigned long long>(P.slow.cells[i]);
  for (int i = threadIdx.x; i < 44 + 5 * P.n_nu; i += blockDim.x) {
    const BlShadeCold &cc = *P.cold;
    double value;
    if (i < 44) {
      value = i < 14 ? cc.fast_cut[i] : (i < 28 ? cc.fast_cut_lo[i - 14] : (i < 42 ? cc.fast_cut_hi[i - 28]
          : (i == 42 ? (double)cc.fallback_rho : (double)cc.fallback_pgas)));
    } else {   // f, f^(1/2), f^(1/3), f^(1/6), 1 / f of every frequency (fast_shade_sample)
      const int which = (i - 44) / P.n_nu;
      const double f = P.frequencies[(i - 44) - which * P.n_nu];
      const double f_1_3 = fastmath::cbrt(f);
      value = which == 0 ? f : (which == 1 ? bl_sqrt_g(f) : (which == 2 ? f_1_3 : (which == 3 ? bl_sqrt_g(f_1_3) : fastmath::rcp(f))));
    }
    fast_table[i] = value;
  }
  __syncthreads();
  if (n_records == 0ull) return;
  // Three samples in flight per lane, one call site per stage:
  //   next: located sample being loaded;
  //   cur:  located sample here -> corner cells requested in this iteration, record halves requested with them;
  //   prev: corner cells and record arriving -> trilinear read at the top of the iteration, arithmetic at its end.
  // A stage without a sample (pipeline filling / draining, lanes beyond the last record) works on record n_records - 1
  // and discards the result, so that the loads of the loop are the same on every path.
  const unsigned long long last = n_records - 1ull;
  FastLocated loc_prev, loc_cur, loc_next;
  FastRay ray_prev, ray_cur;
  float4 lo[8], hi[8];
  unsigned long long idx_prev = 0ull, idx_cur = 0ull;
  bool have_prev = false, have_cur = false, have_next = idx < n_records;
  loc_prev.tag = loc_cur.tag = 0ull;
  loc_prev.l0 = loc_prev.l1 = loc_cur.l0 = loc_cur.l1 = make_double2(0.0, 0.0);
  ray_prev.q0 = ray_prev.q1 = ray_prev.q2 = ray_prev.q3 = make_double2(0.0, 0.0);
  ray_prev.q1.y = __longlong_as_double((long long)BL_DEAD_RAY);
#pragma unroll
  for (int c = 0; c < 8; c++) lo[c] = hi[c] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  fast_load_located(P, have_next ? idx : last, loc_next);
  // Inter-block interpolation (the general instantiation; wave-uniform): the eight cells are the anchors the locate kernel named, loaded
  // with the located sample and requested like corner cells one stage later. Slow light: the primitives come from the exact tier's own
  // sampling function - one or two time slices - read where they lie, not through the pipelined reads; what the tier saves there
  // is the arithmetic behind them (simulation_sampling.cpp:505-546, :736-912)
  // Slow light without it: the first slice's cells come through the pipelined reads (from that slice's array), the second slice's are
  // requested into the same registers once the first slice's eight values are formed - one exposed round trip per sample where the
  // plain loop of the exact kernel has sixteen - and both are summed in the exact tier's order: its bits, nothing to guard.
  const bool slow_pipelined = kSlices && P.slow.n > 0 && P.anchors == nullptr;
  constexpr bool sampled_elsewhere = false;   // (slow light WITH inter-block interpolation stays in the exact tier: bl_render.hip)
  const bool by_anchors = kAnchors && P.anchors != nullptr && P.slow.n == 0;
  FastAnchors anchors_cur, anchors_next;
  anchors_cur.lo = anchors_cur.hi = anchors_next.lo = anchors_next.hi = make_uint4(0u, 0u, 0u, 0u);
  if (kAnchors && by_anchors) fast_load_anchors(P, have_next ? idx : last, anchors_next);
  while (have_prev || have_cur || have_next) {
    // (a dead record slot carries tag 0 = kSampleNone from the locate kernel: cell 0 was requested for it)
    const uint32_t ray = have_prev ? (uint32_t)__double_as_longlong(ray_prev.q1.y) : BL_DEAD_RAY;
    const bool live = ray != BL_DEAD_RAY;
    const uint32_t n = (uint32_t)(((unsigned long long)__double_as_longlong(ray_prev.q1.y)) >> 32);
    const int status = (int)(loc_prev.tag >> 32) & 0xff;
    // per-ray constants of `prev`: requested before the next sample's cells, so that waiting for them does not wait for those
    const double kt = P.ray_kt[live ? ray : 0u], momentum_factor = P.ray_factor[live ? ray : 0u];
    const size_t row = (size_t)P.ray_offset[live ? ray : 0u] + n;
    float pr[8];
    const bool on_slices = kSlices && slow_pipelined && live && (status == kSampleInterp || status == kSampleNearest);
    bool near_midpoint = gather_finish_tolerant(P, (float)fast_table[42], (float)fast_table[43], (live && !sampled_elsewhere && !on_slices) ? status : (int)kSampleNone, lo, hi,
                                                loc_prev.l0.x, loc_prev.l0.y, loc_prev.l1.x, pr);
    if (kSlices && slow_pipelined) {
      double val[8];
      slice_values_from_cells(on_slices ? status : (int)kSampleNearest, lo, hi, loc_prev.l0.x, loc_prev.l0.y, loc_prev.l1.x, val);
      if (P.slow.interp) {   // (wave-uniform) the later slice, into the registers the earlier one has left
        const int t_ind = (int)(loc_prev.tag >> 40);
        gather_issue_slice(reinterpret_cast<const float *>(slice_cells[on_slices ? t_ind + 1 : 0]), P.grid, on_slices ? status : (int)kSampleNone, (uint32_t)loc_prev.tag, lo, hi);
        slice_blend_from_cells(on_slices ? status : (int)kSampleNearest, lo, hi, loc_prev.l0.x, loc_prev.l0.y, loc_prev.l1.x, P.slow.frac[idx_prev], val);
      }
      if (on_slices) {
#pragma unroll
        for (int q = 0; q < 8; q++) pr[q] = (float)val[q];
        near_midpoint = false;
      }
    }
    if (kAnchors && by_anchors) gather_issue_anchors(P, have_cur ? (int)(loc_cur.tag >> 32) & 0xff : (int)kSampleNone, (uint32_t)loc_cur.tag, anchors_cur, lo, hi);
    else if (kSlices && slow_pipelined)
      gather_issue_slice(reinterpret_cast<const float *>(slice_cells[have_cur ? (int)(loc_cur.tag >> 40) : 0]), P.grid, have_cur ? (int)(loc_cur.tag >> 32) & 0xff : (int)kSampleNone,
                         (uint32_t)loc_cur.tag, lo, hi);
    else gather_issue(P, (have_cur && !sampled_elsewhere) ? (int)(loc_cur.tag >> 32) & 0xff : (int)kSampleNone, (uint32_t)loc_cur.tag, lo, hi);
    fast_load_ray(P, have_cur ? idx_cur : last, ray_cur);
    const FastRay rec = ray_prev;
    const unsigned long long idx_rec = idx_prev;
    loc_prev = loc_cur;
    ray_prev = ray_cur;
    idx_prev = idx_cur;
    have_prev = have_cur;
    loc_cur = loc_next;
    idx_cur = idx;
    have_cur = have_next;
    idx += stride;
    have_next = have_next && idx < n_records;
    fast_load_located(P, have_next ? idx : last, loc_next);
    if (kAnchors && by_anchors) {
      anchors_cur = anchors_next;
      fast_load_anchors(P, have_next ? idx : last, anchors_next);
    }
    if (live) {
      // ReverseGeodesics: sample_len = -geodesic_len (geodesics.cpp:840)
      if (near_midpoint || !fast_shade_sample<kSpinZero, kGeneral>(P, fast_table, pr, status, row, rec.q0.x, rec.q0.y, rec.q1.x, rec.q2.x, rec.q2.y,
                                                                    rec.q3.x, kt, momentum_factor, -rec.q3.y)) {
        fast_defer(P, idx_rec);
      }
    }
  }
}

// Tolerant tier, formula mode (formula_coefficients.cpp:62-180; BASELINE configuration 2): one sample per lane, no grid. The
// reference finds the azimuth with atan2 and atan, takes its sine and cosine, and builds u^mu through the Boyer-Lindquist
// metric and the Jacobian to Cartesian Kerr-Schild coordinates; with u_r = u_theta = 0 that Jacobian collapses - r (sin
// theta sin phi) + a (sin theta cos phi) = y and r (sin theta cos phi) - a (sin theta sin phi) = x identically - to
// u^mu = (u^t, -y u^phi, x u^phi, 0): no trigonometric function at all. Powers share one logarithm. The cut at camera_r is
// not decided within 1e-9 of it, nor anything on the polar axis: those samples go to the exact kernel's second pass.
__global__ void __launch_bounds__(256, 4) bl_shade_formula_fast_kernel(const BlShadeArgs P) {
  const BlFormulaDevice fm = P.formula;
  const double bh_m = P.st.bh_m, bh_a = P.st.bh_a, a2 = bh_a * bh_a;
  const bool flat = P.st.ray_flat != 0;
  const unsigned long long n_records = P.counters_in[BL_CNT_RECORDS];
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  const double r0_inv2 = 1.0 / (fm.r0 * fm.r0), h2 = fm.h * fm.h, nup_inv = 1.0 / fm.nup;
  const double band_lo = P.cuts.camera_r * (1.0 - 1.0e-9), band_hi = P.cuts.camera_r * (1.0 + 1.0e-9);
  unsigned long long idx = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool more = idx < n_records;
  FastRay next;
  next.q0 = next.q1 = next.q2 = next.q3 = make_double2(0.0, 0.0);
  if (more) fast_load_ray(P, idx, next);
  while (more) {
    const FastRay rec = next;
    const unsigned long long idx_rec = idx;
    idx += stride;
    more = idx < n_records;
    if (more) fast_load_ray(P, idx, next);
    const uint32_t ray = (uint32_t)__double_as_longlong(rec.q1.y);
    if (ray == BL_DEAD_RAY) continue;
    const uint32_t n = (uint32_t)(((unsigned long long)__double_as_longlong(rec.q1.y)) >> 32);
    const double x = rec.q0.x, y = rec.q0.y, z = rec.q1.x;
    double kx = rec.q2.x, ky = rec.q2.y, kz = rec.q3.x;
    const double delta_lambda = -rec.q3.y;   // ReverseGeodesics: sample_len = -geodesic_len (geodesics.cpp:840)
    const double kt = P.ray_kt[ray], momentum_factor = P.ray_factor[ray];
    double2 *out = P.transfer + ((size_t)P.ray_offset[ray] + n) * P.n_nu;
    // ---- Kerr-Schild scalars
    const double pp2 = x * x + y * y, rr2 = pp2 + z * z;
    const double uu = rr2 - a2, vv = 2.0 * bh_a * z;
    const double r2 = 0.5 * (uu + bl_sqrt_g(uu * uu + vv * vv));
    const double r_inv = fastmath::rsqrt(r2);
    const double r = r2 * r_inv;
    if ((r >= band_lo && r <= band_hi) || pp2 == 0.0) {   // the exact kernel decides the cut / handles the axis
      fast_defer(P, idx_rec);
      continue;
    }
    if (r > P.cuts.camera_r) {                             // formula_coefficients.cpp:72-75: j = alpha = 0
      for (int l = 0; l < P.n_nu; l++) out[l] = make_double2(1.0, 0.0);
      continue;
    }
    // ---- null-condition renormalisation of the stored momentum (geodesics.cpp:352-371)
    if (!P.samples_renormalised) {
      double f = 0.0, lx = 0.0, ly = 0.0, lz = 0.0;
      if (!flat) {
        const double ra_inv = fastmath::rcp(r2 + a2);
        lx = (r * x + bh_a * y) * ra_inv;
        ly = (r * y - bh_a * x) * ra_inv;
        lz = z * r_inv;
        f = 2.0 * bh_m * r2 * r * fastmath::rcp(r2 * r2 + a2 * z * z);
      }
      const double lk = lx * kx + ly * ky + lz * kz;
      const double kk = kx * kx + ky * ky + kz * kz;
      const double ta = kk - f * lk * lk, tb = 2.0 * kt * f * lk, tc = -(1.0 + f) * kt * kt;
      const double td = bl_sqrt_g(tb * tb - 4.0 * ta * tc);
      const double factor = tb < 0.0 ? (td - tb) * fastmath::rcp(2.0 * ta) : -2.0 * tc * fastmath::rcp(tb + td);
      kx *= factor;
      ky *= factor;
      kz *= factor;
    }
    // ---- Boyer-Lindquist metric at the sample and the model's rotation law (:121-147)
    const double cth = z * r_inv, cth2 = cth * cth, sth2 = 1.0 - cth2;
    const double rr = bl_sqrt_g(r2 - z * z);
    const double delta = r2 - 2.0 * bh_m * r + a2, sigma = r2 + a2 * cth2;
    const double ds_inv = fastmath::rcp(delta * sigma);
    const double gtt = -(1.0 + 2.0 * bh_m * r * (r2 + a2) * ds_inv);
    const double gtph = -2.0 * bh_m * bh_a * r * ds_inv;
    const double gphph = (sigma - 2.0 * bh_m * r) * ds_inv * fastmath::rcp(sth2);
    const double ll = fm.l0 * fastmath::rcp(1.0 + rr) * fastmath::pow(rr, 1.0 + fm.q);
    const double u_norm = fastmath::rsqrt(-gtt + 2.0 * gtph * ll - gphph * ll * ll);
    const double ut = u_norm * (gtph * ll - gtt);
    const double uph = u_norm * (gphph * ll - gtph);
    const double nu_ratio = -(ut * kt + uph * (x * ky - y * kx));   // -u^mu k_mu with u^mu = (u^t, -y u^phi, x u^phi, 0)
    const double n_n0 = fastmath::exp(-0.5 * (r2 * r0_inv2 + h2 * cth2));
    // ---- per frequency (:164-179) and the transfer record (unpolarized.cpp:74-110)
    for (int l = 0; l < P.n_nu; l++) {
      const double freq = P.frequencies[l];
      const double nu = nu_ratio * freq * momentum_factor;
      const fastmath::PowBase base = fastmath::pow_base(nu * nup_inv);
      const double nu_inv = fastmath::rcp(nu);
      const double j_val = fm.cn0 * n_n0 * fastmath::pow_of(base, -fm.alpha) * nu_inv * nu_inv;
      const double alpha_val = fm.a * fm.cn0 * n_n0 * fastmath::pow_of(base, -fm.beta - fm.alpha) * nu;
      const double delta_lambda_cgs = delta_lambda * P.x_unit * fastmath::rcp(freq * momentum_factor);
      double2 rec_out;
      if (alpha_val > 0.0) {
        const double delta_tau = alpha_val * delta_lambda_cgs;
        if (delta_tau <= kDeltaTauMax) {
          const double e1 = fastmath::expm1(-delta_tau);
          rec_out = make_double2(1.0 + e1, -(j_val * fastmath::rcp(alpha_val)) * e1);
        } else {
          rec_out = make_double2(BL_AFFINE_THICK, j_val * fastmath::rcp(alpha_val));
        }
      } else {
        rec_out = make_double2(1.0, j_val * delta_lambda_cgs);
      }
      out[l] = rec_out;
    }
  }
}
#pragma clang fp contract(off)

// =================================================================================================
// Launch wrappers (called from bl_render.hip)
// =================================================================================================
// The exact tier's kernel over the records a tolerant kernel deferred (bl_shade.hip)
extern "C" hipError_t bl_launch_shade_redo(const BlShadeArgs *args, int model, int grid, hipStream_t stream);
extern "C" hipError_t bl_launch_shade_fused2(const BlShadeArgs *args, int grid, hipStream_t stream);   // bl_shade_fused.hip

// Tolerant tier in formula mode: the fast kernel, then the exact kernel over the records it deferred
extern "C" hipError_t bl_launch_shade_formula_fast(const BlShadeArgs *args, int grid, hipStream_t stream) {
  hipLaunchKernelGGL(bl_shade_formula_fast_kernel, dim3(grid), dim3(256), 0, stream, *args);
  return bl_launch_shade_redo(args, BL_MODEL_FORMULA, grid, stream);
}

// Tolerant tier, simulations: the fast coefficient kernel, then the exact kernel over the records it deferred
extern "C" hipError_t bl_launch_shade_fast(const BlShadeArgs *args, int grid, hipStream_t stream) {
  const bool spin_zero = args->st.bh_a == 0.0;
  if (args->located == nullptr) {   // no locate kernel ran: the kernel with the locate step inside (bl_shade_fused.hip)
    const hipError_t err = bl_launch_shade_fused2(args, grid, stream);
    if (err != hipSuccess) return err;
    return bl_launch_shade_redo(args, BL_MODEL_SIMULATION, grid, stream);
  }
  const size_t lds = (44 + 5 * args->n_nu + (args->slow.n > 0 ? args->slow.n : 0)) * sizeof(double);
  // power laws, Cartesian grids and an optical-depth image go through the general instantiation; inter-block interpolation and slow light
  // through the one that also knows anchor cells and time slices
  const bool slices = args->anchors != nullptr || args->slow.n > 0;
  const bool general = args->plasma.power_frac != 0.0 || args->tau_inc != nullptr || args->plasma.simulation_coord == BL_COORD_CKS;
#define BL_LAUNCH_F(SPIN, MODE) hipLaunchKernelGGL((bl_shade_fast_kernel<SPIN, MODE>), dim3(grid), dim3(256), lds, stream, *args)
  if (slices) {
    if (args->slow.n > 0) BL_LAUNCH_F(false, 3); else BL_LAUNCH_F(false, 2);
  } else if (general) {
    BL_LAUNCH_F(false, 1);
  } else {
    if (spin_zero) BL_LAUNCH_F(true, 0); else BL_LAUNCH_F(false, 0);
  }
#undef BL_LAUNCH_F
  return bl_launch_shade_redo(args, BL_MODEL_SIMULATION, grid, stream);
}

// bl_coefficients_freq.hip - per-frequency coefficient kernels behind the coefficient kernel proper (gfx950): the polarized
// coefficients of a sample (both tiers), the frame of a polarized sample, the per-frequency factors of many-frequency renders in
// the tolerant tier; and the element-wise diagnostics kernel the math tests use.
#include "bl_sampling_fast.h"

// =================================================================================================
// Polarized coefficient kernel: the per-frequency part of CalculateSimulationCoefficients (simulation_coefficients.cpp:
// 458-698) for polarized runs, one sample record per lane from the scalars the coefficient kernel left
// (BlCoefInputs). Writes (j_I, alpha_I) and the three polarized pairs, [ray][n][frequency].
// =================================================================================================
// kTolerant: the tolerant arithmetic tier's functions and fused multiply-adds in the formulas (bl_coefficients.inc,
// second inclusion) - the kernel is almost entirely the double-double pow / log of the pinned library otherwise.
#ifndef BL_POLCOEF_WAVES
#define BL_POLCOEF_WAVES 2
#endif
// kThermalOnly: no power-law and no kappa-distribution electrons (the host knows): without their formulas the exact kernel fits three
// waves per SIMD instead of two (168 registers), which is worth 8 ms of a 1024^2 frame's 106.
template <bool kTolerant, bool kThermalOnly>
__global__ void __launch_bounds__(256, (kThermalOnly && !kTolerant) ? 3 : BL_POLCOEF_WAVES) bl_polarized_coefficients_kernel(const BlShadeArgs P_at_entry) {
  // (with power-law or kappa electrons the kernel holds ~45 more scalars than there are registers for: it reads its arguments where it uses them)
  const BlShadeArgs &P = kThermalOnly ? P_at_entry : kernel_arguments_in_place<BlShadeArgs>();
  const unsigned long long n_records = P.counters_in[BL_CNT_RECORDS];
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long idx = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n_records; idx += stride) {
    const unsigned long long tag = reinterpret_cast<const unsigned long long *>(P.records_hot + (idx) * P.record_stride)[3];   // (ray, n)
    const uint32_t ray = (uint32_t)tag;
    if (ray == BL_DEAD_RAY) continue;
    const uint32_t n = (uint32_t)(tag >> 32);
    const BlCoefInputs ci = P.coef_inputs[idx];
    const double momentum_factor = P.ray_factor[ray];
    SampleShade sh;
    sh.have_coefficients = ci.have_coefficients != 0.0;
    sh.nu_fluid_over_nu = ci.nu_fluid_over_nu;
    sh.n_e_cgs = ci.n_e_cgs;
    sh.nu_c_cgs = ci.nu_c_cgs;
    sh.theta_e = ci.theta_e;
    sh.kb_tt_e_cgs = ci.kb_tt_e_cgs;
    // :453-455 from cos^2: the same operations the coefficient kernel applies to the same value
    sh.cos2_theta_b = ci.cos2_theta_b;
    sh.sin2_theta_b = 1.0 - ci.cos2_theta_b;
    sh.sin_theta_b = bl_sqrt_g(sh.sin2_theta_b);
    sh.cos_theta_b = bl_sqrt_g(ci.cos2_theta_b) * ci.cos_sign;
    // what does not depend on the frequency, once per sample (the reference recomputes it for every frequency)
    sh.theta_e_096 = sh.kk_0 = sh.kk_1 = sh.kk_2 = 0.0;
    if (sh.have_coefficients && P.plasma.plasma_thermal_frac != 0.0) {
      sh.theta_e_096 = kTolerant ? fastmath::pow(sh.theta_e, 0.96) : bl_pow(sh.theta_e, 0.96);
      if (sh.theta_e >= 0.01) {   // theta_e_zero, radiation_integrator.hpp:190
        if (kTolerant) fastmath::bessel_k012(1.0 / sh.theta_e, &sh.kk_0, &sh.kk_1, &sh.kk_2);
        else bl_cyl_bessel_k012(1.0 / sh.theta_e, &sh.kk_0, &sh.kk_1, &sh.kk_2);
      }
    }
    const size_t at = ((size_t)P.ray_offset[ray] + n) * P.n_nu;
    for (int l = 0; l < P.n_nu; l++) {
      const double freq = P.frequencies[l];
      double j_val = 0.0, alpha_val = 0.0;
      double2 pc[3] = {make_double2(0.0, 0.0), make_double2(0.0, 0.0), make_double2(0.0, 0.0)};
      if (kTolerant) {
        if (sh.have_coefficients) simulation_coefficients_fast<!kThermalOnly>(P, sh, freq, momentum_factor, &j_val, &alpha_val);
        polarized_coefficients_fast<!kThermalOnly>(P, sh, freq, momentum_factor, j_val, alpha_val, pc);
      } else {
        if (sh.have_coefficients) simulation_coefficients<!kThermalOnly>(P, sh, freq, momentum_factor, &j_val, &alpha_val);
        polarized_coefficients<!kThermalOnly>(P, sh, freq, momentum_factor, j_val, alpha_val, pc);
      }
#ifdef BL_POL_CONDITION_STATS
      // (a measurement build, tools/build_variant.sh -DBL_POL_CONDITION_STATS: how many samples' joint coupling step - polarized.cpp:
      // 657-778 - loses more than 2, 3, 4, 6 digits to the cancellations in lambda_1, lambda_2 = sqrt(lambda_a +- lambda_b) and in
      // 1 / (alpha_I^2 - lambda_1^2); BL_CNT_DEBUG + 0 ... 3, samples with coefficients in + 4, with rho and alpha_P both non-zero in + 5)
      if (sh.have_coefficients) {
        const double a_sq = pc[1].x * pc[1].x + pc[1].y * pc[1].y, r_sq = pc[2].x * pc[2].x + pc[2].y * pc[2].y;
        const double a_r = pc[1].x * pc[2].x + pc[1].y * pc[2].y;
        const double lb = 0.5 * (a_sq - r_sq), la = sqrt(lb * lb + a_r * a_r);
        const double l1_sq = la + lb, l2_sq = la - lb;
        double kappa = 1.0;
        if (a_sq > 0.0 && r_sq > 0.0) {
          kappa = fmax(kappa, la / fabs(l1_sq));
          kappa = fmax(kappa, la / fabs(l2_sq));
          kappa = fmax(kappa, alpha_val * alpha_val / fabs(alpha_val * alpha_val - l1_sq));
        }
        atomicAdd(&P.counters[BL_CNT_DEBUG + 4], 1ull);
        if (a_sq > 0.0 && r_sq > 0.0) atomicAdd(&P.counters[BL_CNT_DEBUG + 5], 1ull);
        if (!(kappa < 1e2)) atomicAdd(&P.counters[BL_CNT_DEBUG + 0], 1ull);
        if (!(kappa < 1e3)) atomicAdd(&P.counters[BL_CNT_DEBUG + 1], 1ull);
        if (!(kappa < 1e4)) atomicAdd(&P.counters[BL_CNT_DEBUG + 2], 1ull);
        if (!(kappa < 1e6)) atomicAdd(&P.counters[BL_CNT_DEBUG + 3], 1ull);
      }
#endif
      double2 *out = P.pol_coeffs + (at + l) * 4;
      out[0] = make_double2(j_val, alpha_val);
      out[1] = pc[0];
      out[2] = pc[1];
      out[3] = pc[2];
    }
  }
}

// Exact tier, plain images with several frequencies: one lane per (sample record, frequency). The coefficient kernel's frequency
// loop (simulation_coefficients.cpp:464-523, :556-584; unpolarized.cpp:74-110) with the loop turned into lanes: the lanes of a
// record read the same 64 bytes of inputs, evaluate the pinned formulas at their own frequency - the same operations on the
// same operands as in the loop, so the same bits - and write the record's transfer records side by side (one contiguous
// kilobyte per sample at 64 frequencies instead of 64 scattered 16-byte stores per lane), at four waves per SIMD.
template <bool kExtended>
__global__ void __launch_bounds__(256, 4) bl_coefficients_freq_kernel(const BlShadeArgs P) {
  const unsigned long long n_records = P.counters_in[BL_CNT_RECORDS];
  const unsigned long long total = n_records * (unsigned long long)P.n_nu;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const unsigned long long idx = t / (unsigned long long)P.n_nu;
    const int l = (int)(t - idx * (unsigned long long)P.n_nu);
    const unsigned long long tag = reinterpret_cast<const unsigned long long *>(P.records_hot + (idx) * P.record_stride)[3];   // (ray, n)
    const uint32_t ray = (uint32_t)tag;
    if (ray == BL_DEAD_RAY) continue;
    const uint32_t n = (uint32_t)(tag >> 32);
    const BlCoefInputs ci = P.coef_inputs[idx];
    const double momentum_factor = P.ray_factor[ray];
    SampleShade sh;
    sh.have_coefficients = ci.have_coefficients != 0.0;
    sh.nu_fluid_over_nu = ci.nu_fluid_over_nu;
    sh.n_e_cgs = ci.n_e_cgs;
    sh.nu_c_cgs = ci.nu_c_cgs;
    sh.theta_e = ci.theta_e;
    sh.kb_tt_e_cgs = ci.kb_tt_e_cgs;
    sh.sin_theta_b = ci.cos2_theta_b;
    const double delta_lambda = ci.cos_sign;
    const double freq = P.frequencies[l];
    double j_val = 0.0, alpha_val = 0.0;
    if (sh.have_coefficients) simulation_coefficients<kExtended>(P, sh, freq, momentum_factor, &j_val, &alpha_val);
    const double delta_lambda_cgs = bl_div_g(delta_lambda * P.x_unit, freq * momentum_factor);   // unpolarized.cpp:75-76
    P.transfer[((size_t)P.ray_offset[ray] + n) * P.n_nu + l] = transfer_record(j_val, alpha_val, delta_lambda_cgs);
  }
}

// The fluid frame of the samples without coefficients (polarized.cpp:163-265 at cut samples and cut or field-free cells):
// k^mu and tetrad rows 1, 2 into their BlPolSample, from what the coefficient kernel parked in BlCoefInputs.
__global__ void __launch_bounds__(256) bl_polarized_frame_kernel(const BlShadeArgs P) {
  const unsigned long long n_all = P.counters_in[BL_CNT_RECORDS];
  const unsigned long long n_listed = P.redo_list != nullptr ? P.counters_in[BL_CNT_REDO] : ~0ull;
  const bool listed = n_listed <= P.redo_capacity;   // else: more such samples than the list holds - look at every record
  const unsigned long long n_items = listed ? n_listed : n_all;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long pos = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; pos < n_items; pos += stride) {
    const unsigned long long idx = listed ? P.redo_list[pos] : pos;
    const double2 *hot = reinterpret_cast<const double2 *>(P.records_hot + (idx) * P.record_stride);
    const double2 q1 = hot[1];
    const unsigned long long tag = (unsigned long long)__double_as_longlong(q1.y);   // (ray, n)
    const uint32_t ray = (uint32_t)tag;
    if (ray == BL_DEAD_RAY) continue;
    const double2 *in = reinterpret_cast<const double2 *>(P.coef_inputs + idx);
    const double2 c3 = in[3];
    // have_coefficients: the coefficient kernel wrote the frame (where it evaluated the coefficients itself it left no inputs: a flag)
    if (P.have_flags != nullptr ? P.have_flags[idx] != 0 : c3.y != 0.0) continue;
    const uint32_t n = (uint32_t)(tag >> 32);
    const double2 q0 = hot[0], c0 = in[0], c1 = in[1], c2 = in[2];
    const double kcov[4] = {c0.x, c0.y, c1.x, c1.y};
    const float uu[3] = {__int_as_float(__double2loint(c2.x)), __int_as_float(__double2hiint(c2.x)), __int_as_float(__double2loint(c2.y))};
    const float bb[3] = {__int_as_float(__double2hiint(c2.y)), __int_as_float(__double2loint(c3.x)), __int_as_float(__double2hiint(c3.x))};
    bl_pol::sample_frame(P.st, P.plasma.simulation_coord, q0.x, q0.y, q1.x, kcov, uu, bb, P.pol_samples + ((size_t)P.ray_offset[ray] + n));
  }
}



// Diagnostics: apply one device math function element-wise (bl_debug_math). Lets the tests compare the
// device build of blmath.h and the exact-arithmetic devices of bl_geometry.h with the host, bit for bit.
__global__ void bl_debug_math_kernel(int op, long long n, const double *x, const double *y, double *out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double a = x[i], b = y != nullptr ? y[i] : 0.0;
  double r = 0.0, s_unused, c_unused;
  switch (op) {
    case 0: r = bl_exp(a); break;
    case 1: r = bl_expm1(a); break;
    case 2: r = bl_log(a); break;
    case 3: r = bl_cbrt(a); break;
    case 4: r = bl_sin(a); break;
    case 5: r = bl_cos(a); break;
    case 6: r = bl_acos(a); break;
    case 7: r = bl_atan(a); break;
    case 8: r = bl_atan2(a, b); break;
    case 9: r = bl_pow(a, b); break;
    case 10: r = bl_hypot(a, b); break;
    case 11: r = bl_hypot_g(a, b); break;
    case 12: r = bl_sqrt_g(a); break;
    case 13: r = bl_div_g(a, b); break;
    case 14: r = blm_sqrt(a); break;
    case 15: r = a / b; break;
    case 16: bl_sincos(a, &r, &c_unused); break;
    case 17: bl_sincos(a, &s_unused, &r); break;
    // tolerant tier's functions (not bit-reproducible by contract; accuracy is what the tests check)
    case 25: case 26: case 27: {
      double k0, k1, k2;
      fastmath::bessel_k012(a, &k0, &k1, &k2);
      r = op == 25 ? k0 : (op == 26 ? k1 : k2);
      break;
    }
    case 20: r = fastmath::exp(a); break;
    case 21: r = fastmath::expm1(a); break;
    case 22: r = fastmath::cbrt(a); break;
    case 23: r = fastmath::rcp(a); break;
    case 24: r = fastmath::rsqrt(a); break;
    case 28: r = fastmath::acos(a); break;
    case 29: r = fastmath::atan2(a, b); break;
    case 30: case 31: case 32: {   // the tolerant tier's Bessel functions K_0, K_1, K_2
      double k[3];
      fastmath::bessel_k012(a, &k[0], &k[1], &k[2]);
      r = k[op - 30];
      break;
    }
    case 33: case 34: case 35: {   // the exact tier's
      double k[3];
      bl_cyl_bessel_k012(a, &k[0], &k[1], &k[2]);
      r = k[op - 33];
      break;
    }
    case 36: r = fastmath::log(a); break;
    case 38: r = bl_pow_neg_fifth(a); break;
    case 37: r = fastmath::pow(a, b); break;
    default: break;
  }
  out[i] = r;
}


// frames: 1 the per-frequency coefficients, then the frames of the samples without coefficients; 0 the coefficients only; 2 the frames only
extern "C" hipError_t bl_launch_polarized_coefficients_parts(const BlShadeArgs *args, int grid, int frames, hipStream_t stream) {
  if (frames == 2) {
    hipLaunchKernelGGL(bl_polarized_frame_kernel, dim3(grid), dim3(256), 0, stream, *args);
    return hipGetLastError();
  }
  // (simulation_coefficients<kExtended> also holds the unpolarized kappa terms: never in a polarized run)
  const bool thermal_only = args->plasma.power_frac == 0.0 && args->plasma.kappa_unpolarized == 0 && args->plasma.kappa_frac_zero != 0;
#define BL_LAUNCH_PC(T, O) hipLaunchKernelGGL((bl_polarized_coefficients_kernel<T, O>), dim3(grid), dim3(256), 0, stream, *args)
  // (the exact tier's formulas in either tier: the polarized step amplifies last-place differences of the coefficients by orders of
  // magnitude in optically and Faraday thick cells - docs/notebook.md - so a tolerant kernel here moved rows with the reference's own
  // conditioning; it was a measurement switch until round 6)
  if (thermal_only) BL_LAUNCH_PC(false, true);
  else BL_LAUNCH_PC(false, false);
#undef BL_LAUNCH_PC
  if (frames == 1) hipLaunchKernelGGL(bl_polarized_frame_kernel, dim3(grid), dim3(256), 0, stream, *args);
  return hipGetLastError();
}
extern "C" hipError_t bl_launch_polarized_coefficients(const BlShadeArgs *args, int grid, hipStream_t stream) {
  return bl_launch_polarized_coefficients_parts(args, grid, 1, stream);
}

extern "C" hipError_t bl_launch_debug_math(int op, long long n, const double *x, const double *y, double *out, hipStream_t stream) {
  int grid = (int)((n + 255) / 256);
  hipLaunchKernelGGL(bl_debug_math_kernel, dim3(grid), dim3(256), 0, stream, op, n, x, y, out);
  return hipGetLastError();
}

extern "C" hipError_t bl_launch_coefficients_freq(const BlShadeArgs *args, int grid, hipStream_t stream) {
  // (the instantiation bl_launch_shade chose for the coefficient kernel: power-law electrons only in the extended one)
  const bool extended = args->plasma.power_frac != 0.0 || args->plasma.code_kappa != 0 || args->slow.n > 0 || args->anchors != nullptr
      || args->plasma.kappa_unpolarized != 0;
  (void)extended;   // (one instantiation: the extended formulas evaluate to the plain ones' bits where no power law or entropy is asked for)
  hipLaunchKernelGGL(bl_coefficients_freq_kernel<true>, dim3(grid), dim3(256), 0, stream, *args);
  return hipGetLastError();
}


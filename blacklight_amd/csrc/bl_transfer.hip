// bl_transfer.hip - the transfer stage of the hot path (gfx950).
//
//   bl_transfer_kernel       one lane per (ray, frequency): replays the per-sample records far -> near in the reference's
//                            order and scales by nu^3.   (unpolarized.cpp:71-110, :200-208)
//   bl_transfer_quad_kernel  tolerant tier, one frequency: four lanes per ray, affine maps composed by a DPP scan
//   bl_transfer_freq_kernel  tolerant tier, several frequencies: records built from per-sample factors on the fly
//   bl_transfer_aux_kernel   image_light and the nine auxiliary images.   (unpolarized.cpp:113-196)
//   bl_tau_kernel            optical depth beside the intensities in the tolerant tier
#include "bl_kernel_util.h"

#pragma clang fp contract(fast)
#include "bl_fastmath.h"

// Several frequencies in the tolerant tier: one lane per (ray, frequency) walks the ray far -> near, builds each sample's
// (a, c) from the sample's factors (BlFreqInputs; the lanes of one ray read the same 64 bytes) and the lane's own frequency,
// and applies I <- a I + c at once. The per-frequency transfer records (16 bytes per sample and frequency: 1.5 TB written
// and read per 1024^2 x 64-frequency frame) do not exist on this path; the exact second pass leaves the same factors.
__global__ void __launch_bounds__(256) bl_transfer_freq_kernel(BlTransferArgs P) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int slot = (int)(t / P.n_nu);
  const int l = (int)(t % P.n_nu);
  unsigned long long samples = 0ull, flagged = 0ull;
  int max_num = 0;
  if (slot < bl_rays_done(P.counters, P.chunk_rays)) {
    const int num = P.ray_sample_num[slot];
    const bool flag = P.ray_flags[slot] != 0;
    const long long out_index = P.ray_out_index[slot];
    if (l == 0) {
      const int all = num + (P.ray_skipped != nullptr ? P.ray_skipped[slot] : 0);
      samples = (unsigned long long)all;
      flagged = flag ? 1ull : 0ull;
      max_num = all;
      if (P.out_sample_num != nullptr) P.out_sample_num[out_index] = all;
      if (P.out_flags != nullptr) P.out_flags[out_index] = flag ? 1 : 0;
    }
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    const double f = P.frequencies[l];
    double intensity = 0.0;
    if (P.fallback_nan && flag) {
      intensity = num > 0 ? nan : 0.0;   // every sample of a flagged ray carries NaN primitives (simulation_sampling.cpp:211-216)
    } else {
      const double f_1_2 = bl_sqrt_g(f), f_1_3 = fastmath::cbrt(f);
      const double f_1_6 = bl_sqrt_g(f_1_3), f_inv = fastmath::rcp(f);
      const double f_inv2 = f_inv * f_inv;
      // (one exponential per sample and frequency is most of this loop: its constants stay in scalar registers across it)
      double exp_c[15];
#pragma unroll
      for (int i = 0; i < 15; i++) exp_c[i] = fastmath::resident_constant(fastmath::kExpConstants[i]);
      const double2 *in = reinterpret_cast<const double2 *>(P.freq_inputs + (size_t)P.ray_offset[slot]);
      const double kThird = fastmath::resident_constant(1.0 / 3.0), kInvPlanck = fastmath::resident_constant(kC * kC / (2.0 * kH));
      const double kRoot = fastmath::resident_constant(kPow2_11_12), kThin = fastmath::resident_constant(0x1p-10);
      for (int n = num - 1; n >= 0; n--) {   // reference sample order is reversed integration order (geodesics.cpp:832-840)
        const double2 q0 = in[4 * (size_t)n], q1 = in[4 * (size_t)n + 1], q2 = in[4 * (size_t)n + 2], q3 = in[4 * (size_t)n + 3];
        // bl_shade_fast_kernel's frequency loop (simulation_coefficients.cpp:464-523, unpolarized.cpp:74-110) as one straight line for
        // the sample that has coefficients, a thin step and h nu << k T_e - nearly every one: no expm1, no division, no branch; the
        // rest is selected or, where it needs an exponential of its own, redone behind a branch
        const bool have = q0.x == 1.0;
        const double xx_1_3 = q1.x * f_1_3;
        const double var_c = q0.y * f_1_2 + kRoot * (q1.y * f_1_6);
        const double j_val = q2.y * f_inv2 * fastmath::exp(-xx_1_3, exp_c) * var_c * var_c;
        const double xp = q2.x * f;
        double planck = xp * (1.0 + 0.5 * xp * (1.0 + kThird * xp * (1.0 + 0.25 * xp)));
        if (__builtin_expect(have && !(xp < kThin), 0)) planck = fastmath::expm1(xp);
        double alpha_val = j_val * (planck * kInvPlanck);
        if (alpha_val * alpha_val <= 0x1p-1024) alpha_val = 0.0;
        const double delta_lambda_cgs = q3.x * f_inv;
        const double delta_tau = alpha_val * delta_lambda_cgs;
        // optically thin step: expm1(-t) = -t p(t), p = 1 - t/2 (1 - t/3 (1 - t/4)) to 2^-53: a = 1 - t p, c = j dl p
        const double p = 1.0 - 0.5 * delta_tau * (1.0 - kThird * delta_tau * (1.0 - 0.25 * delta_tau));
        const bool absorbing = alpha_val > 0.0;
        double a = absorbing ? 1.0 - delta_tau * p : 1.0, c = j_val * delta_lambda_cgs * (absorbing ? p : 1.0);
        if (__builtin_expect(have && absorbing && !(delta_tau < kThin), 0)) {
          const double ss = j_val * fastmath::rcp(alpha_val);
          if (delta_tau <= kDeltaTauMax) {
            const double e1 = fastmath::expm1(-delta_tau);
            a = 1.0 + e1;
            c = -ss * e1;
          } else {
            a = 0.0;
            c = ss;
            intensity = 0.0;   // the thick step's intensity replaces what lies behind it, a NaN included (unpolarized.cpp:103-104)
          }
        }
        a = have ? a : 1.0;
        c = have ? c : (q0.x == 2.0 ? nan : 0.0);   // (2: NaN primitives off the grid, simulation_sampling.cpp:377-384; 0: nothing to add)
        intensity = __builtin_fma(a, intensity, c);
      }
    }
    P.image[(size_t)l * P.n_rays_total + out_index] = intensity * (f * f * f);   // unpolarized.cpp:206-207
  }
  // statistics: wave reduce, one atomic per wave
  for (int offset = 32; offset > 0; offset >>= 1) {
    samples += __shfl_xor(samples, offset, 64);
    flagged += __shfl_xor(flagged, offset, 64);
    int other = __shfl_xor(max_num, offset, 64);
    max_num = other > max_num ? other : max_num;
  }
  if ((threadIdx.x & 63) == 0) {
    if (samples) atomicAdd(&P.stats[0], samples);
    if (flagged) atomicAdd(&P.stats[1], flagged);
    atomicMax(&P.stats[2], (unsigned long long)max_num);
  }
}
#pragma clang fp contract(off)

// =================================================================================================
// Transfer kernel
// =================================================================================================
// kAffine: tolerant tier, records are (a, c) of I <- a I + c
template <bool kAffine>
__global__ void __launch_bounds__(256) bl_transfer_kernel(BlTransferArgs P) {
  // One lane per (ray, frequency): consecutive lanes are the frequencies of one ray, whose records of a
  // sample are contiguous, so multi-frequency loads coalesce and the parallelism grows with n_nu.
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int slot = (int)(t / P.n_nu);
  const int l = (int)(t % P.n_nu);
  unsigned long long samples = 0ull, flagged = 0ull;
  int max_num = 0;
  if (slot < bl_rays_done(P.counters, P.chunk_rays)) {
    int num = P.ray_sample_num[slot];
    bool flag = P.ray_flags[slot] != 0;
    long long out_index = P.ray_out_index[slot];
    if (l == 0) {
      const int all = num + (P.ray_skipped != nullptr ? P.ray_skipped[slot] : 0);   // the ray's samples, with or without a record
      samples = (unsigned long long)all;
      flagged = flag ? 1ull : 0ull;
      max_num = all;
      if (P.out_sample_num != nullptr) P.out_sample_num[out_index] = all;
      if (P.out_flags != nullptr) P.out_flags[out_index] = flag ? 1 : 0;
    }
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    {
      double intensity = 0.0;
      if (P.fallback_nan && flag) {
        // simulation: every sample of a flagged ray carries NaN primitives
        // (simulation_sampling.cpp:211-216); formula: only frequency index 0 is NaN-filled
        // (formula_coefficients.cpp:51-59 indexes a 3-D array with two indices)
        bool nan_row = P.model_type == BL_MODEL_SIMULATION || l == 0;
        intensity = (num > 0 && nan_row) ? nan : 0.0;
      } else {
        const double2 *rec = P.transfer + (size_t)P.ray_offset[slot] * P.n_nu + l;
        // reference sample order is reversed integration order (geodesics.cpp:832-840). The
        // recurrence is sequential but the loads are not: fetch 8 records (one 128-byte line of
        // this ray's row when n_nu = 1) at a time so 8 loads are in flight per lane.
        int n = num - 1;
        for (; n >= 7; n -= 8) {
          double2 ab[8];
#pragma unroll
          for (int u = 0; u < 8; u++) ab[u] = rec[(size_t)(n - u) * P.n_nu];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            if (kAffine) intensity = BL_IS_AFFINE_THICK(ab[u].x) ? ab[u].y : __builtin_fma(ab[u].x, intensity, ab[u].y);   // (the thick step, see below)
            else intensity = (ab[u].x == BL_THICK_MARK) ? ab[u].y : ab[u].x * (intensity + ab[u].y);
          }
        }
        for (; n >= 0; n--) {
          double2 ab = rec[(size_t)n * P.n_nu];
          // (a = -0.0 marks an optically thick step, whose intensity replaces what lies behind it - a NaN included,
          // unpolarized.cpp:103-104; behind a thin step a NaN stays one, BL_AFFINE_THICK in bl_device.h)
          if (kAffine) intensity = BL_IS_AFFINE_THICK(ab.x) ? ab.y : __builtin_fma(ab.x, intensity, ab.y);
          else intensity = (ab.x == BL_THICK_MARK) ? ab.y : ab.x * (intensity + ab.y);
        }
      }
      double freq = P.frequencies[l];
      double nu_cu = freq * freq * freq;   // unpolarized.cpp:206-207
      P.image[(size_t)l * P.n_rays_total + out_index] = intensity * nu_cu;
    }
  }
  // statistics: wave reduce, one atomic per wave
  for (int offset = 32; offset > 0; offset >>= 1) {
    samples += __shfl_xor(samples, offset, 64);
    flagged += __shfl_xor(flagged, offset, 64);
    int other = __shfl_xor(max_num, offset, 64);
    max_num = other > max_num ? other : max_num;
  }
  if ((threadIdx.x & 63) == 0) {
    if (samples) atomicAdd(&P.stats[0], samples);
    if (flagged) atomicAdd(&P.stats[1], flagged);
    atomicMax(&P.stats[2], (unsigned long long)max_num);
  }
}

// Auxiliary-image transfer kernel (unpolarized.cpp:53-196 for one pixel per lane): integrates image_light
// and every requested auxiliary image from the (j, alpha) pairs and the BlAuxSample records, far -> near,
// in the reference's order. Not on the benchmark path.
__global__ void __launch_bounds__(64) bl_transfer_aux_kernel(BlTransferArgs P) {
  int slot = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long samples = 0ull, flagged = 0ull;
  int max_num = 0;
  if (slot < bl_rays_done(P.counters, P.chunk_rays)) {
    const BlAuxImages &A = P.aux_images;
    const int num = P.ray_sample_num[slot];
    const bool flag = P.ray_flags[slot] != 0;
    const long long out_index = P.ray_out_index[slot];
    samples = (unsigned long long)num;
    flagged = flag ? 1ull : 0ull;
    max_num = num;
    if (P.out_sample_num != nullptr) P.out_sample_num[out_index] = num;
    if (P.out_flags != nullptr) P.out_flags[out_index] = flag ? 1 : 0;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    const double momentum_factor = P.ray_factor[slot];
    const size_t row = (size_t)P.n_rays_total;
    double *img = P.image + out_index;   // img[q * row] = image(q, pixel)
    for (int q = 0; q < A.n_q; q++) img[(size_t)q * row] = 0.0;
    const BlAuxSample *aux = P.aux + (size_t)P.ray_offset[slot];
    const double2 *ja = P.transfer + (size_t)P.ray_offset[slot] * P.n_nu * P.ja_stride;
    const bool use_j = A.image_light || A.image_emission || A.image_emission_ave;
    const bool use_alpha = A.image_light || A.image_tau || A.image_tau_int;
    if (P.render_params != nullptr) {
      // RadiationIntegrator::Render (rendering.cpp:25-179): false-colour composition along the ray from the
      // recorded cell values, far -> near
      const BlRenderDevice &R = *P.render_params;
      double rgb[BL_MAX_RENDER_IMAGES][3];
      for (int n_i = 0; n_i < BL_MAX_RENDER_IMAGES; n_i++) rgb[n_i][0] = rgb[n_i][1] = rgb[n_i][2] = 0.0;
      double previous_values[BL_NUM_CELL_VALUES];
      for (int a = 0; a < BL_NUM_CELL_VALUES; a++) previous_values[a] = nan;
      for (int n = num - 1; n >= 0; n--) {
        const BlAuxSample s = aux[n];
        const double delta_length = s.length_term;   // 0 unless a fill feature is present
        for (int n_i = 0; n_i < R.n_images; n_i++) {
          for (int n_f = 0; n_f < R.n_features[n_i]; n_f++) {
            const int n_v = R.quantity[n_i][n_f];
            const int type = R.type[n_i][n_f];
            double previous_value = 0.0, current_value = 0.0;
            for (int a = 0; a < BL_NUM_CELL_VALUES; a++)   // register arrays: select instead of indexing
              if (a == n_v) {
                previous_value = previous_values[a];
                current_value = s.cell[a];
              }
            const double cx = R.xyz[n_i][n_f][0], cy = R.xyz[n_i][n_f][1], cz = R.xyz[n_i][n_f][2];
            if (type == BL_RENDER_FILL && current_value >= R.min_val[n_i][n_f] && current_value <= R.max_val[n_i][n_f]) {
              const double delta_tau = delta_length / R.tau_scale[n_i][n_f];
              if (delta_tau <= kDeltaTauMax) {
                const double exp_neg = bl_exp(-delta_tau);
                const double expm1 = bl_expm1(delta_tau);
                rgb[n_i][0] = exp_neg * (rgb[n_i][0] + cx * expm1);
                rgb[n_i][1] = exp_neg * (rgb[n_i][1] + cy * expm1);
                rgb[n_i][2] = exp_neg * (rgb[n_i][2] + cz * expm1);
              } else {
                rgb[n_i][0] = cx;
                rgb[n_i][1] = cy;
                rgb[n_i][2] = cz;
              }
            }
            bool threshold_crossed = false;
            const bool rise_search = type == BL_RENDER_THRESH || type == BL_RENDER_RISE;
            if (rise_search && previous_value < R.thresh[n_i][n_f] && current_value >= R.thresh[n_i][n_f]) threshold_crossed = true;
            const bool fall_search = type == BL_RENDER_THRESH || type == BL_RENDER_FALL;
            if (fall_search && previous_value > R.thresh[n_i][n_f] && current_value <= R.thresh[n_i][n_f]) threshold_crossed = true;
            if (threshold_crossed) {
              const double opacity = R.opacity[n_i][n_f];
              rgb[n_i][0] = (1.0 - opacity) * rgb[n_i][0] + opacity * cx;
              rgb[n_i][1] = (1.0 - opacity) * rgb[n_i][1] + opacity * cy;
              rgb[n_i][2] = (1.0 - opacity) * rgb[n_i][2] + opacity * cz;
            }
          }
        }
        for (int a = 0; a < BL_NUM_CELL_VALUES; a++) previous_values[a] = s.cell[a];
      }
      for (int n_i = 0; n_i < R.n_images; n_i++)
        for (int c = 0; c < 3; c++) P.render[(size_t)(n_i * 3 + c) * row + out_index] = rgb[n_i][c];
    }
    for (int l = 0; l < ((A.n_q > 0 && !A.polarized_rows_only) ? P.n_nu : 0); l++) {
      const double freq = P.frequencies[l];
      double intensity = 0.0, integrated_lambda = 0.0, integrated_emission = 0.0, tau = 0.0;
      double time_min = 0.0, length = 0.0;
      double lambda_ave[BL_NUM_CELL_VALUES], emission_ave[BL_NUM_CELL_VALUES], tau_int[BL_NUM_CELL_VALUES];
      for (int a = 0; a < BL_NUM_CELL_VALUES; a++) lambda_ave[a] = emission_ave[a] = tau_int[a] = 0.0;
      bool plane_sign = num > 0 ? aux[num - 1].plane > 0.0 : false;   // first (farthest) sample, :63-67
      int crossings = 0;
      // reference sample order is reversed integration order (geodesics.cpp:832-840)
      for (int n = num - 1; n >= 0; n--) {
        const BlAuxSample s = aux[n];
        const double2 c = ja[((size_t)n * P.n_nu + l) * P.ja_stride];
        const double delta_lambda = s.delta_lambda;
        const double delta_lambda_cgs = delta_lambda * P.x_unit / (freq * momentum_factor);
        const double t_cgs = s.t * P.t_unit;
        const double j = use_j ? c.x : nan;
        const double alpha = use_alpha ? c.y : nan;
        const double ss = j / alpha;
        const double delta_tau = alpha * delta_lambda_cgs;
        // (consumers: the intensity of an unpolarized run and tau_int; a polarized run integrates its Stokes rows elsewhere)
        const bool need_exp = (A.image_light && !A.polarized) || A.image_tau_int;
        const double exp_neg = need_exp ? bl_exp(-delta_tau) : 0.0;
        const double expm1 = need_exp ? bl_expm1(delta_tau) : 0.0;
        const bool optically_thin = delta_tau <= kDeltaTauMax;
        if (A.image_light && !A.polarized) {
          if (alpha > 0.0) {
            if (optically_thin)
              intensity = exp_neg * (intensity + ss * expm1);
            else
              intensity = ss;
          } else {
            intensity += j * delta_lambda_cgs;
          }
        }
        if (A.image_time && l == 0) time_min = std_min(time_min, t_cgs);
        if (A.image_length && l == 0) length += s.length_term;
        if (A.image_lambda || A.image_lambda_ave) integrated_lambda += delta_lambda_cgs;
        if (A.image_emission || A.image_emission_ave) integrated_emission += j * delta_lambda_cgs;
        if (A.image_tau) tau += delta_tau;
        const bool have_cell = !(s.cell[0] != s.cell[0]);
        if (A.image_lambda_ave && have_cell)
          for (int a = 0; a < BL_NUM_CELL_VALUES; a++) lambda_ave[a] += s.cell[a] * delta_lambda_cgs;
        if (A.image_emission_ave && have_cell)
          for (int a = 0; a < BL_NUM_CELL_VALUES; a++) emission_ave[a] += s.cell[a] * j * delta_lambda_cgs;
        if (A.image_tau_int && have_cell) {
          if (optically_thin)
            for (int a = 0; a < BL_NUM_CELL_VALUES; a++) tau_int[a] = exp_neg * (tau_int[a] + s.cell[a] * expm1);
          else
            for (int a = 0; a < BL_NUM_CELL_VALUES; a++) tau_int[a] = s.cell[a];
        }
        if (A.image_crossings && l == 0) {
          const bool plane_sign_new = s.plane > 0.0;
          if (plane_sign_new != plane_sign) crossings++;
          plane_sign = plane_sign_new;
        }
      }
      if (A.image_light && !A.polarized) img[(size_t)l * row] = intensity * (freq * freq * freq);   // :200-208
      if (A.image_time && l == 0) img[(size_t)A.offset_time * row] = time_min;
      if (A.image_length && l == 0) img[(size_t)A.offset_length * row] = length;
      if (A.image_lambda) img[(size_t)(A.offset_lambda + l) * row] = integrated_lambda;
      if (A.image_emission) img[(size_t)(A.offset_emission + l) * row] = integrated_emission;
      if (A.image_tau) img[(size_t)(A.offset_tau + l) * row] = tau;
      if (A.image_crossings && l == 0) img[(size_t)A.offset_crossings * row] = (double)crossings;
      for (int a = 0; a < BL_NUM_CELL_VALUES; a++) {
        if (A.image_lambda_ave) img[(size_t)(A.offset_lambda_ave + l * BL_NUM_CELL_VALUES + a) * row] = lambda_ave[a] / integrated_lambda;
        if (A.image_emission_ave) img[(size_t)(A.offset_emission_ave + l * BL_NUM_CELL_VALUES + a) * row] = emission_ave[a] / integrated_emission;
        if (A.image_tau_int) img[(size_t)(A.offset_tau_int + l * BL_NUM_CELL_VALUES + a) * row] = tau_int[a];
      }
    }
  }
  for (int offset = 32; offset > 0; offset >>= 1) {
    samples += __shfl_xor(samples, offset, 64);
    flagged += __shfl_xor(flagged, offset, 64);
    int other = __shfl_xor(max_num, offset, 64);
    max_num = other > max_num ? other : max_num;
  }
  if ((threadIdx.x & 63) == 0) {
    if (samples) atomicAdd(&P.stats[0], samples);
    if (flagged) atomicAdd(&P.stats[1], flagged);
    atomicMax(&P.stats[2], (unsigned long long)max_num);
  }
}

extern "C" hipError_t bl_launch_transfer_freq(const BlTransferArgs *args, hipStream_t stream) {
  const long long lanes = (long long)args->chunk_rays * args->n_nu;
  hipLaunchKernelGGL(bl_transfer_freq_kernel, dim3((unsigned int)((lanes + 255) / 256)), dim3(256), 0, stream, *args);
  return hipGetLastError();
}

extern "C" hipError_t bl_launch_transfer_aux(const BlTransferArgs *args, hipStream_t stream) {
  int grid = (args->chunk_rays + 63) / 64;
  hipLaunchKernelGGL(bl_transfer_aux_kernel, dim3(grid), dim3(64), 0, stream, *args);
  return hipGetLastError();
}

// Optical depth beside the intensities in the tolerant tier: tau(ray, frequency) = sum of alpha x length over the ray's samples, far
// -> near as the reference adds them (unpolarized.cpp:63-151), from the increments the fast coefficient kernel left
__global__ void __launch_bounds__(256) bl_tau_kernel(BlTransferArgs P) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int slot = (int)(t / P.n_nu);
  const int l = (int)(t % P.n_nu);
  if (slot >= bl_rays_done(P.counters, P.chunk_rays)) return;
  const int num = P.ray_sample_num[slot];
  const double *inc = P.tau_inc + (size_t)P.ray_offset[slot] * P.n_nu + l;
  double tau = 0.0;
  int n = num - 1;
  if (P.fallback_nan && P.ray_flags[slot] != 0) {   // every sample of a flagged ray carries NaN primitives (simulation_sampling.cpp:211-216)
    tau = num > 0 ? __longlong_as_double(0x7ff8000000000000ll) : 0.0;
    n = -1;
  }
  for (; n >= 7; n -= 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = inc[(size_t)(n - u) * P.n_nu];
#pragma unroll
    for (int u = 0; u < 8; u++) tau += v[u];
  }
  for (; n >= 0; n--) tau += inc[(size_t)n * P.n_nu];
  P.image[(size_t)(P.tau_row + l) * P.n_rays_total + P.ray_out_index[slot]] = tau;
}

extern "C" hipError_t bl_launch_tau(const BlTransferArgs *args, hipStream_t stream) {
  const int grid = (int)(((long long)args->chunk_rays * args->n_nu + 255) / 256);
  hipLaunchKernelGGL(bl_tau_kernel, dim3(grid), dim3(256), 0, stream, *args);
  return hipGetLastError();
}

// Tolerant tier, one frequency: four lanes per ray. I <- a I + c is an affine map and maps compose - (a2, c2) after (a1, c1) is
// (a2 a1, a2 c1 + c2) - so the four lanes of a quad take four consecutive records (64 contiguous bytes where a lane per ray reads
// 16 from each of 64 different lines), compose them in order by two steps of a scan inside the quad, and the quad's map is applied to
// the running intensity. Same records, same order of application; the association differs (rounding level: the tier's tolerance).
// (Measured on the benchmark frame: 2.4 ms against the lane-per-ray kernel's 2.9; two lanes per ray 2.8, eight 3.6, sixteen 6.7 - beyond
// a quad the scan's moves go through the LDS crossbar instead of DPP; 4, 8 or 16 records per lane in flight make no difference.)
__global__ void __launch_bounds__(256) bl_transfer_quad_kernel(BlTransferArgs P) {
  constexpr int kLanes = 4, kShift = 2;
  constexpr int kBatch = 8;   // records per lane in flight
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int slot = (int)(t >> kShift);
  const int q = (int)(t & (kLanes - 1));
  unsigned long long samples = 0ull, flagged = 0ull;
  int max_num = 0;
  if (slot < bl_rays_done(P.counters, P.chunk_rays)) {
    const int num = P.ray_sample_num[slot];
    const bool flag = P.ray_flags[slot] != 0;
    const long long out_index = P.ray_out_index[slot];
    const int all = num + (P.ray_skipped != nullptr ? P.ray_skipped[slot] : 0);
    if (q == 0) {
      samples = (unsigned long long)all;
      flagged = flag ? 1ull : 0ull;
      max_num = all;
      if (P.out_sample_num != nullptr) P.out_sample_num[out_index] = all;
      if (P.out_flags != nullptr) P.out_flags[out_index] = flag ? 1 : 0;
    }
    double intensity = 0.0;
    if (P.fallback_nan && flag) {
      intensity = num > 0 ? __longlong_as_double(0x7ff8000000000000ll) : 0.0;   // simulation_sampling.cpp:211-216
    } else {
      const double2 *rec = P.transfer + (size_t)P.ray_offset[slot];
      for (int top = num - 1; top >= 0; top -= kLanes * kBatch) {   // far -> near (geodesics.cpp:832-840)
        double2 m[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; u++) {
          const int n = top - kLanes * u - q;
          m[u] = n >= 0 ? rec[n] : make_double2(1.0, 0.0);
        }
#pragma unroll
        for (int u = 0; u < kBatch; u++) {
          double a = m[u].x, c = m[u].y;
          // a = -0.0 marks an optically thick step, whose intensity replaces whatever lies behind it - a NaN included (unpolarized.cpp:
          // 103-104, BL_AFFINE_THICK; a thin step's a may round to +0 and a product of them underflow, behind which a NaN stays a
          // NaN). A segment that holds a thick step keeps its map; the sign of a product carries the mark of what a segment without
          // one is composed with (every other a is positive or +0), so the sign bit of a is the flag at every stage of the scan.
#pragma unroll
          for (int d = 1; d < kLanes; d <<= 1) {
            const double pa = __shfl_up(a, d, kLanes), pc = __shfl_up(c, d, kLanes);   // the map of the d records before this lane's segment
            if (q >= d && !BL_IS_AFFINE_THICK(a)) {
              c = __builtin_fma(a, pc, c);
              a *= pa;
            }
          }
          const double block_a = __shfl(a, kLanes - 1, kLanes), block_c = __shfl(c, kLanes - 1, kLanes);
          intensity = BL_IS_AFFINE_THICK(block_a) ? block_c : __builtin_fma(block_a, intensity, block_c);
        }
      }
    }
    if (q == 0) {
      const double freq = P.frequencies[0];
      P.image[out_index] = intensity * (freq * freq * freq);   // unpolarized.cpp:206-207
    }
  }
  for (int offset = 32; offset > 0; offset >>= 1) {
    samples += __shfl_xor(samples, offset, 64);
    flagged += __shfl_xor(flagged, offset, 64);
    const int other = __shfl_xor(max_num, offset, 64);
    max_num = other > max_num ? other : max_num;
  }
  if ((threadIdx.x & 63) == 0) {
    if (samples) atomicAdd(&P.stats[0], samples);
    if (flagged) atomicAdd(&P.stats[1], flagged);
    atomicMax(&P.stats[2], (unsigned long long)max_num);
  }
}

// Tolerant tier, composed maps (BlShadeArgs::composed; one frequency): one lane per ray applies the ray's segment maps far -> near -
// about a seventh of the per-sample records, already composed by the coefficient kernel. A row that stands for per-sample
// records (BL_COMPOSED_EXPANDED: its wave held a sample for the exact kernel or a thick step) is replayed from those.
__global__ void __launch_bounds__(256) bl_transfer_composed_kernel(BlTransferArgs P) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long samples = 0ull, flagged = 0ull;
  int max_num = 0;
  if (slot < bl_rays_done(P.counters, P.chunk_rays)) {
    const int num = P.ray_sample_num[slot];
    const bool flag = P.ray_flags[slot] != 0;
    const long long out_index = P.ray_out_index[slot];
    const int all = num + (P.ray_skipped != nullptr ? P.ray_skipped[slot] : 0);   // the ray's samples, with or without a record
    samples = (unsigned long long)all;
    flagged = flag ? 1ull : 0ull;
    max_num = all;
    if (P.out_sample_num != nullptr) P.out_sample_num[out_index] = all;
    if (P.out_flags != nullptr) P.out_flags[out_index] = flag ? 1 : 0;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    double intensity = 0.0;
    if (P.fallback_nan && flag) {
      intensity = num > 0 ? nan : 0.0;   // every sample of a flagged ray carries NaN primitives (simulation_sampling.cpp:211-216)
    } else {
      const double2 *rows = P.composed + (size_t)P.ray_offset[slot];
      auto apply = [&](const double2 map) {
        if (__builtin_expect(BL_COMPOSED_EXPANDED(map.x), 0)) {
          const int length = (int)(-map.x);
          const double2 *rec = P.transfer + (size_t)(unsigned long long)__double_as_longlong(map.y);
          for (int i = length - 1; i >= 0; i--) {
            const double2 ac = rec[i];
            intensity = BL_IS_AFFINE_THICK(ac.x) ? ac.y : __builtin_fma(ac.x, intensity, ac.y);   // (unpolarized.cpp:103-104: see bl_transfer_kernel)
          }
        } else {
          intensity = BL_IS_AFFINE_THICK(map.x) ? map.y : __builtin_fma(map.x, intensity, map.y);
        }
      };
      int s = P.ray_rows[slot] - 1;
      for (; s >= 7; s -= 8) {   // eight loads in flight per lane
        double2 maps[8];
#pragma unroll
        for (int u = 0; u < 8; u++) maps[u] = rows[s - u];
#pragma unroll
        for (int u = 0; u < 8; u++) apply(maps[u]);
      }
      for (; s >= 0; s--) apply(rows[s]);
    }
    const double freq = P.frequencies[0];
    P.image[out_index] = intensity * (freq * freq * freq);   // unpolarized.cpp:206-207
  }
  for (int offset = 32; offset > 0; offset >>= 1) {
    samples += __shfl_xor(samples, offset, 64);
    flagged += __shfl_xor(flagged, offset, 64);
    int other = __shfl_xor(max_num, offset, 64);
    max_num = other > max_num ? other : max_num;
  }
  if ((threadIdx.x & 63) == 0) {
    if (samples) atomicAdd(&P.stats[0], samples);
    if (flagged) atomicAdd(&P.stats[1], flagged);
    atomicMax(&P.stats[2], (unsigned long long)max_num);
  }
}

extern "C" hipError_t bl_launch_transfer_composed(const BlTransferArgs *args, hipStream_t stream) {
  hipLaunchKernelGGL(bl_transfer_composed_kernel, dim3((args->chunk_rays + 255) / 256), dim3(256), 0, stream, *args);
  return hipGetLastError();
}

extern "C" hipError_t bl_launch_transfer(const BlTransferArgs *args, hipStream_t stream) {
  int grid = (int)(((long long)args->chunk_rays * args->n_nu + 255) / 256);
  if (args->affine && args->n_nu == 1 && !args->lane_transfer) {
    hipLaunchKernelGGL(bl_transfer_quad_kernel, dim3((int)(((long long)args->chunk_rays * 4 + 255) / 256)), dim3(256), 0, stream, *args);
    return hipGetLastError();
  }
  if (args->affine) hipLaunchKernelGGL(bl_transfer_kernel<true>, dim3(grid), dim3(256), 0, stream, *args);
  else hipLaunchKernelGGL(bl_transfer_kernel<false>, dim3(grid), dim3(256), 0, stream, *args);
  return hipGetLastError();
}

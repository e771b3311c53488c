// bl_shade.hip - the exact arithmetic tier's coefficient kernels (gfx950; the locate kernels in front of them: bl_locate.hip).
//
//   bl_shade_kernel      one SAMPLE per lane, 2 waves per SIMD ("coefficient kernel"): the 8-variable trilinear read from the
//   bl_shade_exact_kernel    interleaved grid, per-sample momentum renormalisation, thermal-synchrotron j_nu / alpha_nu (or the
//                        formula model), and the per-sample transfer coefficients (a, b) of I <- a (I + b).
//                        (simulation_sampling.cpp:666-1033, simulation_coefficients.cpp:253-524,
//                        formula_coefficients.cpp:62-180, unpolarized.cpp:74-110)
//   auxiliary images     bl_shade_kernel<., true> writes (j, alpha) and a BlAuxSample per sample.   (unpolarized.cpp:113-196)
#include "bl_sampling.h"

// ---- coefficient kernel: one sample record per lane, pure fp64 arithmetic between one coalesced
// read (record + located sample) and one 16-byte store per frequency. Two waves per SIMD so that one
// wave's scalar work, dependent-issue bubbles and load waits overlap the other's VALU work.
// kAux (any auxiliary image requested, unpolarized.cpp:113-173): the per-frequency pairs written are
// (j_nu, alpha_nu) instead of (a, b), and one BlAuxSample per sample goes with them; the auxiliary
// transfer kernel integrates everything. Kept out of the instantiations the benchmark path runs.
// kExtended: power-law electrons present (simulation_coefficients.cpp:556-584: two more pow() per
// sample and frequency) or plasma_model = code_kappa (:351-358: a ninth grid value per cell); its own
// instantiation so that the thermal-only T_i/T_e(beta) kernel keeps its registers.
// kSpinZero: bh_a == 0.0 known at compile time (the benchmark's instantiations only; bl_geometry.h "zero spin").
// kRedo: second pass of the tolerant tier - shades only the records the tolerant kernel listed (cut decisions
// inside its guard band), or every record when the list overflowed, and writes (a, c) records like that kernel.
template <int kModel, bool kAux, bool kExtended, bool kSksCurved, bool kPolarized, bool kSpinZero, bool kRedo = false>
__global__ void __launch_bounds__(256, 2) bl_shade_kernel(const BlShadeArgs P_at_entry) {
  // (The second pass of the tolerant tier - a few listed records, 170 ... 241 scalar registers spilled with its arguments held from
  // the entry on - reads its arguments where it uses them: bl_kernel_util.h. The first-pass instantiations keep theirs in registers:
  // the extended one spills ~100 of them to vector lanes and is still 8 - 10 % faster that way than with a scalar load per use,
  // measured on the refined-mesh, block-interpolation and slow-light frames of bench.py --workload, profiles/r05_f_rows.txt.)
  const BlShadeArgs &P = kRedo ? kernel_arguments_in_place<BlShadeArgs>() : P_at_entry;
  const BlSpacetime st = P.st;
  // (inter-block interpolation behind bl_shade_fused2_kernel<..., kRefined>: the eight anchor cells of a sample located here - in LDS, a
  // row per lane: FindNearbyInds' loop over the corners indexes them, which registers cannot take without scratch memory)
  __shared__ unsigned int anchor_rows[(kRedo && kExtended) ? 256 * 8 : 1];
  unsigned int *anchors_here = (kRedo && kExtended) ? anchor_rows + threadIdx.x * 8 : nullptr;
  // ... and the mesh's tables for locating those samples, in LDS where bl_launch_shade_redo gave the room (BL_REDO_TABLES_LDS; else in HBM)
  extern __shared__ double redo_tables[];
  RefinedTables refined = refined_tables_in_hbm(P.grid);
  if (kRedo && kExtended && kModel == BL_MODEL_SIMULATION && P.located == nullptr && P.grid.n_blocks > 0 && P.grid.refined_lds_bytes > 0
      && P.grid.refined_lds_bytes <= BL_REDO_TABLES_LDS) {
    stage_refined_tables(P.grid, redo_tables, &refined);
    __syncthreads();
  }
  const unsigned long long first_record = 0ull;
  const unsigned long long n_all = P.counters_in[BL_CNT_RECORDS];
  const unsigned long long n_listed = kRedo ? P.counters_in[BL_CNT_REDO] : 0ull;
  const bool listed = kRedo && n_listed <= P.redo_capacity;
  const unsigned long long n_records = listed ? n_listed : n_all;   // work items: list entries or records
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  // The record and the located sample of the next iteration are requested at the top of this one, behind
  // this sample's grid reads, so they arrive while the arithmetic runs.
  unsigned long long pos = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= n_records) return;
  unsigned long long idx = listed ? P.redo_list[pos] : first_record + pos;
  double2 nq0, nq1, nq2, nq3, nl0 = make_double2(0.0, 0.0), nl1 = nl0;
  unsigned long long ntag = 0ull;
  {
    const double2 *hot = reinterpret_cast<const double2 *>(P.records_hot + (idx) * P.record_stride);
    const double2 *cold = reinterpret_cast<const double2 *>(P.records_cold + (idx) * P.record_stride);
    nq0 = hot[0]; nq1 = hot[1]; nq2 = cold[0]; nq3 = cold[1];
    if (kModel == BL_MODEL_SIMULATION && !(kRedo && P.located == nullptr)) {
      const double2 *loc = reinterpret_cast<const double2 *>(P.located + idx);
      nl0 = loc[0]; nl1 = loc[1];
      ntag = kRedo ? (unsigned long long)__double_as_longlong(nl1.y) : P.located_tag[idx];
    }
  }
  for (bool more = true; more;) {
    const unsigned long long idx_cur = idx;
    const double2 q0 = nq0, q1 = nq1, q2 = nq2, q3 = nq3;
    double2 l0 = nl0, l1 = nl1;
    unsigned long long tag = ntag;
    const uint32_t ray = (uint32_t)__double_as_longlong(q1.y);
    const bool live = ray != BL_DEAD_RAY;
    const uint32_t n = (uint32_t)(((unsigned long long)__double_as_longlong(q1.y)) >> 32);
    const double x1 = q0.x, x2 = q0.y, x3 = q1.x;
    if (kRedo && kModel == BL_MODEL_SIMULATION && P.located == nullptr && live) {
      // second pass behind bl_shade_fused2_kernel, which leaves no located samples: the few samples it deferred are located
      // here, by the locate kernel's own code on the coordinate tables where they lie in HBM (the grid read was counted there)
      GridTables tab;
      for (int a = 0; a < 3; a++) {
        tab.xf[a] = P.grid.xf[a];
        tab.xv[a] = P.grid.xv[a];
        tab.bucket[a] = P.grid.bucket[a];
      }
      double r2_unused;
      const double r_here = bl_radial_coordinate2<kSpinZero>(st, x1, x2, x3, &r2_unused);
      LocatedSample loc;
      loc.f_i = loc.f_j = loc.f_k = loc.ph = 0.0;
      loc.cell = 0u;
      loc.status = kSampleCut;
      unsigned long long counted_already = 0ull;
      // (a mesh with refinement behind bl_shade_fused2_kernel<..., kRefined>: the refined search on its tables in HBM)
      if (!(r_here > P.cuts.camera_r)) {
        if (P.grid.n_blocks > 0) locate_sample<true, kSpinZero, kTableAnywhere>(P, tab, st, x1, x2, x3, r_here, &loc, &counted_already, kExtended ? anchors_here : nullptr, &refined);
        else locate_sample<false, kSpinZero>(P, tab, st, x1, x2, x3, r_here, &loc, &counted_already, nullptr);
      }
      l0 = make_double2(loc.f_i, loc.f_j);
      l1 = make_double2(loc.f_k, 0.0);
      tag = ((unsigned long long)loc.status << 32) | loc.cell;
    }
    const double delta_lambda = -q3.y;   // ReverseGeodesics: sample_len = -geodesic_len (:840)
    double kcov[4] = {0.0, q2.x, q2.y, q3.x};
    double momentum_factor = 0.0;
    size_t row = 0;   // of this sample in the per-sample arrays: the ray's first row + n
    if (live) {
      kcov[0] = P.ray_kt[ray];
      momentum_factor = P.ray_factor[ray];
      row = (size_t)P.ray_offset[ray] + n;
      if (kRedo && P.composed != nullptr) row = (size_t)idx_cur;   // composed transfer maps: per-sample records lie by record index
    }
    float pr[8];
    float kappa_f = 0.0f;
    double ph = 0.0;
    int status = kSampleNone;
    if (kModel == BL_MODEL_SIMULATION && live) {
      ph = l1.y;
      status = kExtended ? ((int)(tag >> 32) & 0xff) : (int)(tag >> 32);   // bits 40..: time slice (slow light only)
      if (kExtended && P.slow.n > 0) {
        sample_primitives_slow(P, status, (uint32_t)tag, P.anchors != nullptr ? P.anchors + idx_cur * 8 : nullptr, (int)(tag >> 40),
                               P.slow.frac[idx_cur], l0.x, l0.y, l1.x, pr, &kappa_f);
      } else if (kExtended && status == kSampleAdvanced) {
        sample_primitives_advanced(P, (kRedo && P.located == nullptr) ? anchors_here : P.anchors + idx_cur * 8, l0.x, l0.y, l1.x, pr, &kappa_f);
      } else {
        sample_primitives(P, status, (uint32_t)tag, l0.x, l0.y, l1.x, pr);
        if (kExtended && P.plasma.code_kappa) kappa_f = sample_kappa(P, status, (uint32_t)tag, l0.x, l0.y, l1.x);
      }
    }
    pos += stride;
    more = pos < n_records;
    if (more) {
      idx = listed ? P.redo_list[pos] : first_record + pos;
      const double2 *hot = reinterpret_cast<const double2 *>(P.records_hot + (idx) * P.record_stride);
      const double2 *cold = reinterpret_cast<const double2 *>(P.records_cold + (idx) * P.record_stride);
      nq0 = hot[0]; nq1 = hot[1]; nq2 = cold[0]; nq3 = cold[1];
      if (kModel == BL_MODEL_SIMULATION && !(kRedo && P.located == nullptr)) {
        const double2 *loc = reinterpret_cast<const double2 *>(P.located + idx);
        nl0 = loc[0]; nl1 = loc[1];
        ntag = kRedo ? (unsigned long long)__double_as_longlong(nl1.y) : P.located_tag[idx];
      }
    }
    if (!live) continue;
    // Kerr-Schild scalars at the sample: evaluated once, shared by the renormalisation, the cuts, the
    // simulation metric and the geodesic metric (the reference recomputes them in each of those
    // functions; identical inputs, identical bits)
    BlKerrSchild ks;
    bl_kerr_schild<kSpinZero>(st, x1, x2, x3, &ks);   // r^2 as the locate kernel computed it: same operations, same bits
    // second pass of the tolerant tier: the record carries the tag where the azimuth would be; the azimuth as the locate
    // kernel computes it (locate_sample: same functions of the same x, y, r - same bits)
    if (kRedo && kModel == BL_MODEL_SIMULATION) ph = kSpinZero ? bl_atan2(x2, x1) : bl_atan2(x2, x1) - bl_atan(st.bh_a / ks.r);
    if (kModel == BL_MODEL_FORMULA) {
      bool skip = ks.r > P.cuts.camera_r;                              // formula_coefficients.cpp:78-116
      if (!skip && P.cuts.any_optional) skip = optional_cuts(*P.cold, x1, x2, x3, ks.r);
      status = skip ? kSampleCut : kSampleFormula;
      for (int v = 0; v < 8; v++) pr[v] = 0.0f;
    }
    if (!P.samples_renormalised) {   // per-sample renormalisation of the stored momentum (geodesics.cpp:352-371); samples
      double gcon[4][4];             // loaded from a geodesic checkpoint carry the renormalised momentum already
      if (!kSksCurved && st.ray_flat)
        bl_minkowski(gcon);
      else
        bl_gcon_ks(ks, gcon);
      double factor = bl_renormalization_factor_g(gcon, kcov[0], kcov[1], kcov[2], kcov[3]);
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
    bool nan_ray = false;
    if (kAux) {
      // rays that ended on ray_max_steps / retries with fallback_nan: simulation mode samples NaN
      // primitives at every sample, cuts not applied (simulation_sampling.cpp:211-216); formula mode
      // fills frequency 0 with NaN coefficients (formula_coefficients.cpp:51-59)
      nan_ray = P.plasma.fallback_nan && P.ray_flags[ray] != 0;
      if (nan_ray && kModel == BL_MODEL_SIMULATION) {
        const float fnan = __int_as_float(0x7fc00000);
        for (int v = 0; v < 8; v++) pr[v] = fnan;
        kappa_f = fnan;
        status = kSampleOffGrid;
      }
    }
    if (status != kSampleCut) {
      if (kModel == BL_MODEL_SIMULATION)
        sample_finish_simulation<kExtended, kSksCurved>(P, st, ks, x3 / ks.r, ph, pr, kappa_f, kcov,
                                            kAux ? P.aux_need_coefficients : 1, &sh,
                                            kPolarized ? P.pol_samples + row : nullptr);
      else if (!(kAux && nan_ray))
        shade_formula(P, st, ks.r, x1, x2, x3, &sh);
    }
    double2 *out = P.transfer + row * P.n_nu;
    if (kAux && !(kPolarized && P.aux_record_unused)) write_aux_record(P, st, ks, idx_cur, row, sh, kcov, x1, x2, x3, delta_lambda);
    if (kPolarized) {
      write_polarized_inputs(P, idx_cur, row, sh, kcov, pr, x1, x2, x3, delta_lambda);
      continue;
    }
    if (!kAux && !kPolarized && !kRedo && kModel == BL_MODEL_SIMULATION && P.coef_split) {
      // exact tier with several frequencies: the per-frequency formulas and transfer records are bl_coefficients_freq_kernel's,
      // one lane per (record, frequency); it gets the six numbers they need, the sample's length in the sign's slot
      BlCoefInputs ci;
      ci.nu_fluid_over_nu = sh.nu_fluid_over_nu;
      ci.n_e_cgs = sh.n_e_cgs;
      ci.nu_c_cgs = sh.nu_c_cgs;
      ci.theta_e = sh.theta_e;
      ci.kb_tt_e_cgs = sh.kb_tt_e_cgs;
      ci.cos2_theta_b = sh.sin_theta_b;   // (sin theta_B itself: nothing here needs the cosine)
      ci.cos_sign = delta_lambda;
      ci.have_coefficients = sh.have_coefficients ? 1.0 : 0.0;
      P.coef_inputs[idx_cur] = ci;
      continue;
    }
    if (kRedo && kModel == BL_MODEL_SIMULATION && P.freq_split) {
      // second pass of the tolerant tier with several frequencies: the cut decisions above were the point; what goes to
      // bl_transfer_freq_kernel are the same per-sample factors the fast kernel leaves (BlFreqInputs), from this sample's
      // exactly computed state - thermal electrons only, as everywhere in that tier
      double2 *dst = reinterpret_cast<double2 *>(P.freq_inputs + row);
      if (!sh.have_coefficients) {
        dst[0] = make_double2(0.0, 0.0);
      } else {
        const double nu_s_cgs = 2.0 / 9.0 * sh.nu_c_cgs * sh.theta_e * sh.theta_e * sh.sin_theta_b;
        const double s_nu = sh.nu_fluid_over_nu * momentum_factor;
        const double s_x = s_nu / nu_s_cgs;
        const double s_1_3 = bl_cbrt(s_x);
        dst[0] = make_double2(1.0, bl_sqrt_g(s_x));
        dst[1] = make_double2(s_1_3, bl_sqrt_g(s_1_3));
        dst[2] = make_double2(kH * s_nu / sh.kb_tt_e_cgs, P.plasma.plasma_thermal_frac * sh.n_e_cgs * kE * kE * sh.nu_c_cgs * (1.0 / kC)
                                  * (kSqrt2 * kPi / 27.0) * sh.sin_theta_b / (s_nu * s_nu));
        dst[3] = make_double2(delta_lambda * P.x_unit / momentum_factor, 0.0);
      }
      continue;
    }
    // ---------------- per-frequency coefficients and transfer records
    for (int l = 0; l < P.n_nu; l++) {
      const double freq = P.frequencies[l];
      double j_val = 0.0, alpha_val = 0.0;
      if (sh.have_coefficients && kModel == BL_MODEL_SIMULATION) {
        simulation_coefficients<kExtended>(P, sh, freq, momentum_factor, &j_val, &alpha_val);
      } else if (sh.have_coefficients && kModel == BL_MODEL_FORMULA) {
        // formula_coefficients.cpp:164-179
        const BlFormulaDevice &fm = P.formula;
        const double nu_fluid_cgs = -(sh.fu[0] * kcov[0] + sh.fu[1] * kcov[1] + sh.fu[2] * kcov[2] + sh.fu[3] * kcov[3]) * freq * momentum_factor;
        const double j_nu_fluid_cgs = fm.cn0 * sh.n_n0_fluid * bl_pow(nu_fluid_cgs / fm.nup, -fm.alpha);
        j_val = j_nu_fluid_cgs / (nu_fluid_cgs * nu_fluid_cgs);
        const double alpha_nu_fluid_cgs = fm.a * fm.cn0 * sh.n_n0_fluid * bl_pow(nu_fluid_cgs / fm.nup, -fm.beta - fm.alpha);
        alpha_val = alpha_nu_fluid_cgs * nu_fluid_cgs;
      }
      if (kAux) {
        if (kModel == BL_MODEL_FORMULA && nan_ray && l == 0) j_val = alpha_val = __longlong_as_double(0x7ff8000000000000ll);
        out[l] = make_double2(j_val, alpha_val);
      } else {
        const double delta_lambda_cgs = bl_div_g(delta_lambda * P.x_unit, freq * momentum_factor);   // unpolarized.cpp:75-76
        double2 rec = transfer_record(j_val, alpha_val, delta_lambda_cgs);
        if (kRedo) rec = rec.x == BL_THICK_MARK ? make_double2(BL_AFFINE_THICK, rec.y) : make_double2(rec.x, rec.x * rec.y);   // (a, b) -> (a, c)
        out[l] = rec;
        if (kRedo && P.tau_inc != nullptr) P.tau_inc[(out - P.transfer) + l] = alpha_val * delta_lambda_cgs;   // unpolarized.cpp:150-151
      }
    }
  }
}


// ---- the exact tier's coefficient kernel for the benchmark's case (plain image of a spherical Kerr-Schild simulation in a
// curved spacetime, thermal electrons): bl_shade_kernel<simulation, false, false, true, false, kSpinZero>'s arithmetic, call for
// call, behind bl_shade_fast_kernel's software pipeline - located sample of `next` loading, corner cells and record of `cur`
// requested, trilinear read and arithmetic of `prev` - with every load of the loop unconditional so that the waits are
// exact. The unpipelined kernel waited for memory in 44 % of its wave cycles. No arithmetic changes: bit-identical.
template <bool kSpinZero>
__global__ void __launch_bounds__(256, 2) bl_shade_exact_kernel(const BlShadeArgs P) {
  const BlSpacetime st = P.st;
  // (record indices as 32-bit numbers: a scratch set holds fewer than 2^32 records, bl_render sees to that)
  const uint32_t n_records = (uint32_t)P.counters_in[BL_CNT_RECORDS];
  const uint32_t stride = gridDim.x * blockDim.x;
  uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;   // record of `next`
  if (n_records == 0u) return;
  const float fallback_rho = P.cold->fallback_rho, fallback_pgas = P.cold->fallback_pgas;
  const uint32_t last = n_records - 1u;
  struct Located {
    double2 l0, l1;   // f_i, f_j | f_k, unwrapped azimuth
    unsigned long long tag;
  };
  auto load_located = [&](uint32_t at, Located &r) {
    const double2 *loc = reinterpret_cast<const double2 *>(P.located + at);
    r.l0 = loc[0];
    r.l1 = loc[1];
    r.tag = P.located_tag[at];
  };
  Located loc_prev, loc_cur, loc_next;
  FastRay ray_prev, ray_cur;
  float4 lo[8], hi[8];
  uint32_t idx_prev = 0u, idx_cur = 0u;
  bool have_prev = false, have_cur = false, have_next = idx < n_records;
  loc_prev.tag = loc_cur.tag = 0ull;
  loc_prev.l0 = loc_prev.l1 = loc_cur.l0 = loc_cur.l1 = make_double2(0.0, 0.0);
  ray_prev.q0 = ray_prev.q1 = ray_prev.q2 = ray_prev.q3 = make_double2(0.0, 0.0);
  ray_prev.q1.y = __longlong_as_double((long long)BL_DEAD_RAY);
#pragma unroll
  for (int c = 0; c < 8; c++) lo[c] = hi[c] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  load_located(have_next ? idx : last, loc_next);
  while (have_prev || have_cur || have_next) {
    const uint32_t ray = have_prev ? (uint32_t)__double_as_longlong(ray_prev.q1.y) : BL_DEAD_RAY;
    const bool live = ray != BL_DEAD_RAY;
    const uint32_t n = (uint32_t)(((unsigned long long)__double_as_longlong(ray_prev.q1.y)) >> 32);
    // (a dead slot has tag 0 = kSampleNone from the locate kernel; a stage without a sample reads the last record's slot)
    const int status = live ? (int)(loc_prev.tag >> 32) : (int)kSampleNone;
    // per-ray constants of `prev`: requested before the next sample's cells
    const double kt = P.ray_kt[live ? ray : 0u], momentum_factor = P.ray_factor[live ? ray : 0u];
    const uint32_t row = (uint32_t)P.ray_offset[live ? ray : 0u] + n;   // (rows are record counts: 32 bits as well)
    float pr[8];
    gather_finish(P, fallback_rho, fallback_pgas, status, lo, hi, loc_prev.l0.x, loc_prev.l0.y, loc_prev.l1.x, pr);
    gather_issue(P, have_cur ? (int)(loc_cur.tag >> 32) : (int)kSampleNone, (uint32_t)loc_cur.tag, lo, hi);
    fast_load_ray(P, have_cur ? idx_cur : last, ray_cur);
    const FastRay rec = ray_prev;
    const double ph = loc_prev.l1.y;
    const uint32_t idx_rec = idx_prev;
    loc_prev = loc_cur;
    ray_prev = ray_cur;
    idx_prev = idx_cur;
    have_prev = have_cur;
    loc_cur = loc_next;
    idx_cur = idx;
    have_cur = have_next;
    have_next = have_next && n_records > stride && idx < n_records - stride;   // (compared before the addition: no wrap-around)
    idx += stride;
    // (with spin the arithmetic below needs the ten registers of the located sample in flight: requested behind it instead)
    if (kSpinZero) load_located(have_next ? idx : last, loc_next);
    if (live) {
    // ---- from here on: bl_shade_kernel's body for this instantiation
    const double x1 = rec.q0.x, x2 = rec.q0.y, x3 = rec.q1.x;
    const double delta_lambda = -rec.q3.y;   // ReverseGeodesics: sample_len = -geodesic_len (:840)
    double kcov[4] = {kt, rec.q2.x, rec.q2.y, rec.q3.x};
    BlKerrSchild ks;
    bl_kerr_schild<kSpinZero>(st, x1, x2, x3, &ks);
    if (!P.samples_renormalised) {   // per-sample renormalisation of the stored momentum (geodesics.cpp:352-371)
      double gcon[4][4];
      bl_gcon_ks(ks, gcon);
      double factor = bl_renormalization_factor_g(gcon, kcov[0], kcov[1], kcov[2], kcov[3]);
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
    if (status != kSampleCut) sample_finish_simulation<false, true>(P, st, ks, x3 / ks.r, ph, pr, 0.0f, kcov, 1, &sh, nullptr);
    double2 *out = P.transfer + (size_t)row * P.n_nu;
    if (P.coef_split) {
      // several frequencies: the per-frequency formulas and transfer records are bl_coefficients_freq_kernel's
      BlCoefInputs ci;
      ci.nu_fluid_over_nu = sh.nu_fluid_over_nu;
      ci.n_e_cgs = sh.n_e_cgs;
      ci.nu_c_cgs = sh.nu_c_cgs;
      ci.theta_e = sh.theta_e;
      ci.kb_tt_e_cgs = sh.kb_tt_e_cgs;
      ci.cos2_theta_b = sh.sin_theta_b;   // (sin theta_B itself: nothing there needs the cosine)
      ci.cos_sign = delta_lambda;
      ci.have_coefficients = sh.have_coefficients ? 1.0 : 0.0;
      P.coef_inputs[idx_rec] = ci;
    } else {
      for (int l = 0; l < P.n_nu; l++) {
        const double freq = P.frequencies[l];
        double j_val = 0.0, alpha_val = 0.0;
        if (sh.have_coefficients) simulation_coefficients<false>(P, sh, freq, momentum_factor, &j_val, &alpha_val);
        const double delta_lambda_cgs = bl_div_g(delta_lambda * P.x_unit, freq * momentum_factor);   // unpolarized.cpp:75-76
        out[l] = transfer_record(j_val, alpha_val, delta_lambda_cgs);
      }
    }
    }
    if (!kSpinZero) load_located(have_next ? idx : last, loc_next);
  }
}


// =================================================================================================
// Launch wrappers (called from bl_render.hip)
// =================================================================================================
// Coefficient kernel
extern "C" hipError_t bl_launch_shade(const BlShadeArgs *args, int model, int grid, hipStream_t stream) {
  const bool aux = args->aux != nullptr;
  const bool power = args->plasma.power_frac != 0.0 || args->plasma.code_kappa != 0 || args->slow.n > 0 || args->pol_samples != nullptr
      || args->anchors != nullptr || args->plasma.kappa_unpolarized != 0;
  // the benchmark path - plain image of a spherical Kerr-Schild simulation in a curved spacetime - has its own
  // instantiation with those two facts known at compile time (62.4 instead of 64.4 ms per 1024^2 frame)
  const bool sks_curved = args->plasma.simulation_coord == BL_COORD_SKS && !args->st.ray_flat;
  const bool spin_zero = args->st.bh_a == 0.0;
#define BL_LAUNCH_S(M, A, W) hipLaunchKernelGGL((bl_shade_kernel<M, A, W, false, false, false>), dim3(grid), dim3(256), 0, stream, *args)
  if (model == BL_MODEL_SIMULATION) {
    // (a polarized run is an auxiliary-image run whether or not it keeps BlAuxSample records: BlShadeArgs::aux_record_unused)
    // polarized run: frame and coefficient inputs per sample, no frequency loop (the grids bl_shade_polarized2_kernel does not take)
    if (args->pol_samples != nullptr && sks_curved)
      hipLaunchKernelGGL((bl_shade_kernel<BL_MODEL_SIMULATION, true, true, true, true, false>), dim3(grid), dim3(256), 0, stream, *args);
    else if (args->pol_samples != nullptr)
      hipLaunchKernelGGL((bl_shade_kernel<BL_MODEL_SIMULATION, true, true, false, true, false>), dim3(grid), dim3(256), 0, stream, *args);
    else if (aux && power) BL_LAUNCH_S(BL_MODEL_SIMULATION, true, true);
    else if (aux) BL_LAUNCH_S(BL_MODEL_SIMULATION, true, false);
    else if (power) BL_LAUNCH_S(BL_MODEL_SIMULATION, false, true);
    else if (sks_curved) {   // plain image of a spherical Kerr-Schild simulation in a curved spacetime: the software-pipelined kernel
      if (spin_zero) hipLaunchKernelGGL((bl_shade_exact_kernel<true>), dim3(grid), dim3(256), 0, stream, *args);
      else hipLaunchKernelGGL((bl_shade_exact_kernel<false>), dim3(grid), dim3(256), 0, stream, *args);
    }
    else BL_LAUNCH_S(BL_MODEL_SIMULATION, false, false);
  } else {
    if (aux) BL_LAUNCH_S(BL_MODEL_FORMULA, true, false);
    else BL_LAUNCH_S(BL_MODEL_FORMULA, false, false);
  }
#undef BL_LAUNCH_S
  return hipGetLastError();
}

// Second pass of the tolerant tier (bl_shade_fast.hip launches it behind its kernels): bl_shade_kernel<..., kRedo> over the records
// the tolerant kernel listed. Its extended instantiation knows power laws, its general one Cartesian grids.
extern "C" hipError_t bl_launch_shade_redo(const BlShadeArgs *args, int model, int grid, hipStream_t stream) {
  if (model == BL_MODEL_FORMULA) {
    hipLaunchKernelGGL((bl_shade_kernel<BL_MODEL_FORMULA, false, false, false, false, false, true>), dim3(grid), dim3(256), 0, stream, *args);
    return hipGetLastError();
  }
  const bool spin_zero = args->st.bh_a == 0.0;
  const bool fused = args->located == nullptr;
  const bool cartesian = !fused && args->plasma.simulation_coord == BL_COORD_CKS;
  // (the extended instantiation; behind the fused kernel it is the one that knows the anchor cells of inter-block interpolation)
  const bool power_law = fused ? args->grid.block_interp != 0
                               : (args->plasma.power_frac != 0.0 || args->tau_inc != nullptr || args->anchors != nullptr || args->slow.n > 0);
  // (the common case - behind a kernel with the locate step inside - knows zero spin at compile time)
  // (behind the fused kernel over a mesh with inter-block interpolation: room for the mesh's tables in LDS, if they are small enough)
  const size_t tables = (fused && args->grid.n_blocks > 0 && args->grid.refined_lds_bytes > 0 && args->grid.refined_lds_bytes <= BL_REDO_TABLES_LDS)
      ? (size_t)args->grid.refined_lds_bytes : 0;
#define BL_LAUNCH_R(EXTENDED, SKS) hipLaunchKernelGGL((bl_shade_kernel<BL_MODEL_SIMULATION, false, EXTENDED, SKS, false, false, true>), dim3(grid), dim3(256), (EXTENDED) ? tables : 0, stream, *args)
  if (cartesian) BL_LAUNCH_R(true, false);
  else if (power_law) BL_LAUNCH_R(true, true);
  else if (spin_zero) hipLaunchKernelGGL((bl_shade_kernel<BL_MODEL_SIMULATION, false, false, true, false, true, true>), dim3(grid), dim3(256), 0, stream, *args);
  else BL_LAUNCH_R(false, true);
#undef BL_LAUNCH_R
  return hipGetLastError();
}

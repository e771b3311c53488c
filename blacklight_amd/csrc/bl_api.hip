// bl_api.hip - C-ABI of the MI355X hot path (include/blacklight_amd.h): context, parameter
// validation in the reference's words, grid repack + upload, settings (bl_render: bl_render.hip).
//
// Host-side counterpart of the reference's GeodesicIntegrator / RadiationIntegrator constructors
// (src/geodesic_integrator/geodesic_integrator.cpp:23-157, src/radiation_integrator/
// radiation_integrator.cpp:26-541) and of their Integrate() drivers, for the configurations in the
// hot-path scope. Configurations outside it are rejected loudly (BL_E_UNSUPPORTED); nothing falls
// back to a CPU path.
#include "bl_ctx.h"

namespace {

thread_local std::string g_global_error;

bool Has(const bl_params &p, int index) { return p.has[index] != 0; }

// std::optional::value() of the reference: a missing key surfaces as bad_optional_access, which
// main() reports per stage (blacklight.cpp:100-104, 147-151)
void Require(const bl_params &p, std::initializer_list<int> keys, const char *stage_message) {
  for (int key : keys)
    if (!Has(p, key)) throw Failure{BL_E_MISSING, stage_message};
}

constexpr const char *kGeoMissing = "GeodesicIntegrator unable to find all needed values in input file.";
constexpr const char *kRadMissing = "RadiationIntegrator unable to find all needed values in input file.";

// GeodesicIntegrator::GeodesicIntegrator (geodesic_integrator.cpp:23-157)
void ValidateGeodesic(bl_ctx *ctx) {
  const bl_params &p = ctx->params;
  Require(p, {BL_P_model_type, BL_P_checkpoint_geodesic_save, BL_P_checkpoint_geodesic_load}, kGeoMissing);
  if (p.checkpoint_geodesic_save && p.checkpoint_geodesic_load)
    throw Failure{BL_E_INPUT, "Cannot both save and load a geodesic checkpoint."};
  if (p.checkpoint_geodesic_save || p.checkpoint_geodesic_load) Require(p, {BL_P_checkpoint_geodesic_file}, kGeoMissing);
  Require(p, {BL_P_camera_type, BL_P_camera_r, BL_P_camera_th, BL_P_camera_ph, BL_P_camera_urn, BL_P_camera_uthn,
              BL_P_camera_uphn, BL_P_camera_k_r, BL_P_camera_k_th, BL_P_camera_k_ph, BL_P_camera_rotation,
              BL_P_camera_width, BL_P_camera_resolution, BL_P_camera_pole},
          kGeoMissing);
  if (p.camera_resolution <= 0) throw Failure{BL_E_INPUT, "Must have positive camera_resolution."};
  Require(p, {BL_P_ray_flat, BL_P_ray_terminate}, kGeoMissing);
  if (p.ray_terminate != BL_TERMINATE_PHOTON) Require(p, {BL_P_ray_factor}, kGeoMissing);
  Require(p, {BL_P_ray_integrator, BL_P_ray_step, BL_P_ray_max_steps}, kGeoMissing);
  if (p.ray_max_steps <= 0) throw Failure{BL_E_INPUT, "Must have positive ray_max_steps."};
  if (p.ray_integrator == BL_INTEGRATOR_DP) {
    Require(p, {BL_P_ray_max_retries}, kGeoMissing);
    if (p.ray_max_retries <= 0) throw Failure{BL_E_INPUT, "Must have nonnegative ray_max_retries."};
    Require(p, {BL_P_ray_tol_abs, BL_P_ray_tol_rel}, kGeoMissing);
  }
  Require(p, {BL_P_image_num_frequencies}, kGeoMissing);
  if (p.image_num_frequencies == 1) {
    Require(p, {BL_P_image_frequency}, kGeoMissing);
    if (p.image_frequency <= 0.0) throw Failure{BL_E_INPUT, "Must choose positive image_frequency."};
  } else if (p.image_num_frequencies > 1) {
    Require(p, {BL_P_image_frequency_start}, kGeoMissing);
    if (p.image_frequency_start <= 0.0) throw Failure{BL_E_INPUT, "Must choose positive image_frequency_start."};
    Require(p, {BL_P_image_frequency_end}, kGeoMissing);
    if (p.image_frequency_end <= 0.0) throw Failure{BL_E_INPUT, "Must choose positive image_frequency_end."};
    Require(p, {BL_P_image_frequency_spacing}, kGeoMissing);
  } else {
    throw Failure{BL_E_INPUT, "Must have positive image_num_frequencies."};
  }
  Require(p, {BL_P_image_normalization, BL_P_adaptive_max_level}, kGeoMissing);
  if (p.adaptive_max_level > 0) {
    Require(p, {BL_P_adaptive_block_size}, kGeoMissing);
    if (p.adaptive_block_size <= 0) throw Failure{BL_E_INPUT, "Must have positive adaptive_block_size."};
    if (p.camera_resolution % p.adaptive_block_size != 0)
      throw Failure{BL_E_INPUT, "Must have adaptive_block_size divide camera_resolution."};
  }
  // geometry (:107-123)
  ctx->st.bh_m = 1.0;
  if (p.model_type == BL_MODEL_SIMULATION) {
    Require(p, {BL_P_simulation_a}, kGeoMissing);
    ctx->st.bh_a = p.simulation_a;
  } else {
    Require(p, {BL_P_formula_spin}, kGeoMissing);
    ctx->st.bh_a = p.formula_spin;
  }
  ctx->st.ray_flat = p.ray_flat;
  bl_camera_frame &f = ctx->frame;
  f.bh_m = ctx->st.bh_m;
  f.bh_a = ctx->st.bh_a;
  f.r_horizon = f.bh_m + blm_sqrt(f.bh_m * f.bh_m - f.bh_a * f.bh_a);
  if (p.ray_terminate == BL_TERMINATE_PHOTON)
    f.r_terminate = 2.0 * f.bh_m * (1.0 + bl_cos(2.0 / 3.0 * bl_acos(-blm_abs(f.bh_a) / f.bh_m)));
  else if (p.ray_terminate == BL_TERMINATE_MULTIPLICATIVE)
    f.r_terminate = f.r_horizon * p.ray_factor;
  else
    f.r_terminate = f.r_horizon + p.ray_factor;
}

// InitializeCamera frequency list (camera.cpp:30-50)
void BuildFrequencies(bl_ctx *ctx) {
  const bl_params &p = ctx->params;
  int nf = p.image_num_frequencies;
  ctx->frequencies.assign(nf, 0.0);
  if (nf == 1) {
    ctx->frequencies[0] = p.image_frequency;
    return;
  }
  ctx->frequencies[0] = p.image_frequency_start;
  ctx->frequencies[nf - 1] = p.image_frequency_end;
  for (int l = 1; l < nf - 1; l++) {
    double frac = static_cast<double>(l) / static_cast<double>(nf - 1);
    if (p.image_frequency_spacing == BL_SPACING_LIN_FREQ)
      ctx->frequencies[l] = p.image_frequency_start + frac * (p.image_frequency_end - p.image_frequency_start);
    else if (p.image_frequency_spacing == BL_SPACING_LIN_WAVE)
      ctx->frequencies[l] = 1.0 / (1.0 / p.image_frequency_start
          + frac * (1.0 / p.image_frequency_end - 1.0 / p.image_frequency_start));
    else
      ctx->frequencies[l] = bl_exp(bl_log(p.image_frequency_start) + frac * bl_log(p.image_frequency_end / p.image_frequency_start));
  }
}

// RadiationIntegrator::RadiationIntegrator (radiation_integrator.cpp:26-541), hot-path subset
void ValidateRadiation(bl_ctx *ctx) {
  bl_params &p = ctx->params;
  const bool simulation = p.model_type == BL_MODEL_SIMULATION;
  Require(p, {BL_P_num_threads}, kRadMissing);
  if (simulation) {
    Require(p, {BL_P_checkpoint_sample_save, BL_P_checkpoint_sample_load}, kRadMissing);
    if (p.checkpoint_sample_save && p.checkpoint_sample_load)
      throw Failure{BL_E_INPUT, "Cannot both save and load a sample checkpoint."};
    if (p.checkpoint_sample_save || p.checkpoint_sample_load) Require(p, {BL_P_checkpoint_sample_file}, kRadMissing);
    if (p.checkpoint_sample_load)
      // The reference cannot read these files itself: LoadSampling() (sample_checkpoint.cpp:49-63) restores sample_inds,
      // sample_fracs, sample_nan and sample_fallback but not sample_cut, which only CalculateSimulationSampling() allocates
      // (simulation_sampling.cpp:155) and SampleSimulation() reads for every sample (:691) - the reference binary built
      // from /root/reference ends in a segmentation fault on checkpoint_sample_load = true. Loading has no defined result to
      // match and is refused; saving writes the reference's file (bl_render.hip, WriteSampleCheckpoint).
      throw Failure{BL_E_UNSUPPORTED, "checkpoint_sample_load is not offered: the reference cannot load its own sample checkpoints (its "
                                      "LoadSampling() leaves sample_cut unallocated); use geodesic checkpoints."};
    Require(p, {BL_P_simulation_format, BL_P_simulation_coord, BL_P_simulation_m_msun, BL_P_simulation_rho_cgs,
                BL_P_simulation_interp},
            kRadMissing);
    if ((p.simulation_format == BL_SIMFMT_ATHENA || p.simulation_format == BL_SIMFMT_ATHENAK) && p.simulation_interp) {
      Require(p, {BL_P_simulation_block_interp}, kRadMissing);
    } else if (Has(p, BL_P_simulation_block_interp)) {
      Warn(ctx, "Ignoring simulation_block_interp selection.");
    }
  } else {
    if (Has(p, BL_P_checkpoint_sample_save) && p.checkpoint_sample_save) Warn(ctx, "Ignoring checkpoint_sample_save selection.");
    if (Has(p, BL_P_checkpoint_sample_load) && p.checkpoint_sample_load) Warn(ctx, "Ignoring checkpoint_sample_load selection.");
    Require(p, {BL_P_formula_mass, BL_P_formula_r0, BL_P_formula_h, BL_P_formula_l0, BL_P_formula_q, BL_P_formula_nup,
                BL_P_formula_cn0, BL_P_formula_alpha, BL_P_formula_a, BL_P_formula_beta},
            kRadMissing);
  }
  Require(p, {BL_P_image_light}, kRadMissing);
  bool polarization = false;
  if (p.image_light) {
    if (simulation) {
      Require(p, {BL_P_image_polarization}, kRadMissing);
      polarization = p.image_polarization != 0;
    } else if (Has(p, BL_P_image_polarization) && p.image_polarization) {
      Warn(ctx, "Ignoring image_polarization selection.");
    }
    if (polarization) Require(p, {BL_P_image_rotation_split}, kRadMissing);
  } else if (Has(p, BL_P_image_polarization) && p.image_polarization) {
    Warn(ctx, "Ignoring image_polarization selection.");
  }
  Require(p, {BL_P_image_time, BL_P_image_length, BL_P_image_lambda, BL_P_image_emission, BL_P_image_tau}, kRadMissing);
  if (simulation) {
    Require(p, {BL_P_image_lambda_ave, BL_P_image_emission_ave, BL_P_image_tau_int}, kRadMissing);
  } else {
    if (Has(p, BL_P_image_lambda_ave) && p.image_lambda_ave) Warn(ctx, "Ignoring image_lambda_ave selection.");
    if (Has(p, BL_P_image_emission_ave) && p.image_emission_ave) Warn(ctx, "Ignoring image_emission_ave selection.");
    if (Has(p, BL_P_image_tau_int) && p.image_tau_int) Warn(ctx, "Ignoring image_tau_int selection.");
    p.image_lambda_ave = p.image_emission_ave = p.image_tau_int = 0;
  }
  Require(p, {BL_P_image_crossings}, kRadMissing);
  int render_num_images = 0;
  if (simulation) {
    Require(p, {BL_P_render_num_images}, kRadMissing);
    render_num_images = p.render_num_images;
  } else if (Has(p, BL_P_render_num_images) && p.render_num_images > 0) {
    Warn(ctx, "Ignoring request for rendering.");
  }
  if (!(p.image_light || p.image_time || p.image_length || p.image_lambda || p.image_emission || p.image_tau
        || p.image_lambda_ave || p.image_emission_ave || p.image_tau_int || p.image_crossings || render_num_images > 0))
    throw Failure{BL_E_INPUT, "No image or rendering selected."};
  // rendering parameters (radiation_integrator.cpp:146-197): every value the features need must be present
  ctx->render_num_images = render_num_images;
  for (int n_i = 0; n_i < render_num_images; n_i++) {
    if (!p.render_num_features_has[n_i]) throw Failure{BL_E_MISSING, kRadMissing};
    const int num_features = p.render_num_features[n_i];
    if (num_features <= 0) throw Failure{BL_E_INPUT, "Must have positive number of features for each rendered image."};
    for (int n_f = 0; n_f < num_features; n_f++) {
      const int has = p.render_has[n_i][n_f];
      int need = BL_RENDER_HAS_QUANTITY | BL_RENDER_HAS_TYPE | BL_RENDER_HAS_XYZ;
      if ((has & BL_RENDER_HAS_TYPE) != 0) {
        if (p.render_type[n_i][n_f] == BL_RENDER_FILL)
          need |= BL_RENDER_HAS_MIN | BL_RENDER_HAS_MAX | BL_RENDER_HAS_TAU_SCALE;
        else
          need |= BL_RENDER_HAS_THRESH | BL_RENDER_HAS_OPACITY;
      }
      if ((has & need) != need) throw Failure{BL_E_MISSING, kRadMissing};
    }
  }
  ctx->polarized = polarization;
  if (simulation) {
    Require(p, {BL_P_slow_light_on}, kRadMissing);
    if (p.slow_light_on) {   // radiation_integrator.cpp:206-215
      if ((p.has[BL_P_checkpoint_sample_save] && p.checkpoint_sample_save) || (p.has[BL_P_checkpoint_sample_load] && p.checkpoint_sample_load))
        throw Failure{BL_E_INPUT, "Cannot use sample checkpoints with slow light."};
      Require(p, {BL_P_slow_interp, BL_P_slow_chunk_size, BL_P_slow_t_start, BL_P_slow_dt}, kRadMissing);
      if (p.slow_chunk_size < 2) throw Failure{BL_E_INPUT, "Must have slow_chunk_size be at least 2."};   // simulation_reader.cpp:77
    }
  }
  if (p.adaptive_max_level > 0) {   // radiation_integrator.cpp:218-270
    if (!p.image_light) throw Failure{BL_E_INPUT, "Adaptive ray tracing requires image_light."};
    if (p.adaptive_max_level > BL_MAX_LEVELS) throw Failure{BL_E_UNSUPPORTED, "adaptive_max_level exceeds BL_MAX_LEVELS."};
    if (p.image_num_frequencies > 1) {
      Require(p, {BL_P_adaptive_frequency_num}, kRadMissing);
      if (p.adaptive_frequency_num - 1 < 0 || p.adaptive_frequency_num - 1 >= p.image_num_frequencies)
        throw Failure{BL_E_INPUT, "Must choose adaptive_frequency_num from 1 to image_num_frequencies."};
    }
    Require(p, {BL_P_adaptive_val_frac}, kRadMissing);
    if (p.adaptive_val_frac >= 0.0) Require(p, {BL_P_adaptive_val_cut}, kRadMissing);
    Require(p, {BL_P_adaptive_abs_grad_frac}, kRadMissing);
    if (p.adaptive_abs_grad_frac >= 0.0) Require(p, {BL_P_adaptive_abs_grad_cut}, kRadMissing);
    Require(p, {BL_P_adaptive_rel_grad_frac}, kRadMissing);
    if (p.adaptive_rel_grad_frac >= 0.0) Require(p, {BL_P_adaptive_rel_grad_cut}, kRadMissing);
    Require(p, {BL_P_adaptive_abs_lapl_frac}, kRadMissing);
    if (p.adaptive_abs_lapl_frac >= 0.0) Require(p, {BL_P_adaptive_abs_lapl_cut}, kRadMissing);
    Require(p, {BL_P_adaptive_rel_lapl_frac}, kRadMissing);
    if (p.adaptive_rel_lapl_frac >= 0.0) Require(p, {BL_P_adaptive_rel_lapl_cut}, kRadMissing);
    Require(p, {BL_P_adaptive_num_regions}, kRadMissing);
    for (int r = 0; r < p.adaptive_num_regions; r++)
      if (p.adaptive_region_has[r] != 31) throw Failure{BL_E_MISSING, kRadMissing};
  }
  if (simulation) {
    Require(p, {BL_P_plasma_mu, BL_P_plasma_ne_ni, BL_P_plasma_model}, kRadMissing);
    if (p.plasma_model == BL_PLASMA_TI_TE_BETA)
      Require(p, {BL_P_plasma_use_p, BL_P_plasma_rat_low, BL_P_plasma_rat_high}, kRadMissing);
    Require(p, {BL_P_plasma_power_frac}, kRadMissing);
    if (p.plasma_power_frac < 0.0 || p.plasma_power_frac > 1.0) Warn(ctx, "Fraction of power-law electrons outside [0, 1].");
    Require(p, {BL_P_plasma_kappa_frac}, kRadMissing);
    if (p.plasma_kappa_frac < 0.0 || p.plasma_kappa_frac > 1.0) Warn(ctx, "Fraction of kappa-distribution electrons outside [0, 1].");
    if (p.plasma_power_frac != 0.0) Require(p, {BL_P_plasma_p, BL_P_plasma_gamma_min, BL_P_plasma_gamma_max}, kRadMissing);
    if (p.plasma_kappa_frac != 0.0) {   // radiation_integrator.cpp:296-308
      Require(p, {BL_P_plasma_kappa}, kRadMissing);
      if (p.image_light && p.image_polarization) {
        if (p.plasma_kappa < 3.5 || p.plasma_kappa > 5.0) throw Failure{BL_E_INPUT, "Polarized transport only supports kappa in [3.5, 5]."};
        if (p.plasma_kappa != 3.5 && p.plasma_kappa != 4.0 && p.plasma_kappa != 4.5 && p.plasma_kappa != 5.0)
          Warn(ctx, "Polarized transport will interpolate formulas based on kappa.");
      }
      Require(p, {BL_P_plasma_w}, kRadMissing);
      // (an unpolarized run with such electrons: bl_render refuses it unless bl_set_undefined_policy(BL_UNDEFINED_KAPPA) was called)
    }
    ctx->plasma_thermal_frac = 1.0 - (p.plasma_power_frac + p.plasma_kappa_frac);
    if (ctx->plasma_thermal_frac < 0.0 || ctx->plasma_thermal_frac > 1.0) Warn(ctx, "Fraction of thermal electrons outside [0, 1].");
    Require(p, {BL_P_cut_rho_min, BL_P_cut_rho_max, BL_P_cut_n_e_min, BL_P_cut_n_e_max, BL_P_cut_p_gas_min,
                BL_P_cut_p_gas_max, BL_P_cut_theta_e_min, BL_P_cut_theta_e_max, BL_P_cut_b_min, BL_P_cut_b_max,
                BL_P_cut_sigma_min, BL_P_cut_sigma_max, BL_P_cut_beta_inverse_min, BL_P_cut_beta_inverse_max},
            kRadMissing);
  }
  Require(p, {BL_P_cut_omit_near, BL_P_cut_omit_far, BL_P_cut_omit_in, BL_P_cut_omit_out, BL_P_cut_midplane_theta,
              BL_P_cut_midplane_z, BL_P_cut_plane},
          kRadMissing);
  if (p.cut_plane)
    Require(p, {BL_P_cut_plane_origin_x, BL_P_cut_plane_origin_y, BL_P_cut_plane_origin_z, BL_P_cut_plane_normal_x,
                BL_P_cut_plane_normal_y, BL_P_cut_plane_normal_z},
            kRadMissing);
  Require(p, {BL_P_fallback_nan}, kRadMissing);
  if (simulation && !p.fallback_nan) Require(p, {BL_P_fallback_rho, BL_P_fallback_pgas}, kRadMissing);
  if (simulation && !p.fallback_nan && p.plasma_model == BL_PLASMA_CODE_KAPPA) Require(p, {BL_P_fallback_kappa}, kRadMissing);

  // geometry data and image rows (:419-520)
  ctx->frame.mass_msun = simulation ? p.simulation_m_msun : p.formula_mass * kC * kC / kGGMsun;
  // image rows and their offsets (radiation_integrator.cpp:436-520); polarization is rejected above
  BlAuxImages &A = ctx->aux_images;
  A = BlAuxImages{};
  const int nf = p.image_num_frequencies;
  A.image_light = p.image_light;
  A.image_time = p.image_time;
  A.image_length = p.image_length;
  A.image_lambda = p.image_lambda;
  A.image_emission = p.image_emission;
  A.image_tau = p.image_tau;
  A.image_lambda_ave = simulation && p.image_lambda_ave;
  A.image_emission_ave = simulation && p.image_emission_ave;
  A.image_tau_int = simulation && p.image_tau_int;
  A.image_crossings = p.image_crossings;
  int n_q = 0;
  A.polarized = ctx->polarized ? 1 : 0;
  if (A.image_light) n_q += nf * (ctx->polarized ? 4 : 1);
  A.offset_time = n_q;
  if (A.image_time) n_q += 1;
  A.offset_length = n_q;
  if (A.image_length) n_q += 1;
  A.offset_lambda = n_q;
  if (A.image_lambda) n_q += nf;
  A.offset_emission = n_q;
  if (A.image_emission) n_q += nf;
  A.offset_tau = n_q;
  if (A.image_tau) n_q += nf;
  A.offset_lambda_ave = n_q;
  if (A.image_lambda_ave) n_q += nf * kNumCellValues;
  A.offset_emission_ave = n_q;
  if (A.image_emission_ave) n_q += nf * kNumCellValues;
  A.offset_tau_int = n_q;
  if (A.image_tau_int) n_q += nf * kNumCellValues;
  A.offset_crossings = n_q;
  if (A.image_crossings) n_q += 1;
  A.n_q = n_q;
  A.any = (A.image_time || A.image_length || A.image_lambda || A.image_emission || A.image_tau || A.image_lambda_ave
           || A.image_emission_ave || A.image_tau_int || A.image_crossings || ctx->render_num_images > 0
           || ctx->polarized) ? 1 : 0;   // polarized transfer runs in auxiliary-image mode
  ctx->image_num_quantities = n_q;
}

void BuildBuckets(const double *xf, int n, int n_bucket, std::vector<int> *table, double *x0, double *inv_w) {
  // bucket b covers [x0 + b w, x0 + (b+1) w); table[b] = first cell c with xf[c+1] >= x0 + (b-1) w,
  // i.e. a start index that is never beyond the reference's linear-scan result for any x whose
  // bucket index evaluates to b (one bucket of slack covers rounding of the index computation)
  double lo = xf[0], hi = xf[n];
  double w = (hi - lo) / n_bucket;
  *x0 = lo;
  *inv_w = 1.0 / w;
  table->resize(n_bucket);
  int c = 0;
  for (int b = 0; b < n_bucket; b++) {
    double edge = lo + (b - 1) * w;
    while (c < n - 1 && !(xf[c + 1] >= edge)) c++;
    (*table)[b] = c;
  }
}

}  // namespace

namespace blhost {
int Fail(bl_ctx *ctx, const Failure &failure) {
  std::string text = "Error: " + failure.message + "\n";
  if (ctx != nullptr)
    ctx->last_error = text;
  else
    g_global_error = text;
  return failure.code;
}
}  // namespace blhost

extern "C" {

int bl_init(const bl_params *p, int device, bl_ctx **out) {
  if (p == nullptr || out == nullptr) return BL_E_ARG;
  *out = nullptr;
  bl_ctx *ctx = new bl_ctx();
  ctx->params = *p;
  {   // measurement switches: the environment is read here and nowhere else (getenv is not safe beside a setenv in another thread)
    const struct { const char *name; unsigned int bit; } kSwitches[] = {
        {"BLACKLIGHT_AMD_TENSOR_TRANSPORT", BL_SWITCH_TENSOR_TRANSPORT}, {"BLACKLIGHT_AMD_SPLIT_RECORDS", BL_SWITCH_SPLIT_RECORDS},
        {"BLACKLIGHT_AMD_RECORD_EVERY_STEP", BL_SWITCH_RECORD_EVERY_STEP},
        {"BLACKLIGHT_AMD_GENERAL_LOCATE", BL_SWITCH_GENERAL_LOCATE}, {"BLACKLIGHT_AMD_LANE_TRANSFER", BL_SWITCH_LANE_TRANSFER},
        {"BLACKLIGHT_AMD_NO_FUSED_LOCATE", BL_SWITCH_NO_FUSED_LOCATE}, {"BLACKLIGHT_AMD_SAMPLE_RECORDS", BL_SWITCH_SAMPLE_RECORDS},
        {"BLACKLIGHT_AMD_QUAD_EVERY_RAY", BL_SWITCH_QUAD_EVERY_RAY}};
    for (const auto &sw : kSwitches)
      if (std::getenv(sw.name) != nullptr) ctx->switches |= sw.bit;
    ctx->debug_counters = std::getenv("BLACKLIGHT_AMD_DEBUG_COUNTERS") != nullptr;
    // the arithmetic tier a context starts in: tolerant (north_star's tolerance), or what the deployment says
    // (exact | tolerant, in any case; anything else is an error of bl_init - a typo must not silently select a tier)
    if (const char *tier = std::getenv("BLACKLIGHT_AMD_ARITHMETIC")) {
      std::string name = tier;
      for (char &c : name) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
      if (name == "exact") ctx->arithmetic = BL_ARITH_EXACT;
      else if (name == "tolerant") ctx->arithmetic = BL_ARITH_TOLERANT;
      else {
        g_global_error = "Error: BLACKLIGHT_AMD_ARITHMETIC must be exact or tolerant, not \"" + std::string(tier) + "\".\n";
        delete ctx;
        return BL_E_INPUT;
      }
    }
    if (const char *policy = std::getenv("BLACKLIGHT_AMD_TAIL_POLICY")) {   // (A/B runs: bl_set_tail_policy from the environment)
      const std::string name = policy;
      if (name != "auto" && name != "wide" && name != "quad" && name != "split") {
        g_global_error = "Error: BLACKLIGHT_AMD_TAIL_POLICY must be auto, wide, quad or split, not \"" + name + "\".\n";
        delete ctx;
        return BL_E_INPUT;
      }
      ctx->tail_policy = name == "wide" ? BL_TAIL_WIDE : (name == "quad" ? BL_TAIL_QUAD : (name == "split" ? BL_TAIL_SPLIT : BL_TAIL_AUTO));
    }
  }
  try {
    ValidateGeodesic(ctx);
    ValidateRadiation(ctx);
    BuildFrequencies(ctx);
    bl_camera_frame_build(ctx->params, ctx->st, &ctx->frame);
    if (device == BL_DEVICE_NONE) {   // host-only context: validation, camera frame, refinement, writer
      ctx->device = BL_DEVICE_NONE;
      *out = ctx;
      return BL_OK;
    }
    int count = 0;
    hipError_t err = hipGetDeviceCount(&count);
    if (err != hipSuccess || count <= 0)
      throw Failure{BL_E_DEVICE, "No HIP device available: the MI355X hot path has no CPU fallback."};
    if (device < 0) Check(hipGetDevice(&device), "hipGetDevice");
    if (device >= count) throw Failure{BL_E_DEVICE, "Requested device index out of range."};
    Check(hipSetDevice(device), "hipSetDevice");
    ctx->device = device;
    hipDeviceProp_t prop;
    Check(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties");
    ctx->num_cus = prop.multiProcessorCount;
    // scratch for sample records: four fifths of the device's memory unless bl_set_scratch_limit says otherwise (MI355X: 230 of
    // 288 GB - a 1024^2 full-Stokes frame in two chunks; whatever a render plans is capped by what is actually free)
    ctx->scratch_limit = static_cast<uint64_t>(0.8 * static_cast<double>(prop.totalGlobalMem));
    EnsureStreams(ctx);
  } catch (const Failure &failure) {
    int code = Fail(nullptr, failure);
    delete ctx;
    return code;
  }
  *out = ctx;
  return BL_OK;
}

namespace {

// Grid of equal blocks at one refinement level tiling a box, in any order (simulation_sampling.cpp:352-394
// searches the blocks per sample): merged into one global array at upload; the locate kernel keeps the
// reference's per-block anchor rules through the block size. Throws kIrregular when the blocks are not such
// a tiling (bl_set_grid then takes the refined-mesh path).
const char *const kIrregular = "Multi-block grid is not a regular tiling by equal blocks of one level.";

// The cells into their HBM layout on the device. The caller's arrays are [variable][block][k][j][i] (simulation_reader.cpp:767-780);
// the kernels read [cell][8 variables]. Eight planes are uploaded as they lie (a copy each, no pass over them on the host) and
// one kernel interleaves them - a lane per cell: eight coalesced reads, 32 contiguous bytes written - placing a block's cells at the
// block's position in the merged array where there is one (block_origin: first target cell of every block, and the target's row
// and plane strides; null: the blocks stay one behind the other). On the host this repack was 0.3 s per 256^3 snapshot, ten times
// the render of a series' frame over resident geodesics.
__global__ void __launch_bounds__(256) bl_interleave_cells_kernel(const float *planes, float *cells, unsigned long long n_cells, int nb_i, int nb_j, int nb_k,
                                                                  const unsigned long long *block_origin, unsigned long long row_stride,
                                                                  unsigned long long plane_stride) {
  const unsigned long long c = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cells) return;
  unsigned long long target = c;
  if (block_origin != nullptr) {
    const unsigned long long block_cells = (unsigned long long)nb_i * nb_j * nb_k;
    const unsigned long long blk = c / block_cells, rest = c - blk * block_cells;
    const unsigned long long k = rest / ((unsigned long long)nb_i * nb_j), j = (rest / nb_i) % nb_j, i = rest % nb_i;
    target = block_origin[blk] + k * plane_stride + j * row_stride + i;
  }
  float v[8];
#pragma unroll
  for (int q = 0; q < 8; q++) v[q] = planes[(unsigned long long)q * n_cells + c];
  float4 *out = reinterpret_cast<float4 *>(cells + target * 8);
  out[0] = make_float4(v[0], v[1], v[2], v[3]);
  out[1] = make_float4(v[4], v[5], v[6], v[7]);
}

// planes[v] = the caller's variable order[v]; block_origin (host, n_blocks entries) or null
// The planes of `g` into `d_cells` as ctx->placement says, on `stream`; returns when the cells are in place (the caller's arrays are
// borrowed for the duration of bl_set_grid)
void UploadCellsAs(bl_ctx *ctx, const bl_grid_desc *g, DeviceBuffer<float> &d_cells, hipStream_t stream, bool upload_origin) {
  const bl_ctx::CellPlacement &pl = ctx->placement;
  const size_t n_cells = pl.n_cells;
  d_cells.Ensure(n_cells * 8);
  ctx->d_cell_planes.Ensure(n_cells * 8);
  for (int v = 0; v < 8; v++)
    Check(hipMemcpyAsync(ctx->d_cell_planes.ptr + static_cast<size_t>(v) * n_cells, g->prim + static_cast<size_t>(pl.order[v]) * n_cells, n_cells * sizeof(float),
                         hipMemcpyHostToDevice, stream), "grid upload");
  const unsigned long long *origin = nullptr;
  if (pl.has_origin) {
    if (upload_origin) {
      ctx->d_block_origin.Ensure(pl.origin.size());
      Check(hipMemcpyAsync(ctx->d_block_origin.ptr, pl.origin.data(), pl.origin.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, stream), "grid upload");
    }
    origin = ctx->d_block_origin.ptr;
  }
  hipLaunchKernelGGL(bl_interleave_cells_kernel, dim3(static_cast<unsigned int>((n_cells + 255) / 256)), dim3(256), 0, stream, ctx->d_cell_planes.ptr, d_cells.ptr,
                     static_cast<unsigned long long>(n_cells), pl.nb[0], pl.nb[1], pl.nb[2], origin, pl.row_stride, pl.plane_stride);
  Check(hipGetLastError(), "cell interleave kernel launch");
  Check(hipStreamSynchronize(stream), "grid upload");
}

// planes[v] = the caller's variable order[v]; block_origin (host, n_blocks entries) or null
void UploadCells(bl_ctx *ctx, const bl_grid_desc *g, const int order[8], size_t n_cells, DeviceBuffer<float> &d_cells, const int nb[3],
                 const std::vector<unsigned long long> *block_origin, unsigned long long row_stride, unsigned long long plane_stride) {
  EnsureStreams(ctx);
  bl_ctx::CellPlacement &pl = ctx->placement;
  pl.valid = false;
  pl.n_cells = n_cells;
  for (int a = 0; a < 3; a++) pl.nb[a] = nb[a];
  for (int v = 0; v < 8; v++) pl.order[v] = order[v];
  pl.has_origin = block_origin != nullptr;
  pl.origin = block_origin != nullptr ? *block_origin : std::vector<unsigned long long>();
  pl.row_stride = row_stride;
  pl.plane_stride = plane_stride;
  UploadCellsAs(ctx, g, d_cells, ctx->stream, true);
  pl.valid = ctx->cells_target == nullptr;   // (the time slices of slow light have no second array to change places with)
}

void UploadMergedGrid(bl_ctx *ctx, const bl_grid_desc *g) {
    // Several blocks (simulation_sampling.cpp:352-394 searches them per sample): supported when they are
    // equal blocks at one refinement level tiling a box, in any order. They are merged into one global
    // array at upload; the locate kernel keeps the reference's per-block anchor rules through the block
    // size. Mesh refinement (blocks of different levels) is not built.
    const int nb_cells[3] = {g->n_i, g->n_j, g->n_k};
    const double *block_xf[3] = {g->x1f, g->x2f, g->x3f};
    const double *block_xv[3] = {g->x1v, g->x2v, g->x3v};
    const int n_b = g->n_blocks;
    std::vector<double> starts[3];          // distinct first faces along each axis, ascending
    std::vector<int> block_pos[3];          // position of every block along each axis
    for (int a = 0; a < 3; a++) {
      for (int blk = 0; blk < n_b; blk++) starts[a].push_back(block_xf[a][static_cast<size_t>(blk) * (nb_cells[a] + 1)]);
      std::sort(starts[a].begin(), starts[a].end());
      starts[a].erase(std::unique(starts[a].begin(), starts[a].end()), starts[a].end());
      block_pos[a].resize(n_b);
      for (int blk = 0; blk < n_b; blk++) {
        const double first = block_xf[a][static_cast<size_t>(blk) * (nb_cells[a] + 1)];
        block_pos[a][blk] = static_cast<int>(std::lower_bound(starts[a].begin(), starts[a].end(), first) - starts[a].begin());
      }
    }
    const int nbl[3] = {static_cast<int>(starts[0].size()), static_cast<int>(starts[1].size()), static_cast<int>(starts[2].size())};
    if (static_cast<long long>(nbl[0]) * nbl[1] * nbl[2] != n_b) throw Failure{BL_E_UNSUPPORTED, kIrregular};
    std::vector<int> block_at(n_b, -1);     // lattice position -> block
    for (int blk = 0; blk < n_b; blk++) {
      const int at = (block_pos[2][blk] * nbl[1] + block_pos[1][blk]) * nbl[0] + block_pos[0][blk];
      if (block_at[at] != -1) throw Failure{BL_E_UNSUPPORTED, kIrregular};
      block_at[at] = blk;
    }
    ctx->merged_block_at = block_at;        // (sample checkpoints name cells by MeshBlock and block-local indices)
    for (int a = 0; a < 3; a++) ctx->merged_blocks[a] = nbl[a];
    // global coordinate tables; every block at the same position along an axis must carry the same rows,
    // and neighbouring rows must meet bit for bit (no gaps, no overlaps)
    const int n_i = nbl[0] * nb_cells[0], n_j = nbl[1] * nb_cells[1], n_k = nbl[2] * nb_cells[2];
    const int n[3] = {n_i, n_j, n_k};
    std::vector<double> global_xf[3], global_xv[3];
    for (int a = 0; a < 3; a++) {
      global_xf[a].assign(n[a] + 1, 0.0);
      global_xv[a].assign(n[a], 0.0);
      std::vector<char> seen(nbl[a], 0);
      for (int blk = 0; blk < n_b; blk++) {
        const int pos = block_pos[a][blk];
        const double *f = block_xf[a] + static_cast<size_t>(blk) * (nb_cells[a] + 1);
        const double *v = block_xv[a] + static_cast<size_t>(blk) * nb_cells[a];
        double *gf = global_xf[a].data() + static_cast<size_t>(pos) * nb_cells[a];
        double *gv = global_xv[a].data() + static_cast<size_t>(pos) * nb_cells[a];
        if (!seen[pos]) {
          if (pos > 0 && seen[pos - 1] && std::memcmp(gf, f, sizeof(double)) != 0) throw Failure{BL_E_UNSUPPORTED, kIrregular};
          std::memcpy(gf, f, sizeof(double) * (nb_cells[a] + 1));
          std::memcpy(gv, v, sizeof(double) * nb_cells[a]);
          seen[pos] = 1;
        } else if (std::memcmp(gf, f, sizeof(double) * (nb_cells[a] + 1)) != 0 || std::memcmp(gv, v, sizeof(double) * nb_cells[a]) != 0) {
          throw Failure{BL_E_UNSUPPORTED, kIrregular};
        }
      }
      // joins written by a later block: the first face of block pos must equal the last face of pos - 1
      for (int blk = 0; blk < n_b; blk++) {
        const int pos = block_pos[a][blk];
        const double *f = block_xf[a] + static_cast<size_t>(blk) * (nb_cells[a] + 1);
        if (std::memcmp(global_xf[a].data() + static_cast<size_t>(pos) * nb_cells[a], f, sizeof(double)) != 0
            || std::memcmp(global_xf[a].data() + static_cast<size_t>(pos + 1) * nb_cells[a], f + nb_cells[a], sizeof(double)) != 0)
          throw Failure{BL_E_UNSUPPORTED, kIrregular};
      }
    }
    const size_t n_cells = static_cast<size_t>(n_i) * n_j * n_k;
    const size_t block_cells = static_cast<size_t>(nb_cells[0]) * nb_cells[1] * nb_cells[2];
    // Repack [var][block][k][j][i] -> global [k][j][i][8]: rho, pgas, uu1, uu2, uu3, bb1, bb2, bb3
    const int order[8] = {g->ind_rho, g->ind_pgas, g->ind_uu1, g->ind_uu2, g->ind_uu3, g->ind_bb1, g->ind_bb2, g->ind_bb3};
    for (int v : order)
      if (v < 0 || v >= g->n_var) throw Failure{BL_E_ARG, "Grid variable index out of range."};
    DeviceBuffer<float> &d_cells = ctx->cells_target != nullptr ? *ctx->cells_target : ctx->d_cells;
    DeviceBuffer<float> &d_kappa = ctx->kappa_target != nullptr ? *ctx->kappa_target : ctx->d_kappa;
    {
      std::vector<unsigned long long> origin(n_b);   // first cell of every block in the merged array
      for (int blk = 0; blk < n_b; blk++)
        origin[blk] = (static_cast<unsigned long long>(block_pos[2][blk]) * nb_cells[2] * n_j + static_cast<unsigned long long>(block_pos[1][blk]) * nb_cells[1]) * n_i
            + static_cast<unsigned long long>(block_pos[0][blk]) * nb_cells[0];
      UploadCells(ctx, g, order, n_cells, d_cells, nb_cells, n_b > 1 ? &origin : nullptr, static_cast<unsigned long long>(n_i), static_cast<unsigned long long>(n_i) * n_j);
    }
    const bool code_kappa = ctx->params.plasma_model == BL_PLASMA_CODE_KAPPA;
    if (code_kappa) {
      // the ninth value of a cell (simulation_reader.cpp:1164-1172) in its own [k][j][i] array: only the
      // extended coefficient kernel reads it
      if (g->ind_kappa < 0 || g->ind_kappa >= g->n_var) throw Failure{BL_E_ARG, "Grid variable index out of range."};
      std::vector<float> kappa(n_cells);
      for (int blk = 0; blk < n_b; blk++) {
        const int pi = block_pos[0][blk], pj = block_pos[1][blk], pk = block_pos[2][blk];
        const float *src = g->prim + (static_cast<size_t>(g->ind_kappa) * n_b + blk) * block_cells;
        for (int k = 0; k < nb_cells[2]; k++)
          for (int j = 0; j < nb_cells[1]; j++) {
            const size_t row = (static_cast<size_t>(pk * nb_cells[2] + k) * n_j + (pj * nb_cells[1] + j)) * n_i + static_cast<size_t>(pi) * nb_cells[0];
            std::memcpy(kappa.data() + row, src + (static_cast<size_t>(k) * nb_cells[1] + j) * nb_cells[0], sizeof(float) * nb_cells[0]);
          }
      }
      d_kappa.Ensure(kappa.size());
      Check(hipMemcpy(d_kappa.ptr, kappa.data(), kappa.size() * sizeof(float), hipMemcpyHostToDevice), "grid upload");
    }
    // coordinates
    const double *xf[3] = {global_xf[0].data(), global_xf[1].data(), global_xf[2].data()};
    const double *xv[3] = {global_xv[0].data(), global_xv[1].data(), global_xv[2].data()};
    std::vector<double> coords;
    size_t off_f[3], off_v[3];
    for (int a = 0; a < 3; a++) {
      off_f[a] = coords.size();
      coords.insert(coords.end(), xf[a], xf[a] + n[a] + 1);
      off_v[a] = coords.size();
      coords.insert(coords.end(), xv[a], xv[a] + n[a]);
    }
    ctx->d_coords.Ensure(coords.size());
    Check(hipMemcpy(ctx->d_coords.ptr, coords.data(), coords.size() * sizeof(double), hipMemcpyHostToDevice), "coordinate upload");
    // bucket tables for the cell search
    if (n_i > 65535 || n_j > 65535 || n_k > 65535)
      throw Failure{BL_E_UNSUPPORTED, "More than 65535 cells along an axis of the merged grid."};
    std::vector<unsigned short> buckets;
    size_t off_b[3];
    BlGridDevice dev{};
    for (int a = 0; a < 3; a++) {
      int n_bucket = std::max(512, 8 * n[a]);
      std::vector<int> table;
      BuildBuckets(xf[a], n[a], n_bucket, &table, &dev.bucket_x0[a], &dev.bucket_inv_w[a]);
      {   // evenly spaced faces?
        const double width = (xf[a][n[a]] - xf[a][0]) / n[a];
        bool even = width > 0.0;
        for (int c = 0; c <= n[a] && even; c++) even = std::abs(xf[a][c] - (xf[a][0] + c * width)) <= 1.0e-4 * width;
        dev.cell_x0[a] = xf[a][0];
        dev.cell_inv_w[a] = even ? 1.0 / width : 0.0;
        if (even) dev.uniform_mask |= 1 << a;
      }
      if (a == 0) {   // radial faces evenly spaced in log r? (bl_shade_fused2_kernel guesses the cell from log2 r and confirms it by the faces)
        dev.r_face_in = xf[0][0];
        dev.r_face_out = xf[0][n[0]];
        bool even = xf[0][0] > 0.0 && xf[0][n[0]] > xf[0][0];
        const double l0 = even ? std::log2(xf[0][0]) : 0.0, width = even ? (std::log2(xf[0][n[0]]) - l0) / n[0] : 1.0;
        for (int c = 0; c <= n[0] && even; c++) even = std::abs(std::log2(xf[0][c]) - (l0 + c * width)) <= 1.0e-4 * width;
        dev.log_uniform = even ? 1 : 0;
        dev.log_l0 = static_cast<float>(l0);
        dev.log_inv_w = static_cast<float>(1.0 / width);
      }
      dev.n_bucket[a] = n_bucket;
      off_b[a] = buckets.size();
      buckets.insert(buckets.end(), table.begin(), table.end());
    }
    // theta from pole to pole and phi all the way round: no sample is off the grid in angle
    dev.full_sphere = (xf[1][0] <= 0.0 && xf[1][n[1]] >= kPi && xf[2][0] <= 0.0 && xf[2][n[2]] >= 2.0 * kPi) ? 1 : 0;
    ctx->d_buckets.Ensure(buckets.size());
    Check(hipMemcpy(ctx->d_buckets.ptr, buckets.data(), buckets.size() * sizeof(unsigned short), hipMemcpyHostToDevice), "bucket upload");
    dev.cells = d_cells.ptr;
    dev.kappa = code_kappa ? d_kappa.ptr : nullptr;
    dev.n_blocks = 0;
    dev.stride_row = n_i;
    dev.stride_plane = n_i * n_j;
    for (int a = 0; a < 3; a++) {
      dev.xf[a] = ctx->d_coords.ptr + off_f[a];
      dev.xv[a] = ctx->d_coords.ptr + off_v[a];
      dev.bucket[a] = ctx->d_buckets.ptr + off_b[a];
      dev.n[a] = n[a];
      dev.nb[a] = nb_cells[a];
    }
    if (ctx->params.simulation_coord == BL_COORD_FMKS) {
      // FMKS grid (simulation_sampling.cpp:66-73): native coordinates above, plus the reader's map and bounds
      if (n_b != 1 || g->sks_map == nullptr || g->sks_map_n1 < 2 || g->sks_map_n2 < 2)
        throw Failure{BL_E_ARG, "simulation_coord = fmks needs a single block and the reader's sks_map in bl_grid_desc."};
      const size_t map_count = static_cast<size_t>(2) * g->sks_map_n1 * g->sks_map_n2;
      ctx->d_sks_map.Ensure(map_count);
      Check(hipMemcpy(ctx->d_sks_map.ptr, g->sks_map, map_count * sizeof(double), hipMemcpyHostToDevice), "sks_map upload");
      dev.fmks = 1;
      dev.sks_map = ctx->d_sks_map.ptr;
      dev.sks_map_n1 = g->sks_map_n1;
      dev.sks_map_n2 = g->sks_map_n2;
      dev.sks_map_r_in = g->sks_map_r_in;
      dev.sks_map_dr = g->sks_map_dr;
      dev.sks_map_dtheta = g->sks_map_dtheta;
      for (int c = 0; c < 6; c++) dev.fmks_bounds[c] = g->simulation_bounds[c];
      dev.fmks_x1_0 = xf[0][0];
      dev.fmks_dx1 = xf[0][1] - xf[0][0];
      dev.fmks_dx2 = xf[1][1] - xf[1][0];
    }
    ctx->grid_dev = dev;
    {
      // The locate kernel stages the tables in LDS when they fit 60 KiB (up to ~640 cells per axis); larger grids
      // are searched in the same tables where they lie in HBM (lds_table_bytes = 0).
      size_t bytes = 0;
      for (int a = 0; a < 3; a++) bytes += (2 * static_cast<size_t>(n[a]) + 1) * sizeof(double) + static_cast<size_t>(dev.n_bucket[a]) * sizeof(unsigned short);
      ctx->lds_table_bytes = bytes > 60 * 1024 ? 0 : static_cast<int>((bytes + 15) / 16 * 16);
      if (ctx->lds_table_bytes == 0 && ctx->params.slow_light_on)
        throw Failure{BL_E_UNSUPPORTED, "Slow light on a grid whose coordinate tables exceed the 60 KiB LDS budget is not built."};
    }
    ctx->n_i = n_i;
    ctx->n_j = n_j;
    ctx->n_k = n_k;
}

// Where the search along n ascending faces starts (BlGridDevice::row_guess, box_guess): kind (1: logarithmic), origin, cells per unit.
// Logarithmic where every face is positive and the ratios of neighbouring faces agree better than their differences do.
void SearchGuess(const double *f, int n, double guess[3]) {
  bool positive = f[0] > 0.0;
  double ratio_lo = 1.0e300, ratio_hi = 0.0, step_lo = 1.0e300, step_hi = 0.0;
  for (int i = 0; i < n && positive; i++) {
    ratio_lo = std::min(ratio_lo, f[i + 1] / f[i]); ratio_hi = std::max(ratio_hi, f[i + 1] / f[i]);
    step_lo = std::min(step_lo, f[i + 1] - f[i]); step_hi = std::max(step_hi, f[i + 1] - f[i]);
  }
  const bool logarithmic = positive && n > 1 && (ratio_hi / ratio_lo - 1.0) < 0.5 * (step_hi / step_lo - 1.0);
  if (logarithmic) {
    guess[0] = 1.0;
    guess[1] = std::log2(f[0]);
    guess[2] = n / (std::log2(f[n]) - std::log2(f[0]));
  } else {
    guess[0] = 0.0;
    guess[1] = f[0];
    guess[2] = n / (f[n] - f[0]);
  }
}

// Mesh with refinement (blocks of several levels), or any other set of non-overlapping equal-sized blocks:
// cells stay block by block; the distinct block boundaries along each axis span a lattice of boxes, each
// covered by at most one block, from which the locate kernel finds the block of a sample (the reference
// scans all blocks per sample, simulation_sampling.cpp:352-394), then the cell from the block's own rows.
void UploadRefinedGrid(bl_ctx *ctx, const bl_grid_desc *g) {
  const int nb[3] = {g->n_i, g->n_j, g->n_k};
  const double *block_xf[3] = {g->x1f, g->x2f, g->x3f};
  const double *block_xv[3] = {g->x1v, g->x2v, g->x3v};
  const int n_b = g->n_blocks;
  const char *kBadMesh = "Multi-block grid has blocks that overlap or faces that do not ascend.";
  std::vector<double> edge[3];
  for (int a = 0; a < 3; a++) {
    if (block_xf[a] == nullptr || block_xv[a] == nullptr) throw Failure{BL_E_ARG, "Bad grid description."};
    for (int blk = 0; blk < n_b; blk++) {
      const double *f = block_xf[a] + static_cast<size_t>(blk) * (nb[a] + 1);
      for (int i = 0; i < nb[a]; i++)
        if (!(f[i] < f[i + 1])) throw Failure{BL_E_UNSUPPORTED, kBadMesh};
      edge[a].push_back(f[0]);
      edge[a].push_back(f[nb[a]]);
    }
    std::sort(edge[a].begin(), edge[a].end());
    edge[a].erase(std::unique(edge[a].begin(), edge[a].end()), edge[a].end());
  }
  const int n_edge[3] = {static_cast<int>(edge[0].size()) - 1, static_cast<int>(edge[1].size()) - 1,
                         static_cast<int>(edge[2].size()) - 1};
  const size_t n_boxes = static_cast<size_t>(n_edge[0]) * n_edge[1] * n_edge[2];
  const size_t block_cells = static_cast<size_t>(nb[0]) * nb[1] * nb[2];
  const size_t n_cells = block_cells * n_b;
  if (n_boxes > (1ull << 28) || n_cells >= (1ull << 32))
    throw Failure{BL_E_UNSUPPORTED, "Multi-block grid too large for the block lattice of this build."};
  std::vector<int> lattice(n_boxes, -1);
  for (int blk = 0; blk < n_b; blk++) {
    int lo[3], hi[3];
    for (int a = 0; a < 3; a++) {
      const double *f = block_xf[a] + static_cast<size_t>(blk) * (nb[a] + 1);
      lo[a] = static_cast<int>(std::lower_bound(edge[a].begin(), edge[a].end(), f[0]) - edge[a].begin());
      hi[a] = static_cast<int>(std::lower_bound(edge[a].begin(), edge[a].end(), f[nb[a]]) - edge[a].begin());
    }
    for (int kk = lo[2]; kk < hi[2]; kk++)
      for (int jj = lo[1]; jj < hi[1]; jj++)
        for (int ii = lo[0]; ii < hi[0]; ii++) {
          int &slot = lattice[(static_cast<size_t>(kk) * n_edge[1] + jj) * n_edge[0] + ii];
          if (slot != -1) throw Failure{BL_E_UNSUPPORTED, kBadMesh};
          slot = blk;
        }
  }
  // cells: [var][block][k][j][i] -> [block][k][j][i][8]
  const int order[8] = {g->ind_rho, g->ind_pgas, g->ind_uu1, g->ind_uu2, g->ind_uu3, g->ind_bb1, g->ind_bb2, g->ind_bb3};
  for (int v : order)
    if (v < 0 || v >= g->n_var) throw Failure{BL_E_ARG, "Grid variable index out of range."};
  DeviceBuffer<float> &d_cells = ctx->cells_target != nullptr ? *ctx->cells_target : ctx->d_cells;
  DeviceBuffer<float> &d_kappa = ctx->kappa_target != nullptr ? *ctx->kappa_target : ctx->d_kappa;
  UploadCells(ctx, g, order, n_cells, d_cells, nb, nullptr, 0, 0);   // (the blocks stay one behind the other: the caller's order)
  const bool code_kappa = ctx->params.plasma_model == BL_PLASMA_CODE_KAPPA;
  if (code_kappa) {
    if (g->ind_kappa < 0 || g->ind_kappa >= g->n_var) throw Failure{BL_E_ARG, "Grid variable index out of range."};
    d_kappa.Ensure(n_cells);
    Check(hipMemcpy(d_kappa.ptr, g->prim + static_cast<size_t>(g->ind_kappa) * n_cells, n_cells * sizeof(float),
                    hipMemcpyHostToDevice), "grid upload");
  }
  // Coordinate rows, each distinct one once (faces and centres of a block along an axis, compared bit for bit: blocks of one level at
  // one position share them), which row every block uses, the first centre of the next block of the file (what the reference reads one
  // past a block's last centre), then the block boundaries
  std::vector<double> coords;
  std::vector<int> block_row(static_cast<size_t>(n_b) * 3);
  size_t off_f[3], off_v[3], off_e[3], off_n[3], off_g[3];
  int n_rows[3];
  for (int a = 0; a < 3; a++) {
    const size_t row_len = static_cast<size_t>(nb[a]) * 2 + 1;
    std::map<std::vector<unsigned long long>, int> seen;   // (rows compared word for word: the same bits, not merely equal values)
    std::vector<const double *> row_f, row_v;
    for (int blk = 0; blk < n_b; blk++) {
      const double *f = block_xf[a] + static_cast<size_t>(blk) * (nb[a] + 1), *v = block_xv[a] + static_cast<size_t>(blk) * nb[a];
      std::vector<unsigned long long> key(row_len);
      std::memcpy(key.data(), f, (static_cast<size_t>(nb[a]) + 1) * sizeof(double));
      std::memcpy(key.data() + nb[a] + 1, v, static_cast<size_t>(nb[a]) * sizeof(double));
      auto found = seen.find(key);
      if (found == seen.end()) {
        found = seen.emplace(std::move(key), static_cast<int>(row_f.size())).first;
        row_f.push_back(f);
        row_v.push_back(v);
      }
      block_row[static_cast<size_t>(a) * n_b + blk] = found->second;
    }
    n_rows[a] = static_cast<int>(row_f.size());
    off_f[a] = coords.size();
    for (const double *f : row_f) coords.insert(coords.end(), f, f + nb[a] + 1);
    off_v[a] = coords.size();
    for (const double *v : row_v) coords.insert(coords.end(), v, v + nb[a]);
    off_n[a] = coords.size();
    for (int blk = 0; blk < n_b; blk++)
      coords.push_back(blk + 1 < n_b ? block_xv[a][static_cast<size_t>(blk + 1) * nb[a]] : std::numeric_limits<double>::quiet_NaN());
    off_g[a] = coords.size();
    for (const double *f : row_f) {
      double guess[3];
      SearchGuess(f, nb[a], guess);
      coords.insert(coords.end(), guess, guess + 3);
    }
    off_e[a] = coords.size();
    coords.insert(coords.end(), edge[a].begin(), edge[a].end());
  }
  ctx->d_coords.Ensure(coords.size());
  Check(hipMemcpy(ctx->d_coords.ptr, coords.data(), coords.size() * sizeof(double), hipMemcpyHostToDevice), "coordinate upload");
  // (the lattice, then the blocks' rows)
  const size_t lattice_ints = lattice.size();
  lattice.insert(lattice.end(), block_row.begin(), block_row.end());
  ctx->d_lattice.Ensure(lattice.size());
  Check(hipMemcpy(ctx->d_lattice.ptr, lattice.data(), lattice.size() * sizeof(int), hipMemcpyHostToDevice), "lattice upload");
  BlGridDevice dev{};
  dev.cells = d_cells.ptr;
  dev.kappa = code_kappa ? d_kappa.ptr : nullptr;
  dev.n_blocks = n_b;
  dev.lattice = ctx->d_lattice.ptr;
  dev.stride_row = nb[0];
  dev.stride_plane = nb[0] * nb[1];
  for (int a = 0; a < 3; a++) {
    dev.bxf[a] = ctx->d_coords.ptr + off_f[a];
    dev.bxv[a] = ctx->d_coords.ptr + off_v[a];
    dev.xv_next[a] = ctx->d_coords.ptr + off_n[a];
    dev.row_guess[a] = ctx->d_coords.ptr + off_g[a];
    dev.block_row[a] = ctx->d_lattice.ptr + lattice_ints + static_cast<size_t>(a) * n_b;
    dev.n_rows[a] = n_rows[a];
    dev.edge[a] = ctx->d_coords.ptr + off_e[a];
    dev.n_edge[a] = n_edge[a];
    dev.edge_first[a] = edge[a].front();
    dev.edge_last[a] = edge[a].back();
    SearchGuess(edge[a].data(), n_edge[a], dev.box_guess[a]);
    dev.n[a] = nb[a];
    dev.nb[a] = nb[a];
  }
  if (ctx->params.simulation_interp && ctx->params.simulation_block_interp) {
    // MeshBlock table for FindNearbyInds (simulation_sampling.cpp:36-39, :84-93) and a hash from (level, location)
    // to block in place of its scans over all blocks
    // n_3_root only enters through the periodic seam of spherical coordinates (simulation_sampling.cpp:1181-1219)
    if (g->levels == nullptr || g->locations == nullptr || (g->n_3_root <= 0 && ctx->params.simulation_coord == BL_COORD_SKS))
      throw Failure{BL_E_ARG, "simulation_block_interp = true needs the MeshBlock table (levels, locations, n_3_root) in bl_grid_desc."};
    int max_level = 0;
    for (int blk = 0; blk < n_b; blk++) {
      if (g->levels[blk] < 0 || g->levels[blk] > 30) throw Failure{BL_E_ARG, "Bad MeshBlock level."};
      max_level = std::max(max_level, g->levels[blk]);
      for (int a = 0; a < 3; a++)
        if (g->locations[3 * blk + a] < 0 || g->locations[3 * blk + a] >= (1 << 19)) throw Failure{BL_E_UNSUPPORTED, "MeshBlock location outside the range of this build."};
    }
    unsigned int slots = 16;
    while (slots < 2u * static_cast<unsigned int>(n_b)) slots *= 2;
    std::vector<unsigned long long> keys(slots, ~0ull);
    std::vector<int> table(static_cast<size_t>(n_b) * 4 + slots, -1);
    for (int blk = 0; blk < n_b; blk++) {
      table[blk] = g->levels[blk];
      for (int a = 0; a < 3; a++) table[n_b + 3 * blk + a] = g->locations[3 * blk + a];
      const unsigned long long key = (static_cast<unsigned long long>(g->levels[blk]) << 57) | (static_cast<unsigned long long>(g->locations[3 * blk]) << 38)
          | (static_cast<unsigned long long>(g->locations[3 * blk + 1]) << 19) | static_cast<unsigned long long>(g->locations[3 * blk + 2]);
      unsigned int slot = static_cast<unsigned int>((key * 0x9e3779b97f4a7c15ull) >> 32) & (slots - 1);
      while (keys[slot] != ~0ull) {
        if (keys[slot] == key) throw Failure{BL_E_UNSUPPORTED, kBadMesh};   // two blocks at one place
        slot = (slot + 1) & (slots - 1);
      }
      keys[slot] = key;
      table[static_cast<size_t>(n_b) * 4 + slot] = blk;
    }
    ctx->d_block_table.Ensure(table.size());
    ctx->d_block_keys.Ensure(keys.size());
    Check(hipMemcpy(ctx->d_block_table.ptr, table.data(), table.size() * sizeof(int), hipMemcpyHostToDevice), "block table upload");
    Check(hipMemcpy(ctx->d_block_keys.ptr, keys.data(), keys.size() * sizeof(unsigned long long), hipMemcpyHostToDevice), "block table upload");
    dev.block_interp = 1;
    dev.levels = ctx->d_block_table.ptr;
    dev.locations = ctx->d_block_table.ptr + n_b;
    dev.hash_blocks = ctx->d_block_table.ptr + static_cast<size_t>(n_b) * 4;
    dev.hash_keys = ctx->d_block_keys.ptr;
    dev.hash_mask = slots - 1;
    dev.max_level = max_level;
    dev.n_3_level0 = g->n_3_root / nb[2];
  }
  {
    // what bl_locate_kernel<kRefined> stages in LDS when it fits (four workgroups to a compute unit up to 36 KiB, one of 1 024 lanes beyond): block boundaries,
    // lattice, rows, the blocks' rows and next centres, and with inter-block interpolation the MeshBlock table and its hash
    size_t doubles = 0, ints = lattice_ints + static_cast<size_t>(n_b) * 3;
    for (int a = 0; a < 3; a++) doubles += static_cast<size_t>(n_edge[a]) + 1 + static_cast<size_t>(n_rows[a]) * (2 * nb[a] + 1 + 3) + n_b;
    if (dev.block_interp) {
      doubles += dev.hash_mask + 1;                                        // (the keys: 8 bytes each)
      ints += static_cast<size_t>(n_b) * 4 + dev.hash_mask + 1;
    }
    const size_t bytes = doubles * sizeof(double) + (ints + 3) / 4 * 4 * sizeof(int);
    dev.refined_lds_bytes = bytes <= static_cast<size_t>(BL_LOCATE_REFINED_LDS) ? static_cast<int>(bytes) : 0;
  }
  {
    // what bl_shade_fused2_kernel<..., kRefined> asks of a mesh (bl_shade_fused.hip): boxes and rows evenly spaced in log r / theta / phi to
    // 1e-4 of a cell (its guesses are then right wherever its margins let it decide), the sphere covered without a hole, 32-bit byte
    // offsets into the cell array, room in LDS for the row chunks and one descriptor per box (one 512-lane workgroup to a compute unit)
    auto evenly = [](const double *f, int n, bool logarithmic, double *origin, double *inv_w) {
      if (n < 1 || (logarithmic && !(f[0] > 0.0))) return false;
      const double first = logarithmic ? std::log2(f[0]) : f[0], last = logarithmic ? std::log2(f[n]) : f[n];
      const double width = (last - first) / n;
      if (!(width > 0.0)) return false;
      for (int c = 0; c <= n; c++)
        if (!(std::abs((logarithmic ? std::log2(f[c]) : f[c]) - (first + c * width)) <= 1.0e-4 * width)) return false;
      *origin = first;
      *inv_w = 1.0 / width;
      return true;
    };
    bool ok = nb[0] >= 2 && nb[1] >= 2 && nb[2] >= 2 && n_cells * 32ull < (1ull << 32) && static_cast<unsigned long long>(nb[1]) * nb[2] < (1ull << 24);
    for (size_t box = 0; box < n_boxes && ok; box++) ok = lattice[box] >= 0;
    ok = ok && edge[1].front() <= 0.0 && edge[1].back() >= kPi && edge[2].front() <= 0.0 && edge[2].back() >= 2.0 * kPi;
    double origin[3] = {0.0, 0.0, 0.0}, inv_w[3] = {0.0, 0.0, 0.0};
    for (int a = 0; a < 3 && ok; a++) ok = evenly(edge[a].data(), n_edge[a], a == 0, &origin[a], &inv_w[a]);
    size_t chunk_at[3] = {0, 0, 0}, bytes = 48 * sizeof(double);
    for (int a = 0; a < 3 && ok; a++) {
      chunk_at[a] = bytes - 48 * sizeof(double);   // (relative to the first chunk: the kernel adds where that lies)
      for (int q = 0; q < n_rows[a] && ok; q++) {
        const double *guess = coords.data() + off_g[a] + 3 * static_cast<size_t>(q);
        double row_origin, row_inv_w;
        ok = (guess[0] != 0.0) == (a == 0) && evenly(coords.data() + off_f[a] + static_cast<size_t>(q) * (nb[a] + 1), nb[a], a == 0, &row_origin, &row_inv_w);
      }
      bytes += static_cast<size_t>(n_rows[a]) * (16 + 64 * static_cast<size_t>(nb[a]));
    }
    bytes += 16 * n_boxes;
    ok = ok && bytes <= static_cast<size_t>(BL_FUSED_REFINED_LDS);
    if (ok) {
      std::vector<unsigned int> desc(4 * n_boxes);
      for (size_t box = 0; box < n_boxes; box++) {
        const int blk = lattice[box];
        desc[4 * box] = static_cast<unsigned int>(static_cast<size_t>(blk) * block_cells * 32);
        for (int a = 0; a < 3; a++)
          desc[4 * box + 1 + a] = static_cast<unsigned int>(chunk_at[a] + static_cast<size_t>(block_row[static_cast<size_t>(a) * n_b + blk]) * (16 + 64 * static_cast<size_t>(nb[a])));
      }
      ctx->d_fused_desc.Ensure(desc.size());
      Check(hipMemcpy(ctx->d_fused_desc.ptr, desc.data(), desc.size() * sizeof(unsigned int), hipMemcpyHostToDevice), "lattice upload");
      dev.fused_desc = ctx->d_fused_desc.ptr;
      dev.fused_lds_bytes = static_cast<int>(bytes);
      dev.box_l0 = static_cast<float>(origin[0]);
      dev.box_linv = static_cast<float>(inv_w[0]);
      for (int a = 1; a < 3; a++) {
        dev.box_x0[a - 1] = origin[a];
        dev.box_inv_w[a - 1] = inv_w[a];
      }
      dev.r_face_in = edge[0].front();
      dev.r_face_out = edge[0].back();
    }
  }
  ctx->grid_dev = dev;
  ctx->lds_table_bytes = 0;
  ctx->n_i = nb[0];
  ctx->n_j = nb[1];
  ctx->n_k = nb[2];
}
}  // namespace

namespace {
// Hash (FNV-1a, 64 bits) of everything about a grid that decides WHERE a sample lies on it: the block table, the coordinate arrays,
// the FMKS look-up table. Two bl_set_grid calls with the same hash hand over the same geometry (the cells may differ): located
// samples of the first stay valid for the second (the reference's `first_time`, radiation_integrator.cpp:693-704).
unsigned long long GridGeometryHash(const bl_grid_desc *g) {
  unsigned long long h = 1469598103934665603ull;
  auto mix = [&h](const void *data, size_t bytes) {
    const unsigned char *p = static_cast<const unsigned char *>(data);
    // (eight bytes a step where they are there: coordinate arrays are doubles)
    size_t at = 0;
    for (; at + 8 <= bytes; at += 8) {
      unsigned long long word;
      std::memcpy(&word, p + at, 8);
      h = (h ^ word) * 1099511628211ull;
      h ^= h >> 29;
    }
    for (; at < bytes; at++) h = (h ^ p[at]) * 1099511628211ull;
  };
  const int32_t dims[5] = {g->n_blocks, g->n_i, g->n_j, g->n_k, g->n_3_root};
  mix(dims, sizeof dims);
  const size_t n_b = static_cast<size_t>(g->n_blocks);
  mix(g->x1f, n_b * (g->n_i + 1) * sizeof(double)); mix(g->x1v, n_b * g->n_i * sizeof(double));
  mix(g->x2f, n_b * (g->n_j + 1) * sizeof(double)); mix(g->x2v, n_b * g->n_j * sizeof(double));
  mix(g->x3f, n_b * (g->n_k + 1) * sizeof(double)); mix(g->x3v, n_b * g->n_k * sizeof(double));
  if (g->levels != nullptr) mix(g->levels, n_b * sizeof(int32_t));
  if (g->locations != nullptr) mix(g->locations, n_b * 3 * sizeof(int32_t));
  if (g->sks_map != nullptr) {
    const int32_t map_dims[2] = {g->sks_map_n1, g->sks_map_n2};
    mix(map_dims, sizeof map_dims);
    mix(g->sks_map, 2 * static_cast<size_t>(g->sks_map_n1) * g->sks_map_n2 * sizeof(double));
    const double scalars[3] = {g->sks_map_r_in, g->sks_map_dr, g->sks_map_dtheta};
    mix(scalars, sizeof scalars);
    mix(g->simulation_bounds, sizeof g->simulation_bounds);
  }
  return h != 0 ? h : 1;
}
}  // namespace

int bl_set_grid(bl_ctx *ctx, const bl_grid_desc *g) {
  if (ctx == nullptr || g == nullptr) return BL_E_ARG;
  try {
    if (ctx->device == BL_DEVICE_NONE) throw Failure{BL_E_DEVICE, "Host-only context: no HIP device selected (the hot path has no CPU fallback)."};
    if (ctx->params.model_type != BL_MODEL_SIMULATION) throw Failure{BL_E_STATE, "bl_set_grid called in formula mode."};
    if (ctx->params.slow_light_on && ctx->cells_target == nullptr)
      throw Failure{BL_E_STATE, "slow_light_on = true: hand the time slices over with bl_set_grid_slice (or bl_slow_light_read)."};
    if (g->n_blocks < 1) throw Failure{BL_E_ARG, "Bad grid description."};
    if (g->n_i < 2 || g->n_j < 2 || g->n_k < 2 || g->prim == nullptr) throw Failure{BL_E_ARG, "Bad grid description."};
    Check(hipSetDevice(ctx->device), "hipSetDevice");
    // The next snapshot of a series - the geometry in place, to the bit, and the same variables: only the cells are new. They go up on
    // a stream of their own into the second cell array while a render of this context may be running on another host thread, and the
    // arrays change places once that render has ended. (With electron entropy from the grid there is a second array to double: the
    // long way.)
    const bool code_kappa = ctx->params.plasma_model == BL_PLASMA_CODE_KAPPA;
    const int order_now[8] = {g->ind_rho, g->ind_pgas, g->ind_uu1, g->ind_uu2, g->ind_uu3, g->ind_bb1, g->ind_bb2, g->ind_bb3};
    if (ctx->have_grid && ctx->cells_target == nullptr && ctx->placement.valid && !code_kappa && ctx->grid_geometry != 0 && GridGeometryHash(g) == ctx->grid_geometry
        && std::equal(order_now, order_now + 8, ctx->placement.order) && g->n_var == ctx->grid_meta.n_var
        && g->plasma_gamma == ctx->grid_meta.plasma_gamma && g->plasma_gamma_i == ctx->grid_meta.plasma_gamma_i && g->plasma_gamma_e == ctx->grid_meta.plasma_gamma_e) {
      if (ctx->stream_upload == nullptr) Check(hipStreamCreateWithFlags(&ctx->stream_upload, hipStreamNonBlocking), "hipStreamCreate");
      UploadCellsAs(ctx, g, ctx->d_cells_back, ctx->stream_upload, false);
      std::lock_guard<std::mutex> guard(ctx->render_lock);
      std::swap(ctx->d_cells, ctx->d_cells_back);
      ctx->grid_dev.cells = ctx->d_cells.ptr;
      ctx->grid_meta = *g;
      return BL_OK;
    }
    std::lock_guard<std::mutex> guard(ctx->render_lock);   // (everything about the grid changes: not beside a render)
    ctx->have_grid = false;   // a failed upload leaves no grid behind (the previous one may be half overwritten)
    ctx->grid_geometry = 0;
    ctx->placement.valid = false;
    if (ctx->params.simulation_interp && ctx->params.simulation_block_interp) {
      UploadRefinedGrid(ctx, g);   // inter-block interpolation works on the MeshBlocks as they are
    } else {
      try {
        UploadMergedGrid(ctx, g);
      } catch (const Failure &failure) {
        if (failure.message != kIrregular) throw;
        UploadRefinedGrid(ctx, g);   // blocks of several levels, or a tiling with holes
      }
    }
    ctx->grid_meta = *g;
    if (ctx->cells_target == nullptr) ctx->grid_geometry = GridGeometryHash(g);   // (slow light locates per snapshot: no located samples are kept)
    ctx->grid_outer_x1 = g->x1f[g->n_i];
    for (int blk = 1; blk < g->n_blocks; blk++)
      ctx->grid_outer_x1 = std::max(ctx->grid_outer_x1, g->x1f[static_cast<size_t>(blk) * (g->n_i + 1) + g->n_i]);
    ctx->have_grid = true;
  } catch (const Failure &failure) {
    return Fail(ctx, failure);
  }
  return BL_OK;
}

int bl_set_grid_slice(bl_ctx *ctx, int slice, const bl_grid_desc *g, double time) {
  if (ctx == nullptr || g == nullptr) return BL_E_ARG;
  try {
    const bl_params &p = ctx->params;
    if (p.model_type != BL_MODEL_SIMULATION || !p.slow_light_on) throw Failure{BL_E_STATE, "bl_set_grid_slice needs slow_light_on = true."};
    if (slice < 0 || slice >= p.slow_chunk_size) throw Failure{BL_E_ARG, "Time slice index outside slow_chunk_size."};
    ctx->slow_slices.resize(p.slow_chunk_size);
    bl_ctx::SlowSlice &target = ctx->slow_slices[slice];
    ctx->cells_target = &target.cells;
    ctx->kappa_target = &target.kappa;
    const int rc = bl_set_grid(ctx, g);
    ctx->cells_target = nullptr;
    ctx->kappa_target = nullptr;
    if (rc != BL_OK) return rc;
    target.time = time;
    target.set = true;
  } catch (const Failure &failure) {
    ctx->cells_target = nullptr;
    ctx->kappa_target = nullptr;
    return Fail(ctx, failure);
  }
  return BL_OK;
}

int bl_shift_grid_slices(bl_ctx *ctx, int count) {
  if (ctx == nullptr) return BL_E_ARG;
  const int chunk = static_cast<int>(ctx->slow_slices.size());
  if (count < 0 || count > chunk) return Fail(ctx, Failure{BL_E_ARG, "Bad slice shift."});
  // prim[n].Swap(prim[n - count]) for n = chunk - 1 ... count (simulation_reader.cpp:289-296)
  for (int n = chunk - 1; n >= count && count > 0; n--) std::swap(ctx->slow_slices[n], ctx->slow_slices[n - count]);
  return BL_OK;
}

int bl_set_snapshot(bl_ctx *ctx, int snapshot) {
  if (ctx == nullptr || snapshot < 0) return BL_E_ARG;
  ctx->snapshot = snapshot;
  return BL_OK;
}

int bl_render_num_images(const bl_ctx *ctx) { return ctx != nullptr ? ctx->render_num_images : 0; }

int bl_image_num_quantities(const bl_ctx *ctx) { return ctx != nullptr ? ctx->image_num_quantities : -1; }

int bl_camera_frame_get(const bl_ctx *ctx, bl_camera_frame *out) {
  if (ctx == nullptr || out == nullptr) return BL_E_ARG;
  *out = ctx->frame;
  return BL_OK;
}

int bl_frequencies(const bl_ctx *ctx, double *out, int n) {
  if (ctx == nullptr || out == nullptr) return BL_E_ARG;
  for (int l = 0; l < n && l < static_cast<int>(ctx->frequencies.size()); l++) out[l] = ctx->frequencies[l];
  return BL_OK;
}

void *bl_host_alloc(bl_ctx *ctx, size_t bytes) {
  if (ctx == nullptr || ctx->device == BL_DEVICE_NONE || bytes == 0) return nullptr;
  void *p = nullptr;
  if (hipSetDevice(ctx->device) != hipSuccess || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}

void bl_host_free(bl_ctx *ctx, void *p) {
  if (ctx == nullptr || p == nullptr || ctx->device == BL_DEVICE_NONE) return;
  (void)hipSetDevice(ctx->device);
  (void)hipHostFree(p);
}

int bl_set_geodesic_reuse(bl_ctx *ctx, int on) {
  if (ctx == nullptr) return BL_E_ARG;
  std::lock_guard<std::mutex> guard(ctx->render_lock);   // (not beside a render of this context)
  ctx->geodesic_reuse = on ? 1 : 0;
  if (!on && ctx->device != BL_DEVICE_NONE) {   // what was kept goes back to the device
    (void)hipSetDevice(ctx->device);
    ctx->resident.valid = ctx->resident.located_valid = false;
    ctx->resident.parked = false;
    ctx->resident.store.Free();
  }
  return BL_OK;
}

int bl_set_overlap(bl_ctx *ctx, int on) {
  if (ctx == nullptr) return BL_E_ARG;
  ctx->overlap_chunks = on ? 1 : 0;
  return BL_OK;
}

int bl_debug_set_guard_band(bl_ctx *ctx, double relative_width) {
  if (ctx == nullptr || !(relative_width >= 0.0)) return BL_E_ARG;
  ctx->guard_band = relative_width;
  return BL_OK;
}

int bl_device_count(void) {
  int count = 0;
  return hipGetDeviceCount(&count) == hipSuccess ? count : 0;
}

int bl_set_undefined_policy(bl_ctx *ctx, int policy) {
  if (ctx == nullptr || policy < 0 || policy > (BL_UNDEFINED_EDGE | BL_UNDEFINED_KAPPA)) return BL_E_ARG;
  ctx->undefined_policy = policy;
  return BL_OK;
}

int bl_set_arithmetic(bl_ctx *ctx, int mode) {
  if (ctx == nullptr || (mode != BL_ARITH_EXACT && mode != BL_ARITH_TOLERANT)) return BL_E_ARG;
  ctx->arithmetic = mode;
  return BL_OK;
}

int bl_set_reproducible(bl_ctx *ctx, int on) {
  if (ctx == nullptr) return BL_E_ARG;
  ctx->reproducible = on ? 1 : 0;
  return BL_OK;
}

int bl_set_tail_policy(bl_ctx *ctx, int policy) {
  if (ctx == nullptr || policy < BL_TAIL_AUTO || policy > BL_TAIL_SPLIT) return BL_E_ARG;
  ctx->tail_policy = policy;
  return BL_OK;
}

int bl_set_caller_stream(bl_ctx *ctx, void *stream, int enabled) {
  if (ctx == nullptr) return BL_E_ARG;
  ctx->caller_stream = static_cast<hipStream_t>(stream);
  ctx->caller_stream_set = enabled != 0;
  return BL_OK;
}

int bl_debug_set_switches(bl_ctx *ctx, uint32_t switches) {
  if (ctx == nullptr) return BL_E_ARG;
  ctx->switches = switches;
  return BL_OK;
}

int bl_set_scratch_limit(bl_ctx *ctx, uint64_t bytes) {
  if (ctx == nullptr || bytes < (1ull << 20)) return BL_E_ARG;
  ctx->scratch_limit = bytes;
  return BL_OK;
}

int bl_debug_math(bl_ctx *ctx, int op, int64_t n, const double *x, const double *y, double *out) {
  if (ctx == nullptr || x == nullptr || out == nullptr || n <= 0) return BL_E_ARG;
  try {
    if (ctx->device == BL_DEVICE_NONE) throw Failure{BL_E_DEVICE, "Host-only context: no HIP device selected."};
    Check(hipSetDevice(ctx->device), "hipSetDevice");
    EnsureStreams(ctx);
    DeviceBuffer<double> dx, dy, dout;
    dx.Ensure(n);
    dout.Ensure(n);
    Check(hipMemcpy(dx.ptr, x, n * sizeof(double), hipMemcpyHostToDevice), "upload");
    if (y != nullptr) {
      dy.Ensure(n);
      Check(hipMemcpy(dy.ptr, y, n * sizeof(double), hipMemcpyHostToDevice), "upload");
    }
    hipError_t err = bl_launch_debug_math(op, n, dx.ptr, y != nullptr ? dy.ptr : nullptr, dout.ptr, ctx->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
    if (err == hipSuccess) err = hipMemcpy(out, dout.ptr, n * sizeof(double), hipMemcpyDeviceToHost);
    dx.Free(); dy.Free(); dout.Free();
    Check(err, "bl_debug_math");
  } catch (const Failure &failure) {
    return Fail(ctx, failure);
  }
  return BL_OK;
}

int bl_get_stats(const bl_ctx *ctx, bl_stats *out) {
  if (ctx == nullptr || out == nullptr) return BL_E_ARG;
  *out = ctx->stats;
  return BL_OK;
}

const char *bl_last_error(const bl_ctx *ctx) { return ctx != nullptr ? ctx->last_error.c_str() : ""; }
const char *bl_last_global_error(void) { return g_global_error.c_str(); }
const char *bl_warnings(const bl_ctx *ctx) { return ctx != nullptr ? ctx->warnings.c_str() : ""; }

void bl_warnings_clear(bl_ctx *ctx) {
  if (ctx != nullptr) ctx->warnings.clear();
}

void bl_free(bl_ctx *ctx) {
  if (ctx == nullptr) return;
  if (ctx->device == BL_DEVICE_NONE) {
    delete ctx;
    return;
  }
  (void)hipSetDevice(ctx->device);
  ctx->d_cells.Free(); ctx->d_kappa.Free(); ctx->d_coords.Free(); ctx->d_buckets.Free(); ctx->slot[0].Free(); ctx->slot[1].Free();
  ctx->d_freq.Free(); ctx->d_pixel_map.Free();
  ctx->d_block_locs.Free(); ctx->d_tile_order.Free(); ctx->d_render_params.Free(); ctx->d_render.Free(); ctx->d_shade_cold.Free(); ctx->d_image.Free(); ctx->d_camera_pos.Free(); ctx->d_camera_dir.Free();
  ctx->d_out_sample_num.Free(); ctx->d_out_flags.Free();
  for (auto &e : ctx->events)
    if (e != nullptr) (void)hipEventDestroy(e);
  if (ctx->host_counters != nullptr) (void)hipHostFree(ctx->host_counters);
  if (ctx->caller_event != nullptr) (void)hipEventDestroy(ctx->caller_event);
  if (ctx->stream != nullptr) (void)hipStreamDestroy(ctx->stream);
  if (ctx->stream_geo != nullptr) (void)hipStreamDestroy(ctx->stream_geo);
  if (ctx->stream_upload != nullptr) (void)hipStreamDestroy(ctx->stream_upload);
  // (stream_few / stream_most are the process's, borrowed: EnsureSplitStreams)
  delete ctx;
}

const char *bl_build_info(void) { return "blacklight_amd;hip;gfx950;fp-contract=off"; }

}  // extern "C"

const bl_params *bl_internal_params(const bl_ctx *ctx) { return &ctx->params; }
bl_slow_state *bl_internal_slow_state(bl_ctx *ctx) { return &ctx->slow_state; }
void bl_internal_warn(bl_ctx *ctx, const char *message) { Warn(ctx, message); }
const bl_camera_frame *bl_internal_frame(const bl_ctx *ctx) { return &ctx->frame; }
const double *bl_internal_frequencies(const bl_ctx *ctx, int *count) {
  *count = static_cast<int>(ctx->frequencies.size());
  return ctx->frequencies.data();
}
int bl_internal_fail(bl_ctx *ctx, int code, const char *message) {
  ctx->last_error = std::string("Error: ") + message + "\n";
  return code;
}

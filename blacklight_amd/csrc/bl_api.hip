// bl_api.hip - C-ABI of the MI355X hot path (include/blacklight_amd.h): context, parameter
// validation in the reference's words, grid repack + upload, chunked kernel pipeline, statistics.
//
// Host-side counterpart of the reference's GeodesicIntegrator / RadiationIntegrator constructors
// (src/geodesic_integrator/geodesic_integrator.cpp:23-157, src/radiation_integrator/
// radiation_integrator.cpp:26-541) and of their Integrate() drivers, for the configurations in the
// hot-path scope. Configurations outside it are rejected loudly (BL_E_UNSUPPORTED); nothing falls
// back to a CPU path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/blacklight_amd.h"
#include "bl_bessel.h"
#include "bl_camera.h"
#include "bl_device.h"
#include "bl_internal.h"

extern "C" hipError_t bl_launch_geodesic(const BlTraceArgs *args, int integrator, int grid, hipStream_t stream);
extern "C" int bl_geodesic_occupancy(int integrator, int with_time, int spin_zero);
extern "C" hipError_t bl_launch_locate(const BlShadeArgs *args, int grid, int lds_bytes, hipStream_t stream);
extern "C" hipError_t bl_launch_shade(const BlShadeArgs *args, int model, int grid, hipStream_t stream);
extern "C" hipError_t bl_launch_shade_fast(const BlShadeArgs *args, int grid, hipStream_t stream);
extern "C" hipError_t bl_launch_polarized_coefficients(const BlShadeArgs *args, int grid, hipStream_t stream);
extern "C" hipError_t bl_launch_transfer(const BlTransferArgs *args, hipStream_t stream);
extern "C" hipError_t bl_launch_transfer_freq(const BlTransferArgs *args, hipStream_t stream);
extern "C" hipError_t bl_launch_coefficients_freq(const BlShadeArgs *args, int grid, hipStream_t stream);
extern "C" hipError_t bl_launch_transfer_aux(const BlTransferArgs *args, hipStream_t stream);
extern "C" hipError_t bl_launch_transfer_polarized(const BlTransferArgs *args, hipStream_t stream);
extern "C" hipError_t bl_launch_transfer_polarized_matrix(const BlTransferArgs *args, int num_cus, hipStream_t stream);
extern "C" hipError_t bl_launch_debug_math(int op, long long n, const double *x, const double *y, double *out, hipStream_t stream);

namespace {

constexpr double kPi = 3.141592653589793;
constexpr double kC = 2.99792458e10;
constexpr double kGGMsun = 1.32712440018e26;
constexpr double kMp = 1.67262192369e-24;
constexpr int kNumCellValues = 7;

thread_local std::string g_global_error;

struct Failure {
  int code;
  std::string message;
};

// Owning HBM allocation. Freed by the destructor (bl_free selects the context's device before it deletes the
// context, so every buffer a context holds - scratch sets, grid, time slices, block tables - goes back to the
// device); movable (the slow-light window swaps slices), not copyable.
template <typename T>
struct DeviceBuffer {
  T *ptr = nullptr;
  size_t count = 0;
  DeviceBuffer() = default;
  DeviceBuffer(const DeviceBuffer &) = delete;
  DeviceBuffer &operator=(const DeviceBuffer &) = delete;
  DeviceBuffer(DeviceBuffer &&other) noexcept : ptr(other.ptr), count(other.count) {
    other.ptr = nullptr;
    other.count = 0;
  }
  DeviceBuffer &operator=(DeviceBuffer &&other) noexcept {
    if (this != &other) {
      Free();
      ptr = other.ptr;
      count = other.count;
      other.ptr = nullptr;
      other.count = 0;
    }
    return *this;
  }
  ~DeviceBuffer() { Free(); }
  void Free() {
    if (ptr != nullptr) (void)hipFree(ptr);
    ptr = nullptr;
    count = 0;
  }
  void Ensure(size_t n) {
    if (n <= count) return;
    Free();
    hipError_t err = hipMalloc(reinterpret_cast<void **>(&ptr), n * sizeof(T));
    if (err != hipSuccess)
      throw Failure{BL_E_DEVICE, std::string("hipMalloc of ") + std::to_string(n * sizeof(T)) + " bytes failed: " + hipGetErrorString(err)};
    count = n;
  }
};

void Check(hipError_t err, const char *what) {
  if (err != hipSuccess) throw Failure{BL_E_DEVICE, std::string(what) + ": " + hipGetErrorString(err)};
}

}  // namespace

struct bl_ctx {
  bl_params params;
  bl_camera_frame frame;
  BlSpacetime st;
  std::vector<double> frequencies;
  std::string last_error, warnings;
  int device = 0;
  int num_cus = 256;
  hipStream_t stream = nullptr;       // shading stream: locate, coefficient and transfer kernels, uploads
  hipStream_t stream_geo = nullptr;   // geodesic kernel of the next chunk, concurrent with the above
  std::vector<hipEvent_t> events;     // kEventsPerChunk per chunk of the last render + one set-up event
  unsigned long long *host_counters = nullptr;   // pinned, (BL_CNT_COUNT + 4) per chunk
  size_t host_counters_chunks = 0;
  uint64_t scratch_limit = 144ull << 30;
  int overlap_chunks = 0;             // bl_set_overlap(): geodesic kernel of chunk c + 1 beside the shading of chunk c
  int arithmetic = BL_ARITH_EXACT;    // bl_set_arithmetic()
  int undefined_policy = BL_UNDEFINED_REFUSE;   // bl_set_undefined_policy()
  double guard_band = 1.0e-9;         // tolerant tier: relative half-width around a cut threshold left to the exact kernel

  // image rows (radiation_integrator.cpp:436-520)
  int image_num_quantities = 0;
  BlAuxImages aux_images{};          // which image rows exist; .any = an auxiliary image or a rendering is requested
  int render_num_images = 0;         // false-colour renderings (0 in formula mode)
  DeviceBuffer<BlRenderDevice> d_render_params;
  DeviceBuffer<double> d_render;     // staging for host output
  double plasma_thermal_frac = 0.0;

  // grid
  bool have_grid = false;
  int n_i = 0, n_j = 0, n_k = 0;
  bl_grid_desc grid_meta{};
  DeviceBuffer<float> d_cells;
  DeviceBuffer<float> d_kappa;   // electron entropy per cell (plasma_model = code_kappa)
  DeviceBuffer<double> d_coords;   // x1f x1v x2f x2v x3f x3v packed
  DeviceBuffer<unsigned short> d_buckets;
  DeviceBuffer<int> d_lattice;   // refined mesh: box of the block-boundary lattice -> block
  DeviceBuffer<double> d_sks_map;   // simulation_coord = fmks: the reader's SKS -> FMKS look-up table
  DeviceBuffer<int> d_block_table;                  // inter-block interpolation: levels, locations, hash blocks
  DeviceBuffer<unsigned long long> d_block_keys;    // ... and hash keys
  BlGridDevice grid_dev{};
  int lds_table_bytes = 0;
  DeviceBuffer<float> *cells_target = nullptr, *kappa_target = nullptr;   // where the grid upload puts the cells

  // slow light: the reader's window of time slices (prim[n], time[n]; n = 0 latest) on one geometry
  struct SlowSlice {
    DeviceBuffer<float> cells, kappa;
    double time = 0.0;
    bool set = false;
  };
  std::vector<SlowSlice> slow_slices;
  bl_slow_state slow_state{};            // reader-side bookkeeping of bl_slow_light_read (bl_snapshot.cpp)
  bool polarized = false;                // image_light and image_polarization in simulation mode
  double power_pol[7] = {};              // polarized power-law constants (simulation_coefficients.cpp:67-80)
  int snapshot = 0;                      // index of the image being rendered (warning texts, camera time)
  long long stats_slow_count[4] = {0, 0, 0, 0};    // pixels needing extrapolation in the last render, by kind
  double stats_slow_val[4] = {0.0, 0.0, 0.0, 0.0};
  DeviceBuffer<unsigned long long> d_slow_table;   // cells pointers, kappa pointers, times, extrapolation maxima
  DeviceBuffer<unsigned int> d_ray_extrap;

  // per-chunk scratch, two sets: the geodesic kernel fills one while the shading kernels drain the other
  struct ChunkSlot {
    DeviceBuffer<BlSampleHot> d_records_hot;
    DeviceBuffer<BlSampleCold> d_records_cold;
    DeviceBuffer<BlLocated> d_located;
    DeviceBuffer<unsigned long long> d_located_tag;
    DeviceBuffer<double2> d_transfer;
    DeviceBuffer<double> d_ray_kt, d_ray_factor;
    DeviceBuffer<int> d_ray_sample_num;
    DeviceBuffer<unsigned char> d_ray_flags;
    DeviceBuffer<long long> d_ray_out_index;
    DeviceBuffer<unsigned long long> d_counters;   // BL_CNT_COUNT + 4 stats
    DeviceBuffer<BlAuxSample> d_aux;               // auxiliary-image mode
    DeviceBuffer<double> d_sample_t;               // image_time, slow light
    DeviceBuffer<double> d_slow_frac;              // slow light: t_frac of every located sample
    DeviceBuffer<BlPolSample> d_pol_samples;       // polarized transfer
    DeviceBuffer<double> d_pol_matrix;             // tolerant tier: 10 doubles per sample
    DeviceBuffer<BlFreqInputs> d_freq_inputs;      // tolerant tier, several frequencies
    DeviceBuffer<double2> d_pol_coeffs;
    DeviceBuffer<unsigned int> d_anchors;          // inter-block interpolation: eight anchor cells per record
    DeviceBuffer<BlCoefInputs> d_coef_inputs;      // polarized runs: coefficient kernel -> polarized coefficient kernel
    DeviceBuffer<unsigned long long> d_redo;       // tolerant tier: records left to the exact coefficient kernel
    void Free() {
      d_redo.Free(); d_aux.Free(); d_sample_t.Free(); d_slow_frac.Free(); d_pol_samples.Free(); d_pol_matrix.Free(); d_freq_inputs.Free(); d_pol_coeffs.Free(); d_coef_inputs.Free(); d_anchors.Free();
      d_records_hot.Free(); d_records_cold.Free(); d_located.Free(); d_located_tag.Free(); d_transfer.Free(); d_ray_kt.Free(); d_ray_factor.Free();
      d_ray_sample_num.Free(); d_ray_flags.Free(); d_ray_out_index.Free(); d_counters.Free();
    }
  };
  ChunkSlot slot[2];
  DeviceBuffer<double> d_freq;
  DeviceBuffer<int> d_pixel_map, d_block_locs, d_tile_order;
  int tile_order_res = 0;
  DeviceBuffer<BlShadeCold> d_shade_cold;
  // host-output staging
  DeviceBuffer<double> d_image, d_camera_pos, d_camera_dir;
  DeviceBuffer<int> d_out_sample_num;
  DeviceBuffer<unsigned char> d_out_flags;

  // geodesic checkpoint of the root level (geodesic_checkpoint.cpp:28-108): what LoadGeodesics() read, by pixel
  struct Checkpoint {
    bool loaded = false;
    int num_steps = 0;
    std::vector<double> camera_pos, camera_dir, factors;   // [n_pix][4], [n_pix][4], [n_pix]
    std::vector<uint8_t> flags;
    std::vector<int32_t> sample_num;
    std::vector<double> pos, dir, len;   // [n_pix][num_steps][4] x 2, [n_pix][num_steps]: reference order (far -> near)
  } checkpoint;

  bl_stats stats{};
};

namespace {

void Warn(bl_ctx *ctx, const std::string &message) { ctx->warnings += "Warning: " + message + "\n"; }

// RadiationIntegrator::Hypergeometric (simulation_coefficients.cpp:740-773): 2F1 for z < 0 through its Pfaff
// transformation, ten terms of the series
double Hypergeometric(double alpha, double beta, double gamma, double z) {
  const double a = alpha, b = gamma - beta, c = gamma;
  const double x = z / (z - 1.0);
  double result = 1.0, a_k = 1.0, b_k = 1.0, c_k = 1.0, xk = 1.0, k_factorial = 1.0;
  for (int k = 1; k <= 10; k++) {
    a_k *= a + k - 1.0;
    b_k *= b + k - 1.0;
    c_k *= c + k - 1.0;
    xk *= x;
    k_factorial *= k;
    result += a_k * b_k * xk / (c_k * k_factorial);
  }
  result *= bl_pow(1.0 - z, -alpha);
  return result;
}

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
    if (p.checkpoint_sample_save || p.checkpoint_sample_load)
      // The reference cannot read these files itself: LoadSampling() (sample_checkpoint.cpp:49-63) restores sample_inds,
      // sample_fracs, sample_nan and sample_fallback but not sample_cut, which only CalculateSimulationSampling() allocates
      // (simulation_sampling.cpp:155) and SampleSimulation() reads for every sample (:691) - the reference binary built
      // from /root/reference ends in a segmentation fault on checkpoint_sample_load = true. A file nobody can read back has
      // no defined result to match, so neither direction is offered; geodesic checkpoints are (LoadGeodesicCheckpoint).
      throw Failure{BL_E_UNSUPPORTED, "Sample checkpoints are not offered: the reference cannot load them (its LoadSampling() "
                                      "leaves sample_cut unallocated); use geodesic checkpoints."};
    Require(p, {BL_P_simulation_format, BL_P_simulation_coord, BL_P_simulation_m_msun, BL_P_simulation_rho_cgs,
                BL_P_simulation_interp},
            kRadMissing);
    if ((p.simulation_format == BL_SIMFMT_ATHENA || p.simulation_format == BL_SIMFMT_ATHENAK) && p.simulation_interp) {
      Require(p, {BL_P_simulation_block_interp}, kRadMissing);
    } else if (Has(p, BL_P_simulation_block_interp)) {
      Warn(ctx, "Ignoring simulation_block_interp selection.");
    }
    if (p.simulation_coord == BL_COORD_FMKS && p.slow_light_on)
      throw Failure{BL_E_UNSUPPORTED, "simulation_coord = fmks with slow light is not built."};
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
      if (!(p.image_light && p.image_polarization))
        throw Failure{BL_E_UNSUPPORTED, "Kappa-distribution electrons (plasma_kappa_frac != 0) are built for polarized runs only: the reference's "
                                        "unpolarized absorptivity reads kappa_aa_high_i, which it only initialises for polarized runs."};
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

// geodesic start / end, locate start, coefficient start, transfer start, end, counters copied to the host
constexpr int kEventsPerChunk = 7;

void EnsureStreams(bl_ctx *ctx) {
  if (ctx->stream == nullptr) Check(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking), "hipStreamCreate");
  if (ctx->stream_geo == nullptr) Check(hipStreamCreateWithFlags(&ctx->stream_geo, hipStreamNonBlocking), "hipStreamCreate");
}

void EnsureChunkResources(bl_ctx *ctx, int n_chunks) {
  const size_t need = static_cast<size_t>(n_chunks) * kEventsPerChunk + 1;
  while (ctx->events.size() < need) {
    hipEvent_t e = nullptr;
    Check(hipEventCreate(&e), "hipEventCreate");
    ctx->events.push_back(e);
  }
  if (ctx->host_counters_chunks < static_cast<size_t>(n_chunks)) {
    if (ctx->host_counters != nullptr) (void)hipHostFree(ctx->host_counters);
    ctx->host_counters = nullptr;
    Check(hipHostMalloc(reinterpret_cast<void **>(&ctx->host_counters),
                        static_cast<size_t>(n_chunks) * (BL_CNT_COUNT + 4) * sizeof(unsigned long long), hipHostMallocDefault),
          "hipHostMalloc");
    ctx->host_counters_chunks = n_chunks;
  }
}

int Fail(bl_ctx *ctx, const Failure &failure) {
  std::string text = "Error: " + failure.message + "\n";
  if (ctx != nullptr)
    ctx->last_error = text;
  else
    g_global_error = text;
  return failure.code;
}

// ---- geodesic checkpoints (geodesic_checkpoint.cpp:28-108, file_io.cpp:65-127): 7 x 4 doubles of camera frame, then Arrays
// - five int32 dimensions n1 ... n5 (fastest first) followed by the data - of camera_pos (n_pix, 4), camera_dir (n_pix, 4),
// image_frequencies, momentum_factors (n_pix), the int geodesic_num_steps, sample_flags (n_pix, bool), sample_num (n_pix, int),
// sample_pos (n_pix, n_steps, 4), sample_dir (n_pix, n_steps, 4), sample_len (n_pix, n_steps); root level only.
template <typename T>
void ReadCheckpointArray(std::ifstream &in, std::vector<T> *data, int dims[5]) {
  in.read(reinterpret_cast<char *>(dims), 5 * sizeof(int));
  size_t count = 1;
  for (int a = 0; a < 5; a++) count *= static_cast<size_t>(std::max(dims[a], 1));
  if (!in || count > (1ull << 36) / sizeof(T)) throw Failure{BL_E_INPUT, "Geodesic checkpoint file is damaged."};
  data->resize(count);
  in.read(reinterpret_cast<char *>(data->data()), static_cast<std::streamsize>(count * sizeof(T)));
  if (!in) throw Failure{BL_E_INPUT, "Geodesic checkpoint file is damaged."};
}

void LoadGeodesicCheckpoint(bl_ctx *ctx) {
  const bl_params &p = ctx->params;
  std::ifstream in(p.checkpoint_geodesic_file.s, std::ios_base::in | std::ios_base::binary);
  if (!in.is_open()) throw Failure{BL_E_INPUT, "Could not open geodesic checkpoint file."};
  bl_camera_frame &f = ctx->frame;
  double *vectors[7] = {f.cam_x, f.u_con, f.u_cov, f.norm_con, f.norm_con_c, f.hor_con_c, f.vert_con_c};
  for (double *v : vectors) in.read(reinterpret_cast<char *>(v), 4 * sizeof(double));
  bl_ctx::Checkpoint &c = ctx->checkpoint;
  const size_t n_pix = static_cast<size_t>(p.camera_resolution) * p.camera_resolution;
  int dims[5];
  std::vector<double> frequencies;
  ReadCheckpointArray(in, &c.camera_pos, dims);
  ReadCheckpointArray(in, &c.camera_dir, dims);
  ReadCheckpointArray(in, &frequencies, dims);
  ReadCheckpointArray(in, &c.factors, dims);
  in.read(reinterpret_cast<char *>(&c.num_steps), sizeof(int));
  ReadCheckpointArray(in, &c.flags, dims);
  ReadCheckpointArray(in, &c.sample_num, dims);
  ReadCheckpointArray(in, &c.pos, dims);
  ReadCheckpointArray(in, &c.dir, dims);
  ReadCheckpointArray(in, &c.len, dims);
  const size_t steps = static_cast<size_t>(std::max(c.num_steps, 0));
  if (c.camera_pos.size() != 4 * n_pix || c.camera_dir.size() != 4 * n_pix || c.factors.size() != n_pix || c.flags.size() != n_pix
      || c.sample_num.size() != n_pix || c.pos.size() != n_pix * steps * 4 || c.dir.size() != n_pix * steps * 4
      || c.len.size() != n_pix * steps || static_cast<int>(frequencies.size()) != p.image_num_frequencies || c.num_steps > p.ray_max_steps)
    throw Failure{BL_E_INPUT, "Geodesic checkpoint does not match this camera (resolution, frequencies or ray_max_steps)."};
  for (size_t m = 0; m < n_pix; m++)
    if (c.sample_num[m] < 0 || c.sample_num[m] > c.num_steps) throw Failure{BL_E_INPUT, "Geodesic checkpoint file is damaged."};
  ctx->frequencies = frequencies;   // LoadGeodesics() replaces what InitializeCamera() would have computed
  c.loaded = true;
}

template <typename T>
void WriteCheckpointHeader(std::ofstream &out, int n1, int n2, int n3) {
  const int dims[5] = {n1, n2, n3, 1, 1};
  out.write(reinterpret_cast<const char *>(dims), sizeof dims);
}

}  // namespace

extern "C" {

int bl_init(const bl_params *p, int device, bl_ctx **out) {
  if (p == nullptr || out == nullptr) return BL_E_ARG;
  *out = nullptr;
  bl_ctx *ctx = new bl_ctx();
  ctx->params = *p;
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
    std::vector<float> cells(n_cells * 8);
    for (int blk = 0; blk < n_b; blk++) {
      const int pi = block_pos[0][blk], pj = block_pos[1][blk], pk = block_pos[2][blk];
      for (int v = 0; v < 8; v++) {
        const float *src = g->prim + (static_cast<size_t>(order[v]) * n_b + blk) * block_cells;
        for (int k = 0; k < nb_cells[2]; k++)
          for (int j = 0; j < nb_cells[1]; j++) {
            const size_t row = (static_cast<size_t>(pk * nb_cells[2] + k) * n_j + (pj * nb_cells[1] + j)) * n_i + static_cast<size_t>(pi) * nb_cells[0];
            const float *line = src + (static_cast<size_t>(k) * nb_cells[1] + j) * nb_cells[0];
            for (int i = 0; i < nb_cells[0]; i++) cells[(row + i) * 8 + v] = line[i];
          }
      }
    }
    DeviceBuffer<float> &d_cells = ctx->cells_target != nullptr ? *ctx->cells_target : ctx->d_cells;
    DeviceBuffer<float> &d_kappa = ctx->kappa_target != nullptr ? *ctx->kappa_target : ctx->d_kappa;
    d_cells.Ensure(cells.size());
    Check(hipMemcpy(d_cells.ptr, cells.data(), cells.size() * sizeof(float), hipMemcpyHostToDevice), "grid upload");
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
      dev.n_bucket[a] = n_bucket;
      off_b[a] = buckets.size();
      buckets.insert(buckets.end(), table.begin(), table.end());
    }
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
  std::vector<float> cells(n_cells * 8);
  for (int v = 0; v < 8; v++) {
    const float *src = g->prim + static_cast<size_t>(order[v]) * n_cells;
    for (size_t c = 0; c < n_cells; c++) cells[c * 8 + v] = src[c];
  }
  DeviceBuffer<float> &d_cells = ctx->cells_target != nullptr ? *ctx->cells_target : ctx->d_cells;
  DeviceBuffer<float> &d_kappa = ctx->kappa_target != nullptr ? *ctx->kappa_target : ctx->d_kappa;
  d_cells.Ensure(cells.size());
  Check(hipMemcpy(d_cells.ptr, cells.data(), cells.size() * sizeof(float), hipMemcpyHostToDevice), "grid upload");
  const bool code_kappa = ctx->params.plasma_model == BL_PLASMA_CODE_KAPPA;
  if (code_kappa) {
    if (g->ind_kappa < 0 || g->ind_kappa >= g->n_var) throw Failure{BL_E_ARG, "Grid variable index out of range."};
    d_kappa.Ensure(n_cells);
    Check(hipMemcpy(d_kappa.ptr, g->prim + static_cast<size_t>(g->ind_kappa) * n_cells, n_cells * sizeof(float),
                    hipMemcpyHostToDevice), "grid upload");
  }
  // coordinate rows of every block, then the block boundaries
  std::vector<double> coords;
  size_t off_f[3], off_v[3], off_e[3];
  for (int a = 0; a < 3; a++) {
    off_f[a] = coords.size();
    coords.insert(coords.end(), block_xf[a], block_xf[a] + static_cast<size_t>(n_b) * (nb[a] + 1));
    off_v[a] = coords.size();
    coords.insert(coords.end(), block_xv[a], block_xv[a] + static_cast<size_t>(n_b) * nb[a]);
    off_e[a] = coords.size();
    coords.insert(coords.end(), edge[a].begin(), edge[a].end());
  }
  ctx->d_coords.Ensure(coords.size());
  Check(hipMemcpy(ctx->d_coords.ptr, coords.data(), coords.size() * sizeof(double), hipMemcpyHostToDevice), "coordinate upload");
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
    dev.edge[a] = ctx->d_coords.ptr + off_e[a];
    dev.n_edge[a] = n_edge[a];
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
  ctx->grid_dev = dev;
  ctx->lds_table_bytes = 0;
  ctx->n_i = nb[0];
  ctx->n_j = nb[1];
  ctx->n_k = nb[2];
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
    ctx->have_grid = false;   // a failed upload leaves no grid behind (the previous one may be half overwritten)
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
  if (ctx == nullptr || (policy != BL_UNDEFINED_REFUSE && policy != BL_UNDEFINED_EDGE)) return BL_E_ARG;
  ctx->undefined_policy = policy;
  return BL_OK;
}

int bl_set_arithmetic(bl_ctx *ctx, int mode) {
  if (ctx == nullptr || (mode != BL_ARITH_EXACT && mode != BL_ARITH_TOLERANT)) return BL_E_ARG;
  ctx->arithmetic = mode;
  return BL_OK;
}

int bl_set_scratch_limit(bl_ctx *ctx, uint64_t bytes) {
  if (ctx == nullptr || bytes < (1ull << 20)) return BL_E_ARG;
  ctx->scratch_limit = bytes;
  return BL_OK;
}

int bl_render(bl_ctx *ctx, const bl_render_desc *d) {
  if (ctx == nullptr || d == nullptr) return BL_E_ARG;
  try {
    const bl_params &p = ctx->params;
    const bool simulation = p.model_type == BL_MODEL_SIMULATION;
    if (ctx->device == BL_DEVICE_NONE) throw Failure{BL_E_DEVICE, "Host-only context: no HIP device selected (the hot path has no CPU fallback)."};
    if (simulation && !ctx->have_grid) throw Failure{BL_E_STATE, "bl_render called before bl_set_grid."};
    if (d->n_rays <= 0 || (d->image == nullptr && ctx->image_num_quantities > 0))
      throw Failure{BL_E_ARG, "bl_render needs n_rays > 0 and an image buffer."};
    if (ctx->render_num_images > 0 && d->render == nullptr) throw Failure{BL_E_ARG, "bl_render needs a render buffer when render_num_images > 0."};
    if (d->n_rays > 0x7fffffffll) throw Failure{BL_E_ARG, "Too many rays in one bl_render call."};
    if (d->level < 0 || d->level > p.adaptive_max_level) throw Failure{BL_E_ARG, "Adaptive level out of range."};
    if (d->level > 0 && (d->block_locs == nullptr || d->n_blocks <= 0)) throw Failure{BL_E_ARG, "Refined level needs block_locs."};
    Check(hipSetDevice(ctx->device), "hipSetDevice");
    EnsureStreams(ctx);
    hipStream_t stream = ctx->stream, stream_geo = ctx->stream_geo;
    if (!ctx->overlap_chunks) stream_geo = stream;   // default: one stream, chunks back to back
    const int n_nu = p.image_num_frequencies;
    const int n_q = ctx->image_num_quantities;
    const int max_steps = p.ray_max_steps;
    const long long n_rays = d->n_rays;
    const bool aux = ctx->aux_images.any != 0;
    const bool slow = simulation && p.slow_light_on;
    if (slow) {
      if (static_cast<int>(ctx->slow_slices.size()) != p.slow_chunk_size) throw Failure{BL_E_STATE, "Slow light: time slices not set."};
      for (const bl_ctx::SlowSlice &slice : ctx->slow_slices)
        if (!slice.set) throw Failure{BL_E_STATE, "Slow light: time slices not set."};
    }
    // geodesic checkpoints (root level only, like the reference's): load replaces the geodesic kernel by the file's
    // samples, save writes what the geodesic kernel produced in the reference's layout
    const bool geo_load = p.checkpoint_geodesic_load && d->level == 0;
    const bool geo_save = p.checkpoint_geodesic_save && d->level == 0;
    if (geo_load && !ctx->checkpoint.loaded) LoadGeodesicCheckpoint(ctx);
    const bool need_time = (aux && ctx->aux_images.image_time) || slow || geo_load || geo_save;

    // level pixel count check
    long long level_pixels = static_cast<long long>(p.camera_resolution) * p.camera_resolution;
    if (d->level > 0) level_pixels = static_cast<long long>(d->n_blocks) * p.adaptive_block_size * p.adaptive_block_size;
    if (d->pixel_map == nullptr && n_rays > level_pixels) throw Failure{BL_E_ARG, "n_rays exceeds the pixels of this level."};

    if (geo_save && (d->pixel_map != nullptr || n_rays != level_pixels))
      throw Failure{BL_E_ARG, "checkpoint_geodesic_save needs the whole root camera in one bl_render call."};
    const bool block_interp = simulation && ctx->grid_dev.block_interp != 0;
    // Tolerant tier: plain unpolarized images of a spherical Kerr-Schild simulation with thermal electrons in a curved
    // spacetime have the fast coefficient kernel; every other configuration is rendered in exact arithmetic whatever
    // bl_set_arithmetic() asked for (bl_stats.arithmetic says which tier ran)
    const bool fast = ctx->arithmetic == BL_ARITH_TOLERANT && simulation && !aux && !ctx->polarized && !slow && !block_interp
        && p.plasma_power_frac == 0.0 && p.plasma_kappa_frac == 0.0 && p.plasma_model != BL_PLASMA_CODE_KAPPA
        && (p.simulation_coord == BL_COORD_SKS || p.simulation_coord == BL_COORD_FMKS) && !p.ray_flat && ctx->plasma_thermal_frac != 0.0
        && n_nu <= 1024;   // (its LDS table holds five numbers per frequency)
    // ... and the per-frequency coefficient kernel of polarized runs (frame, transport and coupling stay exact)
    const bool tolerant_polarized = ctx->arithmetic == BL_ARITH_TOLERANT && ctx->polarized;
    // ... and transport matrices (bl_transport_matrix_kernel) instead of the ray-sequential tensor transport, in curved spacetimes
    const bool matrix_transport = tolerant_polarized && !p.ray_flat;
    // Several frequencies in the fast path: per-sample factors (BlFreqInputs) instead of per-frequency transfer records,
    // evaluated by bl_transfer_freq_kernel with one lane per ray and frequency
    const bool freq_split = fast && n_nu >= 4;
    // ... and in the exact coefficient kernel (plain images): the frequency loop as lanes of bl_coefficients_freq_kernel
    const bool coef_split = !fast && simulation && !aux && !ctx->polarized && n_nu >= 4;
    // (polarized runs list the samples without coefficients there - cut samples, cut cells - which are many more)
    const size_t redo_capacity = ctx->polarized ? (1u << 24) : (1u << 20);
    // chunk size from the scratch budget: per ray max_steps * (2 x 32 B record + 40 B located sample
    // (simulation mode) + 16 B * n_nu transfer)
    const uint64_t per_ray = static_cast<uint64_t>(max_steps)
        * (sizeof(BlSampleHot) + sizeof(BlSampleCold) + (simulation ? sizeof(BlLocated) + sizeof(unsigned long long) : 0) + (freq_split ? 0 : sizeof(double2) * n_nu)
           + (aux ? sizeof(BlAuxSample) + sizeof(double) : 0) + (slow ? 2 * sizeof(double) : 0)
           + (ctx->polarized ? sizeof(BlPolSample) + sizeof(BlCoefInputs) + 3 * sizeof(double2) * n_nu : 0)
           + (coef_split ? sizeof(BlCoefInputs) : 0)
           + (matrix_transport ? BL_POL_MATRIX_DOUBLES * sizeof(double) : 0) + (freq_split ? sizeof(BlFreqInputs) : 0)
           + (block_interp ? 8 * sizeof(unsigned int) : 0)) + 64;
    // One chunk if the whole call fits the budget. With bl_set_overlap(): two scratch sets of half the budget
    // each, so that the geodesic kernel of chunk c + 1 runs while chunk c is being shaded.
    // The budget is also capped by what the device can actually give: 90 % of (free memory + the scratch
    // this context already holds from earlier renders).
    uint64_t budget = ctx->scratch_limit;
    {
      size_t free_bytes = 0, total_bytes = 0;
      if (hipMemGetInfo(&free_bytes, &total_bytes) == hipSuccess) {
        uint64_t held = 0;
        for (const bl_ctx::ChunkSlot &sl : ctx->slot)
          held += sl.d_records_hot.count * (sizeof(BlSampleHot) + sizeof(BlSampleCold))
              + sl.d_located.count * (sizeof(BlLocated) + sizeof(unsigned long long))
              + sl.d_transfer.count * sizeof(double2) + sl.d_aux.count * sizeof(BlAuxSample)
              + (sl.d_sample_t.count + sl.d_slow_frac.count) * sizeof(double)
              + sl.d_pol_samples.count * sizeof(BlPolSample) + (sl.d_pol_coeffs.count) * sizeof(double2) + sl.d_pol_matrix.count * sizeof(double)
              + sl.d_coef_inputs.count * sizeof(BlCoefInputs) + sl.d_anchors.count * sizeof(unsigned int)
              + sl.d_freq_inputs.count * sizeof(BlFreqInputs);
        const uint64_t available = static_cast<uint64_t>(0.9 * static_cast<double>(free_bytes + held));
        if (available < budget) budget = available;
      }
    }
    long long chunk = static_cast<long long>(budget / per_ray);
    if (chunk < n_rays && ctx->overlap_chunks) chunk = static_cast<long long>(budget / (2 * per_ray));
    chunk = std::max<long long>(chunk, 64);
    chunk = std::min<long long>(chunk, n_rays);
    if (chunk < n_rays) chunk = (chunk / 64) * 64;   // keep 8x8 tiles whole
    const int n_chunks = static_cast<int>((n_rays + chunk - 1) / chunk);
    const int n_slots = (n_chunks > 1 && ctx->overlap_chunks) ? 2 : 1;

    const int geo_blocks_per_cu = bl_geodesic_occupancy(p.ray_integrator, need_time ? 1 : 0, ctx->st.bh_a == 0.0 ? 1 : 0);
    const int geo_grid = ctx->num_cus * geo_blocks_per_cu;   // persistent waves of the geodesic kernel
    const size_t record_capacity = static_cast<size_t>(chunk) * max_steps + static_cast<size_t>(geo_grid) * BL_RECORD_BLOCK;
    for (int k = 0; k < n_slots; k++) {
      bl_ctx::ChunkSlot &sl = ctx->slot[k];
      sl.d_records_hot.Ensure(record_capacity);
      sl.d_records_cold.Ensure(record_capacity);
      if (simulation) {
        sl.d_located.Ensure(record_capacity);
        sl.d_located_tag.Ensure(record_capacity);
      }
      if (freq_split) sl.d_freq_inputs.Ensure(static_cast<size_t>(chunk) * max_steps);   // instead of the transfer records
      else sl.d_transfer.Ensure(static_cast<size_t>(chunk) * max_steps * n_nu);
      sl.d_ray_kt.Ensure(chunk);
      sl.d_ray_factor.Ensure(chunk);
      sl.d_ray_sample_num.Ensure(chunk);
      sl.d_ray_flags.Ensure(chunk);
      sl.d_ray_out_index.Ensure(chunk);
      sl.d_counters.Ensure(BL_CNT_COUNT + 4);
      if (aux) sl.d_aux.Ensure(static_cast<size_t>(chunk) * max_steps);
      if (need_time) sl.d_sample_t.Ensure(record_capacity);
      if (slow) sl.d_slow_frac.Ensure(record_capacity);
      if (ctx->polarized) {
        sl.d_pol_samples.Ensure(static_cast<size_t>(chunk) * max_steps);
        if (matrix_transport) sl.d_pol_matrix.Ensure(static_cast<size_t>(chunk) * max_steps * BL_POL_MATRIX_DOUBLES);
        sl.d_pol_coeffs.Ensure(static_cast<size_t>(chunk) * max_steps * n_nu * 3);
        sl.d_coef_inputs.Ensure(record_capacity);
      }
      if (coef_split) sl.d_coef_inputs.Ensure(record_capacity);
      if (block_interp) sl.d_anchors.Ensure(record_capacity * 8);
      if (fast || ctx->polarized) sl.d_redo.Ensure(redo_capacity);   // polarized runs: the samples whose frame bl_polarized_frame_kernel builds
    }
    EnsureChunkResources(ctx, n_chunks);
    ctx->d_freq.Ensure(n_nu);
    Check(hipMemcpyAsync(ctx->d_freq.ptr, ctx->frequencies.data(), n_nu * sizeof(double), hipMemcpyHostToDevice, stream), "freq upload");

    const int *d_pixel_map = nullptr, *d_block_locs = nullptr;
    if (d->pixel_map != nullptr) {
      ctx->d_pixel_map.Ensure(n_rays);
      Check(hipMemcpyAsync(ctx->d_pixel_map.ptr, d->pixel_map, n_rays * sizeof(int), hipMemcpyHostToDevice, stream), "pixel_map upload");
      d_pixel_map = ctx->d_pixel_map.ptr;
    }
    if (d->level > 0) {
      ctx->d_block_locs.Ensure(static_cast<size_t>(d->n_blocks) * 2);
      Check(hipMemcpyAsync(ctx->d_block_locs.ptr, d->block_locs, static_cast<size_t>(d->n_blocks) * 2 * sizeof(int), hipMemcpyHostToDevice, stream), "block_locs upload");
      d_block_locs = ctx->d_block_locs.ptr;
    }

    // output buffers: caller's HBM, or staging
    double *image = d->image, *cam_pos = d->camera_pos, *cam_dir = d->camera_dir;
    int *out_num = d->sample_num;
    unsigned char *out_flags = d->sample_flags;
    if (!d->outputs_on_device) {
      ctx->d_image.Ensure(static_cast<size_t>(n_q) * n_rays);
      image = ctx->d_image.ptr;
      if (d->sample_num != nullptr) { ctx->d_out_sample_num.Ensure(n_rays); out_num = ctx->d_out_sample_num.ptr; }
      if (d->sample_flags != nullptr) { ctx->d_out_flags.Ensure(n_rays); out_flags = ctx->d_out_flags.ptr; }
      if (d->camera_pos != nullptr) { ctx->d_camera_pos.Ensure(static_cast<size_t>(n_rays) * 4); cam_pos = ctx->d_camera_pos.ptr; }
      if (d->camera_dir != nullptr) { ctx->d_camera_dir.Ensure(static_cast<size_t>(n_rays) * 4); cam_dir = ctx->d_camera_dir.ptr; }
    }
    if (ctx->polarized || geo_save) {   // the camera tetrad projection (and the checkpoint) need every ray's initial position and momentum
      if (cam_pos == nullptr) { ctx->d_camera_pos.Ensure(static_cast<size_t>(n_rays) * 4); cam_pos = ctx->d_camera_pos.ptr; }
      if (cam_dir == nullptr) { ctx->d_camera_dir.Ensure(static_cast<size_t>(n_rays) * 4); cam_dir = ctx->d_camera_dir.ptr; }
    }
    if (geo_load && (cam_pos != nullptr || cam_dir != nullptr)) {   // camera_pos / camera_dir come from the file as well
      std::vector<double> rows(static_cast<size_t>(n_rays) * 4);
      for (int which = 0; which < 2; which++) {
        double *target = which == 0 ? cam_pos : cam_dir;
        if (target == nullptr) continue;
        const std::vector<double> &source = which == 0 ? ctx->checkpoint.camera_pos : ctx->checkpoint.camera_dir;
        for (long long ray = 0; ray < n_rays; ray++) {
          const size_t m = d->pixel_map != nullptr ? static_cast<size_t>(d->pixel_map[ray]) : static_cast<size_t>(ray);
          for (int mu = 0; mu < 4; mu++) rows[4 * ray + mu] = source[4 * m + mu];
        }
        Check(hipMemcpy(target, rows.data(), rows.size() * sizeof(double), hipMemcpyHostToDevice), "checkpoint upload");
      }
    }
    double *render_out = nullptr;
    bool fill_present = false;
    if (ctx->render_num_images > 0) {
      BlRenderDevice rp{};
      rp.n_images = ctx->render_num_images;
      for (int n_i = 0; n_i < rp.n_images; n_i++) {
        rp.n_features[n_i] = p.render_num_features[n_i];
        for (int n_f = 0; n_f < rp.n_features[n_i]; n_f++) {
          rp.quantity[n_i][n_f] = p.render_quantity[n_i][n_f];
          rp.type[n_i][n_f] = p.render_type[n_i][n_f];
          rp.min_val[n_i][n_f] = p.render_min[n_i][n_f];
          rp.max_val[n_i][n_f] = p.render_max[n_i][n_f];
          rp.thresh[n_i][n_f] = p.render_thresh[n_i][n_f];
          rp.tau_scale[n_i][n_f] = p.render_tau_scale[n_i][n_f];
          rp.opacity[n_i][n_f] = p.render_opacity[n_i][n_f];
          rp.xyz[n_i][n_f][0] = p.render_x[n_i][n_f];
          rp.xyz[n_i][n_f][1] = p.render_y[n_i][n_f];
          rp.xyz[n_i][n_f][2] = p.render_z[n_i][n_f];
          if (rp.type[n_i][n_f] == BL_RENDER_FILL) fill_present = true;
        }
      }
      rp.fill_present = fill_present ? 1 : 0;
      ctx->d_render_params.Ensure(1);
      Check(hipMemcpyAsync(ctx->d_render_params.ptr, &rp, sizeof(BlRenderDevice), hipMemcpyHostToDevice, stream), "render parameter upload");
      Check(hipStreamSynchronize(stream), "render parameter upload");   // rp is a local
      render_out = d->render;
      if (!d->outputs_on_device) {
        ctx->d_render.Ensure(static_cast<size_t>(ctx->render_num_images) * 3 * n_rays);
        render_out = ctx->d_render.ptr;
      }
    }

    // ---- kernel arguments common to all chunks
    BlTraceArgs ta{};
    ta.st = ctx->st;
    BlCameraDevice &cam = ta.cam;
    for (int mu = 0; mu < 4; mu++) {
      cam.cam_x[mu] = ctx->frame.cam_x[mu];
      cam.u_con[mu] = ctx->frame.u_con[mu];
      cam.u_cov[mu] = ctx->frame.u_cov[mu];
      cam.norm_con[mu] = ctx->frame.norm_con[mu];
      cam.norm_con_c[mu] = ctx->frame.norm_con_c[mu];
      cam.hor_con_c[mu] = ctx->frame.hor_con_c[mu];
      cam.vert_con_c[mu] = ctx->frame.vert_con_c[mu];
    }
    cam.camera_width = p.camera_width;
    cam.camera_r = p.camera_r;
    cam.camera_type = p.camera_type;
    cam.image_normalization = p.image_normalization;
    cam.camera_resolution = p.camera_resolution;
    cam.level = d->level;
    cam.block_size = p.adaptive_max_level > 0 ? p.adaptive_block_size : 1;
    cam.effective_resolution = p.camera_resolution;
    for (int l = 1; l <= d->level; l++) cam.effective_resolution *= 2;
    ta.r_terminate = ctx->frame.r_terminate;
    ta.r_horizon = ctx->frame.r_horizon;
    ta.camera_r = p.camera_r;
    ta.ray_step = p.ray_step;
    ta.ray_tol_abs = p.ray_integrator == BL_INTEGRATOR_DP ? p.ray_tol_abs : 0.0;
    ta.ray_tol_rel = p.ray_integrator == BL_INTEGRATOR_DP ? p.ray_tol_rel : 0.0;
    ta.ray_max_steps = max_steps;
    ta.ray_max_retries = p.ray_integrator == BL_INTEGRATOR_DP ? p.ray_max_retries : 0;
    ta.n_rays_total = n_rays;
    ta.swizzle_tiles = (d->level == 0 && d->pixel_map == nullptr && p.camera_resolution % 8 == 0 && n_rays == level_pixels)
        ? p.camera_resolution : 0;
    // Order in which the 8x8 pixel tiles of a full frame are traced: centre of the image first. Rays near
    // the centre (photon ring, disc) are the long ones, the periphery is short; a chunk that ends on short
    // rays drains its persistent waves quickly (measured: geodesic kernel 33.9 -> 29.4 ms per frame at four
    // chunks), and waves of similar ray lengths also diverge less in the transfer kernel.
    ta.tile_order = nullptr;
    if (ta.swizzle_tiles > 0) {
      if (ctx->tile_order_res != p.camera_resolution) {
        const int tiles_per_row = p.camera_resolution / 8;
        const int n_tiles = tiles_per_row * tiles_per_row;
        std::vector<int> order(n_tiles);
        for (int t = 0; t < n_tiles; t++) order[t] = t;
        const double centre = 0.5 * (tiles_per_row - 1);
        auto dist2 = [&](int t) {
          double dy = t / tiles_per_row - centre, dx = t % tiles_per_row - centre;
          return dx * dx + dy * dy;
        };
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return dist2(a) < dist2(b); });
        ctx->d_tile_order.Ensure(n_tiles);
        Check(hipMemcpy(ctx->d_tile_order.ptr, order.data(), n_tiles * sizeof(int), hipMemcpyHostToDevice), "tile order upload");
        ctx->tile_order_res = p.camera_resolution;
      }
      ta.tile_order = ctx->d_tile_order.ptr;
    }
    ta.pixel_map = d_pixel_map;
    ta.block_locs = d_block_locs;
    ta.record_capacity = static_cast<long long>(record_capacity);
    ta.camera_pos = cam_pos;
    ta.camera_dir = cam_dir;

    BlShadeArgs sa{};
    sa.st = ctx->st;
    BlShadeCold cold{};
    cold.omit_near = p.cut_omit_near;
    cold.omit_far = p.cut_omit_far;
    cold.plane = p.cut_plane;
    cold.omit_in = p.cut_omit_in;
    cold.omit_out = p.cut_omit_out;
    cold.midplane_theta = p.cut_midplane_theta;
    cold.midplane_z = p.cut_midplane_z;
    cold.plane_origin[0] = p.cut_plane_origin_x;
    cold.plane_origin[1] = p.cut_plane_origin_y;
    cold.plane_origin[2] = p.cut_plane_origin_z;
    cold.plane_normal[0] = p.cut_plane_normal_x;
    cold.plane_normal[1] = p.cut_plane_normal_y;
    cold.plane_normal[2] = p.cut_plane_normal_z;
    for (int mu = 0; mu < 4; mu++) cold.cam_x[mu] = ctx->frame.cam_x[mu];
    sa.cuts.camera_r = p.camera_r;
    sa.cuts.any_optional = (p.cut_omit_near || p.cut_omit_far || p.cut_omit_in >= 0.0 || p.cut_omit_out >= 0.0
                            || p.cut_midplane_theta != 0.0 || p.cut_midplane_z != 0.0 || p.cut_plane) ? 1 : 0;
    if (simulation) {
      BlPlasmaDevice &pl = sa.plasma;
      pl.d_unit = p.simulation_rho_cgs;                       // simulation_coefficients.cpp:237-239
      pl.e_unit = pl.d_unit * kC * kC;
      pl.b_unit = blm_sqrt(4.0 * kPi * pl.e_unit);
      pl.plasma_mu = p.plasma_mu;
      pl.plasma_ne_ni = p.plasma_ne_ni;
      pl.plasma_rat_low = p.plasma_rat_low;
      pl.plasma_rat_high = p.plasma_rat_high;
      pl.plasma_thermal_frac = ctx->plasma_thermal_frac;
      pl.power_frac = p.plasma_power_frac;
      pl.plasma_p = 0.0;
      pl.power_jj = pl.power_aa = 0.0;
      if (p.plasma_power_frac != 0.0) {
        // simulation_coefficients.cpp:54-66 (unpolarized part); pow is the pinned one, tgamma the host libm's
        const double plasma_p = p.plasma_p;
        const double var_a = bl_pow(3.0, plasma_p / 2.0) * (plasma_p - 1.0);
        const double var_b = 2.0 * (plasma_p + 1.0);
        const double var_c = bl_pow(p.plasma_gamma_min, 1.0 - plasma_p) - bl_pow(p.plasma_gamma_max, 1.0 - plasma_p);
        const double var_d = std::tgamma((3.0 * plasma_p - 1.0) / 12.0);
        const double var_e = std::tgamma((3.0 * plasma_p + 19.0) / 12.0);
        const double var_f = bl_pow(3.0, (plasma_p + 1.0) / 2.0) * (plasma_p - 1.0) / 4.0;
        const double var_g = std::tgamma((3.0 * plasma_p + 2.0) / 12.0);
        const double var_h = std::tgamma((3.0 * plasma_p + 22.0) / 12.0);
        pl.plasma_p = plasma_p;
        pl.power_jj = var_a / var_b / var_c * var_d * var_e;
        pl.power_aa = var_f / var_c * var_g * var_h;
        if (ctx->polarized) {   // simulation_coefficients.cpp:67-80
          const double var_i = 2.0 * (plasma_p + 2.0) / (plasma_p + 1.0);
          const double var_j = bl_pow(p.plasma_gamma_min, -(plasma_p + 1.0));
          const double var_k = bl_log(p.plasma_gamma_min);
          ctx->power_pol[0] = -(plasma_p + 1.0) / (plasma_p + 7.0 / 3.0);
          ctx->power_pol[1] = 0.684 * bl_pow(plasma_p, 0.49);
          ctx->power_pol[2] = -bl_pow(0.034 * plasma_p - 0.0344, 0.086);
          ctx->power_pol[3] = bl_pow(0.71 * plasma_p + 0.0352, 0.394);
          ctx->power_pol[4] = (plasma_p - 1.0) / var_c;
          ctx->power_pol[5] = -bl_pow(p.plasma_gamma_min, 2.0 - plasma_p) / (plasma_p / 2.0 - 1.0);
          ctx->power_pol[6] = var_i * var_j * var_k;
        }
      }
      cold.kappa = BlKappaDevice{};
      if (p.plasma_kappa_frac != 0.0) {
        // simulation_coefficients.cpp:82-193 for a polarized run; pow / exp / log and K_nu are the pinned ones,
        // tgamma the host libm's
        BlKappaDevice &kk = cold.kappa;
        const double plasma_kappa = p.plasma_kappa, plasma_w = p.plasma_w;
        kk.frac = p.plasma_kappa_frac;
        kk.kappa = plasma_kappa;
        kk.w = plasma_w;
        const double var_a = 4.0 * kPi * std::tgamma(plasma_kappa - 4.0 / 3.0);
        const double var_b = bl_pow(3.0, 7.0 / 3.0) * std::tgamma(plasma_kappa - 2.0);
        const double var_c = bl_pow(3.0, (plasma_kappa - 1.0) / 2.0);
        const double var_d = (plasma_kappa - 2.0) * (plasma_kappa - 1.0) / 4.0;
        const double var_e = std::tgamma(plasma_kappa / 4.0 - 1.0 / 3.0);
        const double var_f = std::tgamma(plasma_kappa / 4.0 + 4.0 / 3.0);
        const double var_g = bl_pow(3.0, 1.0 / 6.0) * 10.0 / 41.0;
        const double var_h = plasma_w * plasma_kappa;
        const double var_i = 2.0 * kPi * bl_pow(var_h, plasma_kappa - 10.0 / 3.0);
        const double var_j = (plasma_kappa - 2.0) * (plasma_kappa - 1.0) * plasma_kappa;
        const double var_k = 3.0 * plasma_kappa - 1.0;
        const double var_l = std::tgamma(5.0 / 3.0);
        const double var_m = Hypergeometric(plasma_kappa - 1.0 / 3.0, plasma_kappa + 1.0, plasma_kappa + 2.0 / 3.0, -var_h);
        const double var_n = bl_pow(kPi, 1.5) / 3.0;
        const double var_o = var_j / (var_h * var_h * var_h);
        const double var_p = 2.0 * std::tgamma(2.0 + plasma_kappa / 2.0) / (2.0 + plasma_kappa) - 1.0;
        kk.jj_low = var_a / var_b;
        kk.jj_high = var_c * var_d * var_e * var_f;
        kk.jj_x_i = 3.0 * bl_pow(plasma_kappa, -1.5);
        kk.aa_low = var_g * var_i * var_j / var_k * var_l * var_m;
        kk.aa_high = var_n * var_o * var_p;
        kk.aa_x_i = bl_pow(-1.75 + 1.6 * plasma_kappa, -0.86);
        const double var_q = 14.3 * bl_pow(plasma_w, -0.928);
        const double var_r = 169.0 * bl_pow(plasma_kappa, -8.0) + 0.0052 * plasma_kappa - 0.0526 + 47.0 / (200.0 * plasma_kappa);
        kk.jj_low_q = 0.5;
        kk.jj_low_v = 0.5625 * bl_pow(plasma_kappa, -0.528) / plasma_w;
        kk.jj_high_q = 0.64 + 0.02 * plasma_kappa;
        kk.jj_high_v = 0.765625 * bl_pow(plasma_kappa, -0.44) / plasma_w;
        kk.jj_x_q = 3.7 * bl_pow(plasma_kappa, -1.6);
        kk.jj_x_v = kk.jj_x_i;
        kk.aa_low_q = 25.0 / 48.0;
        kk.aa_low_v = 77.0 / (100.0 * plasma_w) * bl_pow(plasma_kappa, -0.7);
        kk.aa_high_i = bl_pow(3.0 / plasma_kappa, 4.75) + 0.6;
        kk.aa_high_q = 441.0 * bl_pow(plasma_kappa, -5.76) + 0.55;
        kk.aa_high_v = var_q * var_r;
        kk.aa_x_q = 1.4 * bl_pow(plasma_kappa, -1.15);
        kk.aa_x_v = 1.22 * bl_pow(plasma_kappa, -1.136) + 0.007;
        kk.rho_v = bl_cyl_bessel_k(0, 1.0 / plasma_w) / bl_cyl_bessel_k(2, 1.0 / plasma_w);
        // rotativity fits at kappa = 3.5, 4, 4.5, 5 (:128-192); kappa is bracketed by two of them
        const double sqrt_w = blm_sqrt(plasma_w), exp_w = bl_exp(-5.0 * plasma_w);
        const double fit_q[4][5] = {
            {17.0 * plasma_w + sqrt_w * (-3.0 + 7.0 * exp_w), -1.0 / 30.0, 0.1, -1.5, 0.471},
            {46.0 / 3.0 * plasma_w + sqrt_w * (-5.0 / 3.0 + 17.0 / 3.0 * exp_w), -1.0 / 18.0, 1.0 / 6.0, -1.75, 0.5},
            {14.0 * plasma_w + sqrt_w * (-1.625 + 4.5 * exp_w), -1.0 / 12.0, 0.25, -2.0, 0.525},
            {12.5 * plasma_w + sqrt_w * (-1.0 + 5.0 * exp_w), -0.125, 0.375, -2.25, 0.541}};
        const double fit_v[4][2] = {
            {(plasma_w * plasma_w + 2.0 * plasma_w + 1.0) / (3.125 * plasma_w * plasma_w + 4.0 * plasma_w + 1.0), 0.447},
            {(plasma_w * plasma_w + 54.0 * plasma_w + 50.0) / (30.0 / 11.0 * plasma_w * plasma_w + 134.0 * plasma_w + 50.0), 0.391},
            {(plasma_w * plasma_w + 43.0 * plasma_w + 38.0) / (7.0 / 3.0 * plasma_w * plasma_w + 92.5 * plasma_w + 38.0), 0.348},
            {(plasma_w + 13.0 / 14.0) / (2.0 * plasma_w + 13.0 / 14.0), 0.313}};
        const int lo = plasma_kappa < 4.0 ? 0 : (plasma_kappa < 4.5 ? 1 : 2);
        const double k_lo = 3.5 + 0.5 * lo, k_hi = 4.0 + 0.5 * lo;
        kk.rho_frac = (plasma_kappa - k_lo) / (k_hi - k_lo);
        for (int c = 0; c < 5; c++) {
          kk.rho_q_low[c] = fit_q[lo][c];
          kk.rho_q_high[c] = fit_q[lo + 1][c];
        }
        for (int c = 0; c < 2; c++) {
          kk.rho_v_low[c] = fit_v[lo][c];
          kk.rho_v_high[c] = fit_v[lo + 1][c];
        }
      }
      cold.plasma_gamma = ctx->grid_meta.plasma_gamma;
      cold.plasma_gamma_i = ctx->grid_meta.plasma_gamma_i;
      cold.plasma_gamma_e = ctx->grid_meta.plasma_gamma_e;
      pl.plasma_use_p = p.plasma_use_p;
      pl.simulation_interp = p.simulation_interp;
      // fmks: the reader has put vectors on the spherical Kerr-Schild basis; everything but the cell search treats the
      // grid as sks (radiation_geometry.cpp:39, :94, :460, :541)
      pl.simulation_coord = p.simulation_coord == BL_COORD_FMKS ? BL_COORD_SKS : p.simulation_coord;
      pl.fallback_nan = p.fallback_nan;
      cold.fallback_rho = p.fallback_nan ? 0.0f : p.fallback_rho;
      cold.fallback_pgas = p.fallback_nan ? 0.0f : p.fallback_pgas;
      cold.fallback_kappa = p.fallback_nan ? 0.0f : p.fallback_kappa;
      pl.code_kappa = p.plasma_model == BL_PLASMA_CODE_KAPPA ? 1 : 0;
      // cell cuts (simulation_coefficients.cpp:361-375): "cut >= 0 and value < cut". A disabled threshold goes to the
      // device as -inf (lower) / +inf (upper), against which no value - NaN included - compares true: same
      // decisions, one compare per threshold
      const double kInf = std::numeric_limits<double>::infinity();
      auto lower = [&](double cut) { return cut >= 0.0 ? cut : -kInf; };
      auto upper = [&](double cut) { return cut >= 0.0 ? cut : kInf; };
      cold.cut_rho_min = lower(p.cut_rho_min); cold.cut_rho_max = upper(p.cut_rho_max);
      cold.cut_n_e_min = lower(p.cut_n_e_min); cold.cut_n_e_max = upper(p.cut_n_e_max);
      cold.cut_p_gas_min = lower(p.cut_p_gas_min); cold.cut_p_gas_max = upper(p.cut_p_gas_max);
      cold.cut_theta_e_min = lower(p.cut_theta_e_min); cold.cut_theta_e_max = upper(p.cut_theta_e_max);
      cold.cut_b_min = lower(p.cut_b_min); cold.cut_b_max = upper(p.cut_b_max);
      cold.cut_sigma_min = lower(p.cut_sigma_min); cold.cut_sigma_max = upper(p.cut_sigma_max);
      cold.cut_beta_inverse_min = lower(p.cut_beta_inverse_min); cold.cut_beta_inverse_max = upper(p.cut_beta_inverse_max);
      {
        const double cuts[14] = {p.cut_rho_min, p.cut_rho_max, p.cut_n_e_min, p.cut_n_e_max, p.cut_p_gas_min, p.cut_p_gas_max,
                                 p.cut_theta_e_min, p.cut_theta_e_max, p.cut_b_min, p.cut_b_max, p.cut_sigma_min, p.cut_sigma_max,
                                 p.cut_beta_inverse_min, p.cut_beta_inverse_max};
        pl.cut_mask = 0;
        for (int c = 0; c < 14; c++) {
          const bool active = cuts[c] >= 0.0;
          if (active) pl.cut_mask |= 1 << c;
          cold.fast_cut[c] = active ? cuts[c] : 0.0;
          cold.fast_cut_lo[c] = active ? cuts[c] * (1.0 - ctx->guard_band) : 0.0;
          cold.fast_cut_hi[c] = active ? cuts[c] * (1.0 + ctx->guard_band) : 0.0;
        }
        sa.fast_n_e_factor = 1.0 / (p.plasma_mu * kMp * (1.0 + 1.0 / p.plasma_ne_ni));
        sa.fast_gamma[0] = 1.0 / (ctx->grid_meta.plasma_gamma - 1.0);
        sa.fast_gamma[1] = 1.0 / (ctx->grid_meta.plasma_gamma_i - 1.0);
        sa.fast_gamma[2] = 1.0 / (ctx->grid_meta.plasma_gamma_e - 1.0);
      }
      pl.any_cell_cut = (p.cut_rho_min >= 0.0 || p.cut_rho_max >= 0.0 || p.cut_n_e_min >= 0.0 || p.cut_n_e_max >= 0.0
                         || p.cut_p_gas_min >= 0.0 || p.cut_p_gas_max >= 0.0 || p.cut_theta_e_min >= 0.0
                         || p.cut_theta_e_max >= 0.0 || p.cut_b_min >= 0.0 || p.cut_b_max >= 0.0 || p.cut_sigma_min >= 0.0
                         || p.cut_sigma_max >= 0.0 || p.cut_beta_inverse_min >= 0.0 || p.cut_beta_inverse_max >= 0.0) ? 1 : 0;
      sa.grid = ctx->grid_dev;
      sa.lds_table_bytes = ctx->lds_table_bytes;
      sa.undefined_edge = ctx->undefined_policy == BL_UNDEFINED_EDGE ? 1 : 0;
      sa.tolerant = (fast || tolerant_polarized) ? 1 : 0;
      sa.samples_renormalised = geo_load ? 1 : 0;
    } else {
      BlFormulaDevice &fm = sa.formula;
      fm.r0 = p.formula_r0; fm.h = p.formula_h; fm.l0 = p.formula_l0; fm.q = p.formula_q; fm.nup = p.formula_nup;
      fm.cn0 = p.formula_cn0; fm.alpha = p.formula_alpha; fm.a = p.formula_a; fm.beta = p.formula_beta;
    }
    sa.samples_renormalised = geo_load ? 1 : 0;
    ctx->d_shade_cold.Ensure(1);
    Check(hipMemcpyAsync(ctx->d_shade_cold.ptr, &cold, sizeof(BlShadeCold), hipMemcpyHostToDevice, stream), "shade parameter upload");
    sa.cold = ctx->d_shade_cold.ptr;
    sa.frequencies = ctx->d_freq.ptr;
    sa.n_nu = n_nu;
    sa.ray_max_steps = max_steps;
    sa.x_unit = kGGMsun * ctx->frame.mass_msun / (kC * kC);   // unpolarized.cpp:42
    sa.aux_need_coefficients = (p.image_light || p.image_emission || p.image_tau || ctx->aux_images.image_emission_ave
                                || ctx->aux_images.image_tau_int) ? 1 : 0;   // simulation_coefficients.cpp:389
    sa.aux_need_length = (ctx->aux_images.image_length || fill_present) ? 1 : 0;
    // polarized run with no per-sample row but tau and no rendering: tau is integrated by the polarized transfer kernel
    const BlAuxImages &AI = ctx->aux_images;
    const bool rows_only = ctx->polarized && ctx->render_num_images == 0 && !fill_present && !(AI.image_time || AI.image_length || AI.image_lambda
        || AI.image_emission || AI.image_lambda_ave || AI.image_emission_ave || AI.image_tau_int || AI.image_crossings);
    sa.aux_record_unused = rows_only ? 1 : 0;
    for (int mu = 0; mu < 4; mu++) sa.cam_x[mu] = ctx->frame.cam_x[mu];

    const double snapshot_time = slow ? p.slow_t_start + p.slow_dt * ctx->snapshot : 0.0;   // simulation_reader.cpp:214
    if (slow) {
      const int chunk_size = p.slow_chunk_size;
      std::vector<unsigned long long> table(3 * static_cast<size_t>(chunk_size) + 4, 0ull);
      for (int n = 0; n < chunk_size; n++) {
        const bl_ctx::SlowSlice &slice = ctx->slow_slices[n];
        table[n] = reinterpret_cast<unsigned long long>(slice.cells.ptr);
        table[chunk_size + n] = reinterpret_cast<unsigned long long>(slice.kappa.ptr);
        std::memcpy(&table[2 * static_cast<size_t>(chunk_size) + n], &slice.time, sizeof(double));
      }
      ctx->d_slow_table.Ensure(table.size());
      Check(hipMemcpyAsync(ctx->d_slow_table.ptr, table.data(), table.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, stream), "slow-light table upload");
      ctx->d_ray_extrap.Ensure(n_rays);
      Check(hipMemsetAsync(ctx->d_ray_extrap.ptr, 0, n_rays * sizeof(unsigned int), stream), "slow-light flags reset");
      sa.slow.n = chunk_size;
      sa.slow.interp = p.slow_interp ? 1 : 0;
      sa.slow.snapshot_time = snapshot_time;
      sa.slow.cells = reinterpret_cast<const float *const *>(ctx->d_slow_table.ptr);
      sa.slow.kappa = reinterpret_cast<const float *const *>(ctx->d_slow_table.ptr + chunk_size);
      sa.slow.times = reinterpret_cast<const double *>(ctx->d_slow_table.ptr + 2 * static_cast<size_t>(chunk_size));
      sa.slow.extrap_max = ctx->d_slow_table.ptr + 3 * static_cast<size_t>(chunk_size);
    }

    BlTransferArgs xa{};
    xa.frequencies = ctx->d_freq.ptr;
    xa.n_nu = n_nu;
    xa.ray_max_steps = max_steps;
    xa.fallback_nan = p.fallback_nan;
    xa.model_type = p.model_type;
    xa.affine = fast ? 1 : 0;
    xa.n_rays_total = n_rays;
    xa.image = image;
    xa.out_sample_num = out_num;
    xa.out_flags = out_flags;
    xa.aux_images = ctx->aux_images;
    xa.aux_images.polarized_rows_only = rows_only ? 1 : 0;
    xa.x_unit = kGGMsun * ctx->frame.mass_msun / (kC * kC);
    xa.t_unit = xa.x_unit / kC;   // unpolarized.cpp:43
    xa.render_params = ctx->render_num_images > 0 ? ctx->d_render_params.ptr : nullptr;
    xa.render = render_out;
    if (ctx->polarized) {
      xa.camera_pos = cam_pos;
      xa.camera_dir = cam_dir;
      xa.st = ctx->st;
      xa.simulation_coord = p.simulation_coord == BL_COORD_FMKS ? BL_COORD_SKS : p.simulation_coord;
      xa.rotation_split = p.image_rotation_split ? 1 : 0;
      for (int mu = 0; mu < 4; mu++) {
        xa.cam_u_con[mu] = ctx->frame.u_con[mu];
        xa.cam_u_cov[mu] = ctx->frame.u_cov[mu];
        xa.cam_vert_con_c[mu] = ctx->frame.vert_con_c[mu];
      }
      for (int c = 0; c < 7; c++) sa.power_pol[c] = ctx->power_pol[c];
      sa.plasma_gamma_min = p.plasma_power_frac != 0.0 ? p.plasma_gamma_min : 0.0;
    }

    // Locate kernel: 256-thread workgroups. Alone (single chunk) it runs 4 waves per SIMD; when chunks are
    // pipelined it shares each SIMD with one 328-register wave of the next chunk's geodesic kernel, which
    // leaves room for exactly one 128-register locate wave - one workgroup per CU, so that whichever of the
    // two kernels is dispatched first cannot fill the register file and lock the other out.
    const int locate_grid_alone = ctx->num_cus * 4 * 4;
    const int locate_grid_shared = ctx->num_cus;
    const int shade_grid = ctx->num_cus * 2 * 4;  // 256-thread workgroups, 2 waves per SIMD, x4 for tail balance

    bl_stats st{};
    st.n_rays = n_rays;
    st.n_chunks = n_chunks;

    // Chunk c uses scratch set c % 2. Two streams:
    //   stream_geo: [wait until set c % 2 was drained by chunk c - 2]  geodesic(c)
    //   stream:     [wait for geodesic(c)]  locate(c)  coefficients(c)  transfer(c)  counters -> host
    // so geodesic(c + 1) overlaps the shading of chunk c. The geodesic kernel holds one 328-register
    // wave per SIMD; a 128-register locate wave fits beside it and issues into the slots its dependent
    // fp64 chains leave idle, and the coefficient / transfer waves take over SIMDs as geodesic waves retire.
    struct {   // geodesic checkpoint being assembled: samples of every pixel, far -> near, packed
      std::vector<int32_t> sample_num;
      std::vector<uint8_t> flags;
      std::vector<double> factors, pos, dir, len;
      std::vector<size_t> offset;
    } save;
    hipEvent_t *ev = ctx->events.data();
    hipEvent_t ev_setup = ev[static_cast<size_t>(n_chunks) * kEventsPerChunk];
    Check(hipEventRecord(ev_setup, stream), "event");            // uploads above were queued on `stream`
    Check(hipStreamWaitEvent(stream_geo, ev_setup, 0), "stream wait");
    const size_t n_counters = BL_CNT_COUNT + 4;
    for (int c = 0; c < n_chunks; c++) {
      const long long begin = static_cast<long long>(c) * chunk;
      const int rays = static_cast<int>(std::min<long long>(chunk, n_rays - begin));
      bl_ctx::ChunkSlot &sl = ctx->slot[c % n_slots];
      hipEvent_t *e = ev + static_cast<size_t>(c) * kEventsPerChunk;
      ta.chunk_begin = begin;
      ta.chunk_rays = rays;
      ta.records_hot = sl.d_records_hot.ptr;
      ta.records_cold = sl.d_records_cold.ptr;
      ta.counters = sl.d_counters.ptr;
      ta.ray_kt = sl.d_ray_kt.ptr;
      ta.ray_factor = sl.d_ray_factor.ptr;
      ta.ray_sample_num = sl.d_ray_sample_num.ptr;
      ta.ray_flags = sl.d_ray_flags.ptr;
      ta.ray_out_index = sl.d_ray_out_index.ptr;
      sa.records_hot = sl.d_records_hot.ptr;
      sa.records_cold = sl.d_records_cold.ptr;
      sa.located = simulation ? sl.d_located.ptr : nullptr;
      sa.located_tag = simulation ? sl.d_located_tag.ptr : nullptr;
      sa.tag_in_record = fast ? 1 : 0;
      sa.freq_split = freq_split ? 1 : 0;
      sa.coef_split = coef_split ? 1 : 0;
      sa.freq_inputs = freq_split ? sl.d_freq_inputs.ptr : nullptr;
      xa.freq_inputs = sa.freq_inputs;
      sa.counters_in = sl.d_counters.ptr;
      sa.counters = sl.d_counters.ptr;
      sa.ray_kt = sl.d_ray_kt.ptr;
      sa.ray_factor = sl.d_ray_factor.ptr;
      sa.transfer = sl.d_transfer.ptr;
      xa.chunk_rays = rays;
      xa.transfer = sl.d_transfer.ptr;
      xa.ray_sample_num = sl.d_ray_sample_num.ptr;
      xa.ray_flags = sl.d_ray_flags.ptr;
      xa.ray_out_index = sl.d_ray_out_index.ptr;
      xa.stats = sl.d_counters.ptr + BL_CNT_COUNT;
      ta.sample_t = need_time ? sl.d_sample_t.ptr : nullptr;
      sa.aux = aux ? sl.d_aux.ptr : nullptr;
      sa.sample_t = ta.sample_t;
      if (coef_split) sa.coef_inputs = sl.d_coef_inputs.ptr;
      if (ctx->polarized) {
        sa.pol_samples = sl.d_pol_samples.ptr;
        sa.pol_coeffs = sl.d_pol_coeffs.ptr;
        sa.coef_inputs = sl.d_coef_inputs.ptr;
        xa.pol_samples = sl.d_pol_samples.ptr;
        xa.pol_coeffs = sl.d_pol_coeffs.ptr;
        xa.pol_matrix = matrix_transport ? sl.d_pol_matrix.ptr : nullptr;
      }
      sa.anchors = block_interp ? sl.d_anchors.ptr : nullptr;
      sa.redo_list = (fast || ctx->polarized) ? sl.d_redo.ptr : nullptr;
      sa.redo_capacity = (fast || ctx->polarized) ? redo_capacity : 0;
      if (slow) {
        sa.slow.frac = sl.d_slow_frac.ptr;
        sa.slow.ray_extrap = ctx->d_ray_extrap.ptr + begin;
      }
      sa.ray_flags = sl.d_ray_flags.ptr;
      xa.aux = sa.aux;
      xa.ray_factor = sl.d_ray_factor.ptr;

      // ---- geodesic stream
      // the scratch set (and its counters) is free again once the chunk that used it before has had its
      // counters copied out - event 6, recorded behind that copy, not event 5 in front of it
      if (c >= n_slots) Check(hipStreamWaitEvent(stream_geo, (e - n_slots * kEventsPerChunk)[6], 0), "stream wait");
      Check(hipMemsetAsync(sl.d_counters.ptr, 0, n_counters * sizeof(unsigned long long), stream_geo), "counter reset");
      Check(hipEventRecord(e[0], stream_geo), "event");
      if (geo_load) {
        // LoadGeodesics(): the chunk's sample records come from the file instead of the geodesic kernel. The file holds
        // them far -> near (ReverseGeodesics) with the renormalised momentum; records are near -> far, so sample n of a ray
        // is entry num - 1 - n, and len = -sample_len.
        Check(hipStreamSynchronize(stream), "kernel execution");       // the scratch set may still be in use
        Check(hipStreamSynchronize(stream_geo), "kernel execution");
        const bl_ctx::Checkpoint &ck = ctx->checkpoint;
        const size_t steps = static_cast<size_t>(ck.num_steps);
        std::vector<BlSampleHot> hot;
        std::vector<BlSampleCold> cold;
        std::vector<double> sample_t, ray_kt(rays), ray_factor(rays);
        std::vector<int> ray_num(rays);
        std::vector<unsigned char> ray_flags(rays);
        std::vector<long long> ray_out(rays);
        for (int q = 0; q < rays; q++) {
          const long long ray = begin + q;
          const size_t m = d->pixel_map != nullptr ? static_cast<size_t>(d->pixel_map[ray]) : static_cast<size_t>(ray);
          if (m >= ck.sample_num.size()) throw Failure{BL_E_ARG, "pixel_map names a pixel the geodesic checkpoint does not hold."};
          const int num = ck.sample_num[m];
          ray_kt[q] = ck.camera_dir[4 * m];
          ray_factor[q] = ck.factors[m];
          ray_num[q] = num;
          ray_flags[q] = ck.flags[m];
          ray_out[q] = ray;
          for (int n = 0; n < num; n++) {
            const size_t at = m * steps + static_cast<size_t>(num - 1 - n);
            BlSampleHot h;
            h.x = ck.pos[4 * at + 1]; h.y = ck.pos[4 * at + 2]; h.z = ck.pos[4 * at + 3];
            h.ray = static_cast<uint32_t>(q);
            h.n = static_cast<uint32_t>(n);
            BlSampleCold c;
            c.kx = ck.dir[4 * at + 1]; c.ky = ck.dir[4 * at + 2]; c.kz = ck.dir[4 * at + 3];
            c.len = -ck.len[at];
            hot.push_back(h);
            cold.push_back(c);
            sample_t.push_back(ck.pos[4 * at]);
          }
        }
        const unsigned long long n_loaded = hot.size();
        if (n_loaded > 0) {
          Check(hipMemcpy(sl.d_records_hot.ptr, hot.data(), n_loaded * sizeof(BlSampleHot), hipMemcpyHostToDevice), "checkpoint upload");
          Check(hipMemcpy(sl.d_records_cold.ptr, cold.data(), n_loaded * sizeof(BlSampleCold), hipMemcpyHostToDevice), "checkpoint upload");
          Check(hipMemcpy(sl.d_sample_t.ptr, sample_t.data(), n_loaded * sizeof(double), hipMemcpyHostToDevice), "checkpoint upload");
        }
        Check(hipMemcpy(sl.d_ray_kt.ptr, ray_kt.data(), rays * sizeof(double), hipMemcpyHostToDevice), "checkpoint upload");
        Check(hipMemcpy(sl.d_ray_factor.ptr, ray_factor.data(), rays * sizeof(double), hipMemcpyHostToDevice), "checkpoint upload");
        Check(hipMemcpy(sl.d_ray_sample_num.ptr, ray_num.data(), rays * sizeof(int), hipMemcpyHostToDevice), "checkpoint upload");
        Check(hipMemcpy(sl.d_ray_flags.ptr, ray_flags.data(), rays, hipMemcpyHostToDevice), "checkpoint upload");
        Check(hipMemcpy(sl.d_ray_out_index.ptr, ray_out.data(), rays * sizeof(long long), hipMemcpyHostToDevice), "checkpoint upload");
        Check(hipMemcpy(sl.d_counters.ptr + BL_CNT_RECORDS, &n_loaded, sizeof n_loaded, hipMemcpyHostToDevice), "checkpoint upload");
      } else {
        Check(bl_launch_geodesic(&ta, p.ray_integrator, std::min(geo_grid, (rays + 63) / 64), stream_geo), "geodesic kernel launch");
      }
      Check(hipEventRecord(e[1], stream_geo), "event");
      // ---- shading stream
      Check(hipStreamWaitEvent(stream, e[1], 0), "stream wait");
      Check(hipEventRecord(e[2], stream), "event");
      if (simulation) {
        const bool shares_gpu = stream_geo != stream && c + 1 < n_chunks;   // geodesic(c + 1) is running beside it
        Check(bl_launch_locate(&sa, shares_gpu ? locate_grid_shared : locate_grid_alone, ctx->lds_table_bytes, stream), "locate kernel launch");
      }
      Check(hipEventRecord(e[3], stream), "event");
      if (fast) Check(bl_launch_shade_fast(&sa, shade_grid, stream), "coefficient kernel launch");
      else Check(bl_launch_shade(&sa, p.model_type, shade_grid, stream), "coefficient kernel launch");
      if (ctx->polarized) Check(bl_launch_polarized_coefficients(&sa, ctx->num_cus * 8, stream), "polarized coefficient kernel launch");
      if (coef_split) Check(bl_launch_coefficients_freq(&sa, ctx->num_cus * 16, stream), "per-frequency coefficient kernel launch");
      Check(hipEventRecord(e[4], stream), "event");
      Check(aux ? bl_launch_transfer_aux(&xa, stream) : (freq_split ? bl_launch_transfer_freq(&xa, stream) : bl_launch_transfer(&xa, stream)),
            "transfer kernel launch");
      if (ctx->polarized)
        Check(matrix_transport ? bl_launch_transfer_polarized_matrix(&xa, ctx->num_cus, stream) : bl_launch_transfer_polarized(&xa, stream),
              "polarized transfer kernel launch");
      Check(hipEventRecord(e[5], stream), "event");
      Check(hipMemcpyAsync(ctx->host_counters + static_cast<size_t>(c) * n_counters, sl.d_counters.ptr,
                           n_counters * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream), "counter download");
      Check(hipEventRecord(e[6], stream), "event");
      if (geo_save) {
        // SaveGeodesics(): bring this chunk's records back while its scratch set still holds them
        Check(hipStreamSynchronize(stream_geo), "kernel execution");
        Check(hipStreamSynchronize(stream), "kernel execution");
        unsigned long long n_written = 0;
        Check(hipMemcpy(&n_written, sl.d_counters.ptr + BL_CNT_RECORDS, sizeof n_written, hipMemcpyDeviceToHost), "checkpoint download");
        std::vector<BlSampleHot> hot(n_written);
        std::vector<BlSampleCold> cold(n_written);
        std::vector<double> sample_t(n_written), ray_kt(rays), ray_factor(rays);
        std::vector<int> ray_num(rays);
        std::vector<unsigned char> ray_flags(rays);
        std::vector<long long> ray_out(rays);
        if (n_written > 0) {
          Check(hipMemcpy(hot.data(), sl.d_records_hot.ptr, n_written * sizeof(BlSampleHot), hipMemcpyDeviceToHost), "checkpoint download");
          Check(hipMemcpy(cold.data(), sl.d_records_cold.ptr, n_written * sizeof(BlSampleCold), hipMemcpyDeviceToHost), "checkpoint download");
          Check(hipMemcpy(sample_t.data(), sl.d_sample_t.ptr, n_written * sizeof(double), hipMemcpyDeviceToHost), "checkpoint download");
        }
        Check(hipMemcpy(ray_kt.data(), sl.d_ray_kt.ptr, rays * sizeof(double), hipMemcpyDeviceToHost), "checkpoint download");
        Check(hipMemcpy(ray_factor.data(), sl.d_ray_factor.ptr, rays * sizeof(double), hipMemcpyDeviceToHost), "checkpoint download");
        Check(hipMemcpy(ray_num.data(), sl.d_ray_sample_num.ptr, rays * sizeof(int), hipMemcpyDeviceToHost), "checkpoint download");
        Check(hipMemcpy(ray_flags.data(), sl.d_ray_flags.ptr, rays, hipMemcpyDeviceToHost), "checkpoint download");
        Check(hipMemcpy(ray_out.data(), sl.d_ray_out_index.ptr, rays * sizeof(long long), hipMemcpyDeviceToHost), "checkpoint download");
        if (save.sample_num.empty()) {
          save.sample_num.assign(n_rays, 0);
          save.flags.assign(n_rays, 0);
          save.factors.assign(n_rays, 0.0);
          save.offset.assign(n_rays, 0);
        }
        std::vector<size_t> slot_offset(rays);
        for (int q = 0; q < rays; q++) {
          const size_t m = static_cast<size_t>(ray_out[q]);
          save.sample_num[m] = ray_num[q];
          save.flags[m] = ray_flags[q];
          save.factors[m] = ray_factor[q];
          save.offset[m] = save.len.size();
          slot_offset[q] = save.len.size();
          save.pos.resize(save.pos.size() + 4 * static_cast<size_t>(ray_num[q]));
          save.dir.resize(save.dir.size() + 4 * static_cast<size_t>(ray_num[q]));
          save.len.resize(save.len.size() + static_cast<size_t>(ray_num[q]));
        }
        for (unsigned long long r = 0; r < n_written; r++) {
          const BlSampleHot &h = hot[r];
          if (h.ray == BL_DEAD_RAY) continue;
          const int num = ray_num[h.ray];
          if (static_cast<int>(h.n) >= num) continue;
          const BlSampleCold &c = cold[r];
          // ReverseGeodesics (geodesics.cpp:820-842) behind the per-sample renormalisation (:352-371)
          const size_t at = slot_offset[h.ray] + static_cast<size_t>(num - 1 - static_cast<int>(h.n));
          const double kt = ray_kt[h.ray];
          const double factor = bl_renormalization_factor(ctx->st, h.x, h.y, h.z, kt, c.kx, c.ky, c.kz);
          save.pos[4 * at] = sample_t[r]; save.pos[4 * at + 1] = h.x; save.pos[4 * at + 2] = h.y; save.pos[4 * at + 3] = h.z;
          save.dir[4 * at] = kt; save.dir[4 * at + 1] = c.kx * factor; save.dir[4 * at + 2] = c.ky * factor; save.dir[4 * at + 3] = c.kz * factor;
          save.len[at] = -c.len;
        }
      }
    }
    Check(hipStreamSynchronize(stream_geo), "kernel execution");
    Check(hipStreamSynchronize(stream), "kernel execution");

    if (geo_save) {   // SaveGeodesics() (geodesic_checkpoint.cpp:28-59)
      std::vector<double> camera_pos(static_cast<size_t>(n_rays) * 4), camera_dir(static_cast<size_t>(n_rays) * 4);
      Check(hipMemcpy(camera_pos.data(), cam_pos, camera_pos.size() * sizeof(double), hipMemcpyDeviceToHost), "checkpoint download");
      Check(hipMemcpy(camera_dir.data(), cam_dir, camera_dir.size() * sizeof(double), hipMemcpyDeviceToHost), "checkpoint download");
      std::ofstream out(p.checkpoint_geodesic_file.s, std::ios_base::out | std::ios_base::binary);
      if (!out.is_open()) throw Failure{BL_E_INPUT, "Could not open geodesic checkpoint file."};
      const bl_camera_frame &f = ctx->frame;
      const double *vectors[7] = {f.cam_x, f.u_con, f.u_cov, f.norm_con, f.norm_con_c, f.hor_con_c, f.vert_con_c};
      for (const double *v : vectors) out.write(reinterpret_cast<const char *>(v), 4 * sizeof(double));
      const int n_pix = static_cast<int>(n_rays);
      int num_steps = 0;
      for (int32_t num : save.sample_num) num_steps = std::max(num_steps, static_cast<int>(num));
      WriteCheckpointHeader<double>(out, 4, n_pix, 1);
      out.write(reinterpret_cast<const char *>(camera_pos.data()), static_cast<std::streamsize>(camera_pos.size() * sizeof(double)));
      WriteCheckpointHeader<double>(out, 4, n_pix, 1);
      out.write(reinterpret_cast<const char *>(camera_dir.data()), static_cast<std::streamsize>(camera_dir.size() * sizeof(double)));
      WriteCheckpointHeader<double>(out, n_nu, 1, 1);
      out.write(reinterpret_cast<const char *>(ctx->frequencies.data()), static_cast<std::streamsize>(n_nu * sizeof(double)));
      WriteCheckpointHeader<double>(out, n_pix, 1, 1);
      out.write(reinterpret_cast<const char *>(save.factors.data()), static_cast<std::streamsize>(save.factors.size() * sizeof(double)));
      out.write(reinterpret_cast<const char *>(&num_steps), sizeof(int));
      WriteCheckpointHeader<uint8_t>(out, n_pix, 1, 1);
      out.write(reinterpret_cast<const char *>(save.flags.data()), static_cast<std::streamsize>(save.flags.size()));
      WriteCheckpointHeader<int32_t>(out, n_pix, 1, 1);
      out.write(reinterpret_cast<const char *>(save.sample_num.data()), static_cast<std::streamsize>(save.sample_num.size() * sizeof(int32_t)));
      // sample_pos, sample_dir (n_pix, n_steps, 4) and sample_len (n_pix, n_steps): a pixel's samples, then zeros (the
      // reference leaves the tail of sample_pos / sample_dir as allocated; nothing reads it)
      std::vector<double> row(static_cast<size_t>(num_steps) * 4);
      for (int which = 0; which < 2; which++) {
        const std::vector<double> &source = which == 0 ? save.pos : save.dir;
        WriteCheckpointHeader<double>(out, 4, num_steps, n_pix);
        for (int m = 0; m < n_pix; m++) {
          std::fill(row.begin(), row.end(), 0.0);
          std::copy(source.begin() + 4 * save.offset[m], source.begin() + 4 * (save.offset[m] + save.sample_num[m]), row.begin());
          out.write(reinterpret_cast<const char *>(row.data()), static_cast<std::streamsize>(row.size() * sizeof(double)));
        }
      }
      WriteCheckpointHeader<double>(out, num_steps, n_pix, 1);
      row.resize(num_steps);
      for (int m = 0; m < n_pix; m++) {
        std::fill(row.begin(), row.end(), 0.0);
        std::copy(save.len.begin() + save.offset[m], save.len.begin() + save.offset[m] + save.sample_num[m], row.begin());
        out.write(reinterpret_cast<const char *>(row.data()), static_cast<std::streamsize>(row.size() * sizeof(double)));
      }
      if (!out) throw Failure{BL_E_INPUT, "Could not write geodesic checkpoint file."};
    }
    float ms_geo = 0.0f, ms_locate = 0.0f, ms_shade = 0.0f, ms_transfer = 0.0f, ms_wall = 0.0f;
    unsigned long long total_samples = 0, total_flagged = 0, total_records = 0, total_gathers = 0, total_redo = 0, total_undefined = 0, max_num = 0;
    for (int c = 0; c < n_chunks; c++) {
      hipEvent_t *e = ev + static_cast<size_t>(c) * kEventsPerChunk;
      const unsigned long long *hc = ctx->host_counters + static_cast<size_t>(c) * n_counters;
      float ms = 0.0f;
      Check(hipEventElapsedTime(&ms, e[0], e[1]), "event time"); ms_geo += ms;
      Check(hipEventElapsedTime(&ms, e[2], e[3]), "event time"); ms_locate += ms;
      Check(hipEventElapsedTime(&ms, e[3], e[4]), "event time"); ms_shade += ms;
      Check(hipEventElapsedTime(&ms, e[4], e[5]), "event time"); ms_transfer += ms;
      if (hc[BL_CNT_OVERFLOW] != 0) throw Failure{BL_E_DEVICE, "Sample record buffer overflow."};
      if (hc[BL_CNT_INTERP_FAILED] != 0) throw Failure{BL_E_INPUT, "Grid interpolation failed."};   // simulation_sampling.cpp:1319
      if (hc[BL_CNT_UNDEFINED] != 0 && ctx->undefined_policy != BL_UNDEFINED_EDGE) {
        if (p.simulation_coord == BL_COORD_FMKS)
          throw Failure{BL_E_UNSUPPORTED, "FMKS sampling reached the last polar zone of the last azimuthal plane (or the last entry of the "
                                          "coordinate table), where the reference reads past its arrays (simulation_sampling.cpp:405-415, "
                                          ":809-819): no defined result to reproduce. bl_set_undefined_policy(BL_UNDEFINED_EDGE) uses the edge cell instead."};
        throw Failure{BL_E_UNSUPPORTED, "Inter-block interpolation reached an upper edge of the last MeshBlock, where the reference reads past the end "
                                        "of its cell-centre arrays (simulation_sampling.cpp:520-522): no defined result to reproduce. "
                                        "bl_set_undefined_policy(BL_UNDEFINED_EDGE) mirrors the last cell centre about the block's face instead."};
      }
      total_undefined += hc[BL_CNT_UNDEFINED];
      total_records += hc[BL_CNT_RECORDS];
      total_gathers += hc[BL_CNT_GATHERS];
      if (fast) total_redo += hc[BL_CNT_REDO];   // (polarized runs use the list for something else: bl_polarized_frame_kernel)
      total_samples += hc[BL_CNT_COUNT + 0];
      total_flagged += hc[BL_CNT_COUNT + 1];
      max_num = std::max<unsigned long long>(max_num, hc[BL_CNT_COUNT + 2]);
      st.launches_geodesic++;
      if (simulation) st.launches_locate++;
      st.launches_shade++;
      st.launches_transfer++;
    }
    Check(hipEventElapsedTime(&ms_wall, ev[0], (ev + static_cast<size_t>(n_chunks - 1) * kEventsPerChunk)[5]), "event time");

    if (!d->outputs_on_device) {
      if (n_q > 0) Check(hipMemcpy(d->image, image, static_cast<size_t>(n_q) * n_rays * sizeof(double), hipMemcpyDeviceToHost), "image download");
      if (d->sample_num != nullptr) Check(hipMemcpy(d->sample_num, out_num, n_rays * sizeof(int), hipMemcpyDeviceToHost), "sample_num download");
      if (d->sample_flags != nullptr) Check(hipMemcpy(d->sample_flags, out_flags, n_rays, hipMemcpyDeviceToHost), "flags download");
      if (d->camera_pos != nullptr) Check(hipMemcpy(d->camera_pos, cam_pos, static_cast<size_t>(n_rays) * 32, hipMemcpyDeviceToHost), "camera_pos download");
      if (d->camera_dir != nullptr) Check(hipMemcpy(d->camera_dir, cam_dir, static_cast<size_t>(n_rays) * 32, hipMemcpyDeviceToHost), "camera_dir download");
      if (ctx->render_num_images > 0)
        Check(hipMemcpy(d->render, render_out, static_cast<size_t>(ctx->render_num_images) * 3 * n_rays * sizeof(double), hipMemcpyDeviceToHost), "render download");
    }

    st.n_samples = static_cast<int64_t>(total_samples);
    st.n_samples_emitted = static_cast<int64_t>(total_records);
    st.n_gathers = static_cast<int64_t>(total_gathers);
    st.n_flagged = static_cast<int64_t>(total_flagged);
    st.max_sample_num = static_cast<int32_t>(max_num);
    const double bytes_per_gather = (simulation && !p.simulation_interp) ? 32.0 : 256.0;
    st.algorithmic_bytes = bytes_per_gather * static_cast<double>(total_gathers) + 13.0 * static_cast<double>(n_rays);
    st.ms_geodesic = geo_load ? 0.0f : ms_geo;   // nothing was integrated
    st.ms_locate = ms_locate;
    st.ms_shade = ms_shade;
    st.ms_transfer = ms_transfer;
    st.ms_total = ms_geo + ms_locate + ms_shade + ms_transfer;
    st.ms_wall = ms_wall;
    st.arithmetic = (fast || tolerant_polarized) ? BL_ARITH_TOLERANT : BL_ARITH_EXACT;
    st.n_deferred = static_cast<int64_t>(total_redo);
    ctx->stats = st;
    // Warning text of the reference (geodesics.cpp:389-394)
    if (total_flagged > 0)
      Warn(ctx, std::to_string(total_flagged) + " out of " + std::to_string(n_rays) + " geodesics terminate unexpectedly.");
    if (total_undefined > 0)   // BL_UNDEFINED_EDGE (this text has no counterpart in the reference)
      Warn(ctx, std::to_string(total_undefined) + " samples lie where the reference reads past its arrays; the edge cell was used for them.");
    if (slow) {   // simulation_sampling.cpp:553-617: pixels whose samples fall outside the window of files
      std::vector<unsigned int> flags(n_rays);
      unsigned long long maxima[4];
      Check(hipMemcpy(flags.data(), ctx->d_ray_extrap.ptr, n_rays * sizeof(unsigned int), hipMemcpyDeviceToHost), "slow-light flags download");
      Check(hipMemcpy(maxima, ctx->d_slow_table.ptr + 3 * static_cast<size_t>(p.slow_chunk_size), sizeof maxima, hipMemcpyDeviceToHost), "slow-light maxima download");
      long long count[4] = {0, 0, 0, 0};
      for (unsigned int f : flags)
        for (int e = 0; e < 4; e++) count[e] += (f >> e) & 1u;
      auto text = [&](int kind, const char *degree, const char *direction) {
        double by;
        std::memcpy(&by, &maxima[kind], sizeof(double));
        std::ostringstream message;
        message << "Snapshot " << ctx->snapshot << " at time " << snapshot_time << " requires " << degree << " extrapolation "
                << direction << " in time (" << count[kind] << "/" << n_rays << " pixels, by up to " << by << " gravitational times).";
        return message.str();
      };
      for (int e = 0; e < 4; e++) {
        ctx->stats_slow_count[e] = count[e];
        std::memcpy(&ctx->stats_slow_val[e], &maxima[e], sizeof(double));
      }
      if (count[1] > 0) throw Failure{BL_E_INPUT, text(1, "significant", "forward")};
      if (count[3] > 0) throw Failure{BL_E_INPUT, text(3, "significant", "backward")};
      if (count[0] > 0) Warn(ctx, text(0, "moderate", "forward"));
      if (count[2] > 0) Warn(ctx, text(2, "moderate", "backward"));
    }
  } catch (const Failure &failure) {
    return Fail(ctx, failure);
  }
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
  if (ctx->stream != nullptr) (void)hipStreamDestroy(ctx->stream);
  if (ctx->stream_geo != nullptr) (void)hipStreamDestroy(ctx->stream_geo);
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

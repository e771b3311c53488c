// bl_render.hip - bl_render(): the reference's GeodesicIntegrator::Integrate() + RadiationIntegrator::Integrate() for one
// level of rays (src/blacklight.cpp:93-94, 203-204) as a pipeline of HIP kernels over chunks of rays, and what hangs off
// it: geodesic checkpoints (geodesic_checkpoint.cpp:28-108), statistics, the reference's warning texts.
//
// One call = plan (which kernels, what a sample costs in HBM) -> scratch -> kernel arguments -> chunks -> outputs.
// A chunk is not sized on the host: every scratch array has one entry per sample record (or per kept sample), a scratch set
// holds `record_capacity` of each, and the geodesic kernel hands out rays only while the records of the rays in flight
// are sure to fit (BlTraceArgs::record_gate). What it did not get to is the next chunk. The benchmark frame - 704 samples
// per ray where ray_max_steps allows 2 000 - is one chunk this way; sized for the worst case it was two.
#include <sys/stat.h>

#include <thread>

#include "bl_ctx.h"

namespace {

// geodesic start / end, locate start, coefficient start, transfer start, end, counters copied to the host; [7 ... 9] the split or
// geodesic stage; [10, 11] polarized runs: frames built / transport matrices built (on the second stream)
constexpr int kEventsPerChunk = 12;

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

void EnsureRenderResources(bl_ctx *ctx) {
  const size_t need = 2 * kEventsPerChunk + 2;
  while (ctx->events.size() < need) {
    hipEvent_t e = nullptr;
    Check(hipEventCreate(&e), "hipEventCreate");
    ctx->events.push_back(e);
  }
  if (ctx->host_counters == nullptr)
    Check(hipHostMalloc(reinterpret_cast<void **>(&ctx->host_counters), 2 * BL_CNT_TOTAL * sizeof(unsigned long long), hipHostMallocDefault),
          "hipHostMalloc");
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

std::shared_ptr<const bl_ctx::Checkpoint> ReadGeodesicCheckpoint(const bl_params &p) {
  std::ifstream in(p.checkpoint_geodesic_file.s, std::ios_base::in | std::ios_base::binary);
  if (!in.is_open()) throw Failure{BL_E_INPUT, "Could not open geodesic checkpoint file."};
  auto loaded = std::make_shared<bl_ctx::Checkpoint>();
  bl_ctx::Checkpoint &c = *loaded;
  for (double (&v)[4] : c.frame) in.read(reinterpret_cast<char *>(v), 4 * sizeof(double));
  const size_t n_pix = static_cast<size_t>(p.camera_resolution) * p.camera_resolution;
  int dims[5];
  std::vector<double> &frequencies = c.frequencies;
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
  return loaded;
}

// LoadGeodesics() for this context. The file's contents are shared between the contexts of a process that load the same file for the
// same camera (path, size, modification time, resolution, frequencies, ray_max_steps): the table holds weak references, so the
// memory goes when the last context that uses it does.
void LoadGeodesicCheckpoint(bl_ctx *ctx) {
  static std::mutex table_lock;
  static std::map<std::string, std::weak_ptr<const bl_ctx::Checkpoint>> table;
  const bl_params &p = ctx->params;
  struct stat info {};
  std::string key = p.checkpoint_geodesic_file.s;
  if (stat(p.checkpoint_geodesic_file.s, &info) == 0)
    key += "|" + std::to_string(static_cast<long long>(info.st_size)) + "|" + std::to_string(static_cast<long long>(info.st_mtim.tv_sec)) + "."
        + std::to_string(static_cast<long long>(info.st_mtim.tv_nsec)) + "|" + std::to_string(static_cast<long long>(info.st_ctim.tv_sec)) + "."
        + std::to_string(static_cast<long long>(info.st_ctim.tv_nsec)) + "|" + std::to_string(static_cast<long long>(info.st_ino));   // (a file rewritten in place within a second is another file)
  key += "|" + std::to_string(p.camera_resolution) + "|" + std::to_string(p.image_num_frequencies) + "|" + std::to_string(p.ray_max_steps);
  // (The table's lock covers the look-up only. Two contexts that ask for the same file for the first time at the same moment both read
  // it - tens of GB at 1024^2 - and the second keeps the first one's copy; a context that loads another file never waits behind them.)
  std::shared_ptr<const bl_ctx::Checkpoint> loaded;
  {
    std::lock_guard<std::mutex> guard(table_lock);
    loaded = table[key].lock();
  }
  if (!loaded) {
    std::shared_ptr<const bl_ctx::Checkpoint> mine = ReadGeodesicCheckpoint(p);
    std::lock_guard<std::mutex> guard(table_lock);
    loaded = table[key].lock();
    if (!loaded) {
      loaded = mine;
      table[key] = loaded;
    }
  }
  bl_camera_frame &f = ctx->frame;
  double *vectors[7] = {f.cam_x, f.u_con, f.u_cov, f.norm_con, f.norm_con_c, f.hor_con_c, f.vert_con_c};
  for (int v = 0; v < 7; v++) std::memcpy(vectors[v], loaded->frame[v], 4 * sizeof(double));
  ctx->frequencies = loaded->frequencies;   // LoadGeodesics() replaces what InitializeCamera() would have computed
  ctx->checkpoint = loaded;
}

template <typename T>
void WriteCheckpointHeader(std::ofstream &out, int n1, int n2, int n3) {
  const int dims[5] = {n1, n2, n3, 1, 1};
  out.write(reinterpret_cast<const char *>(dims), sizeof dims);
}

// A geodesic checkpoint being assembled: samples of every pixel, far -> near, packed
struct CheckpointSave {
  std::vector<int32_t> sample_num;
  std::vector<uint8_t> flags;
  std::vector<double> factors, pos, dir, len;
  std::vector<size_t> offset;
};

struct SplitIncomplete {};   // BL_TAIL_SPLIT: the chunk ended before its last ray (RunChunks)
struct ReuseImpossible {};   // scratch for a render over the resident records could not be allocated beside them (EnsureScratch)

// Everything one bl_render call decides before its first kernel, and what its chunks add up to
struct RenderJob {
  bl_ctx *ctx = nullptr;
  const bl_render_desc *d = nullptr;
  // which path
  bool simulation = false, aux = false, slow = false, geo_load = false, geo_save = false, need_time = false, block_interp = false;
  bool fast = false, tolerant_polarized = false, matrix_transport = false, freq_split = false, coef_split = false;
  bool rows_only = false, fill_present = false;
  bool interleaved = false;   // sample records as one 64-byte array instead of two of 32-byte halves
  bool fast_formula = false;   // tolerant tier in formula mode: bl_shade_formula_fast_kernel
  bool tau_row = false;   // tolerant tier: an optical-depth image beside the intensities on the plain path (bl_tau_kernel)
  bool skip_shell = false;   // steps between the grid's outer edge and the camera's sphere leave no records (BlTraceArgs::skip_low)
  bool fused2 = false;  // tolerant tier, the benchmark's grids: the locate step runs inside the coefficient kernel (bl_shade_fused2_kernel, bl_shade_fused.hip)
  bool composed = false;   // ... writing one affine transfer map per ray segment instead of one per sample (BlShadeArgs::composed)
  bool exact_fused = false;   // exact tier, the same grids, plain image at one frequency: bl_shade_exact2_kernel locates its samples itself
  bool pol_fused = false;     // polarized runs over the same grids (either tier): bl_shade_polarized2_kernel locates its samples itself
  bool pol_coefficients_inside = false;   // ... and, at one frequency with thermal electrons only, evaluates their polarized coefficients itself
  bool locate_inside = false; // fused2 || exact_fused || pol_fused: no locate kernel, no located samples in HBM
  bool park = false;          // BL_TAIL_QUAD: the last rays of a chunk go to bl_geodesic_quad_kernel (BlTraceArgs::parked)
  bool split_long = false;    // BL_TAIL_SPLIT: rays predicted long on compute units of their own (bl_split_long_kernel)
  bool allow_split = true;    // false: the call is being rendered again after its split chunk closed the reservation gate early
  bool allow_reuse = true;    // false: ... after memory for a render over the resident records could not be had
  bool keepable = false;      // root level, geodesics integrated here: what this render leaves may serve the next (bl_set_geodesic_reuse)
  bool reuse = false;         // the resident records of an earlier render of this camera are shaded again: no geodesic stage
  bool reuse_located = false; // ... and its located samples: no locate kernel
  bool raster = false;        // large host outputs (many image rows): rays in pixel order, so that a chunk is a range of columns of
                              // every row and goes to the caller's buffer while the next chunk renders (DownloadChunk)
  bool chunk_downloads = false;   // ... and this call does download chunk by chunk (more than one chunk, or a first chunk that left rays)
  std::vector<std::thread> downloads;   // blocking copies into pageable memory, one host thread per chunk in flight
  std::vector<hipError_t> download_status;
  RenderJob() { download_status.reserve(4096); }
  ~RenderJob() {
    for (std::thread &t : downloads)
      if (t.joinable()) t.join();
  }
  std::vector<unsigned char> geo_key, located_key;
  int split_cus = 0;          // ... how many compute units
  double split_b_lo = 0.0, split_b_hi = 0.0;   // ... and which impact parameters
  size_t park_capacity = 0;
  int quad_grid = 0;          // waves of bl_geodesic_quad_kernel
  int n_nu = 0, n_q = 0, max_steps = 0;
  long long n_rays = 0, level_pixels = 0;
  size_t redo_capacity = 0;
  // scratch
  uint64_t bytes_per_record = 0;
  size_t record_capacity = 0;
  long long record_gate = 0;
  int n_slots = 1, geo_grid = 1, geo_waves_per_cu = 1;
  // outputs (device pointers: the caller's, or staging)
  double *image = nullptr, *cam_pos = nullptr, *cam_dir = nullptr, *render_out = nullptr;
  int *out_num = nullptr;
  unsigned char *out_flags = nullptr;
  const int *d_pixel_map = nullptr, *d_block_locs = nullptr;
  double snapshot_time = 0.0;
  // kernel arguments common to all chunks
  BlTraceArgs ta{};
  BlShadeArgs sa{};
  BlTransferArgs xa{};
  int locate_grid_alone = 0, locate_grid_shared = 0, shade_grid = 0;
  // a chunk in flight on scratch set k
  struct InFlight {
    bool busy = false;
    long long begin = 0;
    int rays = 0;
    long long done = -1;   // rays the chunk covered, once known
  } in_flight[2];
  // totals
  int n_chunks = 0;
  float ms_geo = 0.0f, ms_locate = 0.0f, ms_shade = 0.0f, ms_transfer = 0.0f;
  unsigned long long total_samples = 0, total_flagged = 0, total_records = 0, total_gathers = 0, total_redo = 0, total_undefined = 0, total_parked = 0, max_num = 0;
  unsigned long long debug_counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  CheckpointSave save;
  // checkpoint_sample_save: where every kept sample of the level sits on the grid, by pixel and reversed sample index
  bool sample_save = false;
  struct SampleSave {
    int per_sample = 4;                     // indices per sample: 4, or 8 x 4 with inter-block interpolation
    std::vector<int32_t> sample_num;        // [pixel]
    std::vector<size_t> offset;             // [pixel]: first entry of the pixel in the packed arrays below
    std::vector<int32_t> inds;              // [sample][per_sample]
    std::vector<double> fracs;              // [sample][3] (trilinear sampling)
    std::vector<uint8_t> nan, fallback;     // [sample]
  } sampling;
};

hipEvent_t *SlotEvents(RenderJob &job, int k) { return job.ctx->events.data() + static_cast<size_t>(k) * kEventsPerChunk; }

// ---- plan: validation of the call, the path it takes
void PlanJob(RenderJob &job) {
  bl_ctx *ctx = job.ctx;
  const bl_render_desc *d = job.d;
  const bl_params &p = ctx->params;
  job.simulation = p.model_type == BL_MODEL_SIMULATION;
  if (ctx->device == BL_DEVICE_NONE) throw Failure{BL_E_DEVICE, "Host-only context: no HIP device selected (the hot path has no CPU fallback)."};
  if (job.simulation && !ctx->have_grid) throw Failure{BL_E_STATE, "bl_render called before bl_set_grid."};
  if (d->n_rays <= 0 || (d->image == nullptr && ctx->image_num_quantities > 0))
    throw Failure{BL_E_ARG, "bl_render needs n_rays > 0 and an image buffer."};
  if (ctx->render_num_images > 0 && d->render == nullptr) throw Failure{BL_E_ARG, "bl_render needs a render buffer when render_num_images > 0."};
  if (d->n_rays > 0x7fffffffll) throw Failure{BL_E_ARG, "Too many rays in one bl_render call."};
  if (d->level < 0 || d->level > p.adaptive_max_level) throw Failure{BL_E_ARG, "Adaptive level out of range."};
  if (d->level > 0 && (d->block_locs == nullptr || d->n_blocks <= 0)) throw Failure{BL_E_ARG, "Refined level needs block_locs."};
  if (job.simulation && p.plasma_kappa_frac != 0.0 && !ctx->polarized) {
    if (!(ctx->undefined_policy & BL_UNDEFINED_KAPPA))
      throw Failure{BL_E_UNSUPPORTED, "Kappa-distribution electrons (plasma_kappa_frac != 0) in an unpolarized run: the reference's absorptivity reads "
                                      "kappa_aa_high_i, which it only initialises for polarized runs - no defined result to reproduce. "
                                      "bl_set_undefined_policy(BL_UNDEFINED_KAPPA) uses the polarized definition instead."};
    if (!ctx->kappa_warned)
      Warn(ctx, "Unpolarized kappa-distribution electrons: kappa_aa_high_i, which the reference leaves uninitialised here, is (3 / kappa)^4.75 + 0.6.");
    ctx->kappa_warned = true;
  }
  job.n_nu = p.image_num_frequencies;
  job.n_q = ctx->image_num_quantities;
  job.max_steps = p.ray_max_steps;
  job.n_rays = d->n_rays;
  job.aux = ctx->aux_images.any != 0;
  job.slow = job.simulation && p.slow_light_on;
  if (job.slow) {
    if (static_cast<int>(ctx->slow_slices.size()) != p.slow_chunk_size) throw Failure{BL_E_STATE, "Slow light: time slices not set."};
    for (const bl_ctx::SlowSlice &slice : ctx->slow_slices)
      if (!slice.set) throw Failure{BL_E_STATE, "Slow light: time slices not set."};
  }
  // geodesic checkpoints (root level only, like the reference's): load replaces the geodesic kernel by the file's
  // samples, save writes what the geodesic kernel produced in the reference's layout
  job.geo_load = p.checkpoint_geodesic_load && d->level == 0;
  job.geo_save = p.checkpoint_geodesic_save && d->level == 0;
  if (job.geo_load && !ctx->checkpoint) LoadGeodesicCheckpoint(ctx);
  job.need_time = (job.aux && ctx->aux_images.image_time) || job.slow || job.geo_load || job.geo_save;

  job.level_pixels = static_cast<long long>(p.camera_resolution) * p.camera_resolution;
  if (d->level > 0) job.level_pixels = static_cast<long long>(d->n_blocks) * p.adaptive_block_size * p.adaptive_block_size;
  if (d->pixel_map == nullptr && job.n_rays > job.level_pixels) throw Failure{BL_E_ARG, "n_rays exceeds the pixels of this level."};
  if (job.geo_save && (d->pixel_map != nullptr || job.n_rays != job.level_pixels))
    throw Failure{BL_E_ARG, "checkpoint_geodesic_save needs the whole root camera in one bl_render call."};
  if (job.geo_load && d->pixel_map != nullptr) {   // a rank's tiles can be served from the one file; the map must stay inside it
    const size_t n_pix = ctx->checkpoint->sample_num.size();
    for (long long ray = 0; ray < job.n_rays; ray++)
      if (d->pixel_map[ray] < 0 || static_cast<size_t>(d->pixel_map[ray]) >= n_pix)
        throw Failure{BL_E_ARG, "pixel_map names a pixel the geodesic checkpoint does not hold."};
  }
  job.block_interp = job.simulation && ctx->grid_dev.block_interp != 0;
  // SaveSampling() (sample_checkpoint.cpp:22-46): with the first image of the root level only (radiation_integrator.cpp:693-704)
  job.sample_save = job.simulation && p.checkpoint_sample_save && d->level == 0 && !ctx->sample_checkpoint_saved;
  if (job.sample_save && (d->pixel_map != nullptr || job.n_rays != job.level_pixels))
    throw Failure{BL_E_ARG, "checkpoint_sample_save needs the whole root camera in one bl_render call."};
  // Tolerant tier: plain unpolarized images of a simulation with thermal (and power-law) electrons in a curved
  // spacetime have the fast coefficient kernel; every other configuration is rendered in exact arithmetic whatever
  // bl_set_arithmetic() asked for (bl_stats.arithmetic says which tier ran)
  // (an optical-depth image as the only auxiliary row is the plain path plus one sum per ray: not an auxiliary run below)
  const BlAuxImages &rows = ctx->aux_images;
  const bool tau_only = job.aux && !ctx->polarized && p.image_light && rows.image_tau && ctx->render_num_images == 0 && !job.geo_load && !job.geo_save
      && !(rows.image_time || rows.image_length || rows.image_lambda || rows.image_emission || rows.image_lambda_ave || rows.image_emission_ave
           || rows.image_tau_int || rows.image_crossings);
  // (inter-block interpolation and slow light: bl_shade_fast_kernel behind their locate kernels, primitives by the exact tier's sampling)
  job.fast = ctx->arithmetic == BL_ARITH_TOLERANT && job.simulation && (!job.aux || tau_only) && !ctx->polarized && !(job.slow && job.block_interp)
      && p.plasma_kappa_frac == 0.0 && p.plasma_model != BL_PLASMA_CODE_KAPPA
      && !p.ray_flat && ctx->plasma_thermal_frac != 0.0
      && job.n_nu <= 1024;   // (its LDS table holds five numbers per frequency)
  if (job.fast && job.aux) {
    job.tau_row = true;
    job.aux = false;
  }
  // ... formula mode has a fast kernel of its own (plain images, no optional geometric cut)
  job.fast_formula = ctx->arithmetic == BL_ARITH_TOLERANT && !job.simulation && !job.aux
      && !(p.cut_omit_near || p.cut_omit_far || p.cut_omit_in >= 0.0 || p.cut_omit_out >= 0.0 || p.cut_midplane_theta != 0.0 || p.cut_midplane_z != 0.0 || p.cut_plane);
  // ... and the per-frequency coefficient kernel of polarized runs (frame, transport and coupling stay exact)
  job.tolerant_polarized = ctx->arithmetic == BL_ARITH_TOLERANT && ctx->polarized;
  // ... and transport matrices (bl_transport_matrix_kernel) instead of the ray-sequential tensor transport, in curved spacetimes
  job.matrix_transport = job.tolerant_polarized && !p.ray_flat && !(ctx->switches & BL_SWITCH_TENSOR_TRANSPORT);
  // Several frequencies in the fast path: per-sample factors (BlFreqInputs) instead of per-frequency transfer records,
  // evaluated by bl_transfer_freq_kernel with one lane per ray and frequency
  job.freq_split = job.fast && job.n_nu >= 4 && p.plasma_power_frac == 0.0 && !job.tau_row;   // (the factors are the thermal formulas')
  // Plain images of a spherical Kerr-Schild simulation with fallback values beyond the grid: nothing is recorded of the steps that
  // lie in the empty shell between the grid's outer edge and the camera's sphere (both tiers; the samples count as ever)
  job.skip_shell = job.simulation && !job.aux && !ctx->polarized && !job.slow && !job.geo_load && !job.geo_save && !job.sample_save
      && !job.need_time && p.simulation_coord == BL_COORD_SKS && !ctx->grid_dev.fmks && !p.fallback_nan && ctx->grid_outer_x1 > 0.0
      && p.ray_integrator == BL_INTEGRATOR_DP   // (the fixed-step steppers have no instantiation for it: bl_launch_geodesic)
      && ctx->grid_outer_x1 < p.camera_r && !(ctx->switches & BL_SWITCH_RECORD_EVERY_STEP);
  // The fast path over one grid (or equal blocks merged into one) with its coordinate tables in LDS, trilinear sampling and no
  // optional geometric cut locates its samples inside the coefficient kernel: no located samples in HBM at all
  // (one frequency - with four or more it ends at the sample's factors, which no frequency enters: BlFreqInputs - over a single block
  // with evenly spaced faces, bl_fused2_applicable; everything else goes through a locate kernel and bl_shade_fast_kernel)
  // (a mesh with refinement whose blocks and rows are evenly spaced has an instantiation of that kernel too - one frequency, composed
  // maps: bl_fused2_refined_applicable and the conditions of job.composed below)
  const bool records_every_sample = ctx->reproducible || (ctx->switches & BL_SWITCH_SAMPLE_RECORDS) != 0;
  const bool fused2_grid = ctx->grid_dev.n_blocks == 0
      ? ctx->lds_table_bytes > 0 && bl_fused2_applicable(&ctx->grid_dev, job.freq_split ? 1 : job.n_nu, job.n_rays) != 0
      : !job.freq_split && !records_every_sample && !job.skip_shell && bl_fused2_refined_applicable(&ctx->grid_dev, job.n_nu, job.n_rays) != 0;
  job.fused2 = job.fast && !job.tau_row && !job.slow && fused2_grid && !ctx->grid_dev.fmks && p.simulation_interp && p.plasma_power_frac == 0.0
      && p.simulation_coord == BL_COORD_SKS   // (its locate step is the spherical one: Cartesian grids go through the locate kernel)
      && !(p.cut_omit_near || p.cut_omit_far || p.cut_omit_in >= 0.0 || p.cut_omit_out >= 0.0 || p.cut_midplane_theta != 0.0 || p.cut_midplane_z != 0.0 || p.cut_plane)
      && !job.sample_save && !(ctx->switches & (BL_SWITCH_NO_FUSED_LOCATE | BL_SWITCH_SPLIT_RECORDS))   // (a sample checkpoint is made of the located samples)
      && !job.geo_load && !job.geo_save;  // (interleaved records whose momenta are not renormalised yet)
  // The exact tier's plain image at one frequency over such a grid: the locate step inside bl_shade_exact2_kernel (bit-identical
  // to bl_locate_plain_kernel + bl_shade_exact_kernel, whose conditions these are)
  job.exact_fused = !job.fast && job.simulation && !job.aux && !ctx->polarized && !job.slow && !job.block_interp && job.n_nu == 1
      && p.plasma_kappa_frac == 0.0 && p.plasma_power_frac == 0.0 && p.plasma_model != BL_PLASMA_CODE_KAPPA && !p.ray_flat
      && ctx->plasma_thermal_frac != 0.0 && !ctx->grid_dev.fmks && p.simulation_interp && p.simulation_coord == BL_COORD_SKS
      && !(p.cut_omit_near || p.cut_omit_far || p.cut_omit_in >= 0.0 || p.cut_omit_out >= 0.0 || p.cut_midplane_theta != 0.0 || p.cut_midplane_z != 0.0 || p.cut_plane)
      && !job.geo_load && !job.geo_save && !job.sample_save && !(ctx->switches & (BL_SWITCH_NO_FUSED_LOCATE | BL_SWITCH_SPLIT_RECORDS))
      && (ctx->grid_dev.n_blocks == 0 ? bl_fused2_applicable(&ctx->grid_dev, job.n_nu, job.n_rays) != 0 : bl_polarized2_refined_applicable(&ctx->grid_dev, job.n_rays) != 0);
  // Polarized runs over such a grid: the frame-and-inputs kernel with the locate step inside (bit-identical to bl_locate_plain_kernel +
  // bl_shade_kernel<polarized>, whose conditions these are; electron entropy from the grid is a ninth value it does not gather)
  // (... or a mesh with refinement whose tables the tolerant tier's fused kernel takes: the polarized kernel's locate step knows them too)
  job.pol_fused = ctx->polarized && job.simulation && !job.slow && !job.block_interp && p.plasma_model != BL_PLASMA_CODE_KAPPA && !p.ray_flat
      && !ctx->grid_dev.fmks && p.simulation_interp && p.simulation_coord == BL_COORD_SKS
      && !(p.cut_omit_near || p.cut_omit_far || p.cut_omit_in >= 0.0 || p.cut_omit_out >= 0.0 || p.cut_midplane_theta != 0.0 || p.cut_midplane_z != 0.0 || p.cut_plane)
      && !job.geo_load && !job.geo_save && !job.sample_save && !job.need_time && !(ctx->switches & (BL_SWITCH_NO_FUSED_LOCATE | BL_SWITCH_SPLIT_RECORDS))
      && (ctx->grid_dev.n_blocks == 0 ? bl_fused2_applicable(&ctx->grid_dev, 1, job.n_rays) != 0 : bl_polarized2_refined_applicable(&ctx->grid_dev, job.n_rays) != 0);
  job.interleaved = (job.fused2 || job.exact_fused || job.pol_fused || !job.simulation) && !job.geo_load && !job.geo_save && !job.sample_save && !(ctx->switches & BL_SWITCH_SPLIT_RECORDS);
  job.locate_inside = job.fused2 || job.exact_fused || job.pol_fused;
  // The benchmark's kernel also composes the affine maps of a ray's neighbouring samples before they leave it (the geodesic kernel
  // numbers the segments: BlTraceArgs::segment_rows)
  job.composed = job.fused2 && !job.freq_split && !records_every_sample;
  // (the geodesic kernel's instantiation that skips the shell has no register to number segments with: per-sample records there)
  if (job.skip_shell) job.composed = false;
  // The last rays of a chunk finished with a ray per quad of lanes (Dormand-Prince stepper without sample times; the instantiation
  // that skips the shell has no register for it): pays on frames whose last rays run for thousands of steps, costs the others
  const bool parkable = p.ray_integrator == BL_INTEGRATOR_DP && !job.need_time && !job.skip_shell && !job.geo_load;
  // (bl_set_tail_policy; BL_TAIL_AUTO: formula-mode frames - rays that circle for thousands of steps while the SIMDs around them idle,
  // configuration 2: 77 -> 67 ms - and not over a simulation grid, where the benchmark frame loses 6 ms to it. Bit-identical either way.)
  const bool quad_wanted = ctx->tail_policy == BL_TAIL_QUAD || (ctx->tail_policy == BL_TAIL_AUTO && !job.simulation && job.n_rays >= 64 * 64);
  job.park = parkable && (quad_wanted || (ctx->switches & BL_SWITCH_QUAD_EVERY_RAY) != 0);
  // BL_TAIL_SPLIT: the rays of a plane camera's root level whose impact parameter lies in a band around the photon ring's, stepped by
  // bl_geodesic_quad_kernel on compute units the other stepper is kept off
  const bool split_forced = ctx->tail_policy == BL_TAIL_SPLIT;
  // (BL_TAIL_AUTO: where the geodesic stage waits for single rays - up to eight rays per lane of a grid of one wave per SIMD: a share
  // of a frame tiled over two or more GPUs, measured 1.13 / 1.09 / 1.02 x at an eighth / a quarter / a half of the benchmark frame, 1.00 for
  // the whole - and the critical curve is the circle b = 3 sqrt(3) M: no spin)
  const bool split_auto = ctx->tail_policy == BL_TAIL_AUTO && !ctx->split_unavailable && ctx->st.bh_a == 0.0 && !p.ray_flat
      && job.n_rays >= 32768 && job.n_rays <= 8ll * 256 * ctx->num_cus;
  job.split_long = job.allow_split && parkable && !job.park && (split_forced || split_auto) && p.camera_type == BL_CAMERA_PLANE && d->level == 0;
  // ... and in the exact coefficient kernel (plain images): the frequency loop as lanes of bl_coefficients_freq_kernel
  job.coef_split = !job.fast && job.simulation && !job.aux && !ctx->polarized && job.n_nu >= 4;
  // (polarized runs list the samples without coefficients there - cut samples, cut cells - which are many more)
  job.redo_capacity = ctx->polarized ? (1u << 24) : (1u << 20);
  // (inter-block interpolation with the locate step inside the coefficient kernel: the samples with an anchor beyond their own block -
  // up to a quarter of them, bl_fused2_refined_applicable - are the exact pass's; room for an eighth of the samples the rays may have)
  if (job.fused2 && job.block_interp)
    job.redo_capacity = static_cast<size_t>(std::min<long long>(1ll << 29, std::max<long long>(1ll << 20, job.n_rays * static_cast<long long>(p.ray_max_steps) / 8)));
  // polarized run with no per-sample row but tau and no rendering: tau is integrated by the polarized transfer kernel
  bool fill_present = false;
  for (int n_i = 0; n_i < ctx->render_num_images; n_i++)
    for (int n_f = 0; n_f < p.render_num_features[n_i]; n_f++)
      if (p.render_type[n_i][n_f] == BL_RENDER_FILL) fill_present = true;
  job.fill_present = fill_present;
  const BlAuxImages &AI = ctx->aux_images;
  job.rows_only = ctx->polarized && ctx->render_num_images == 0 && !fill_present && !(AI.image_time || AI.image_length || AI.image_lambda
      || AI.image_emission || AI.image_lambda_ave || AI.image_emission_ave || AI.image_tau_int || AI.image_crossings);
  // configuration 4's case: no BlCoefInputs through HBM, no bl_polarized_coefficients_kernel launch (bl_shade_fused.hip: kCoefficients)
  job.pol_coefficients_inside = job.pol_fused && job.n_nu == 1 && job.rows_only && p.plasma_power_frac == 0.0 && p.plasma_kappa_frac == 0.0 && ctx->st.bh_a == 0.0;
  // Host outputs of a quarter of a GiB and more in eight rows or more (configuration 5: 64 frequencies): the rays are traced in pixel
  // order - not the 8 x 8 tiles, centre first, that make chunks drain faster - so that what a chunk finishes is a range of columns,
  // downloaded while the next chunk renders (the image rows of a 4096^2 x 64 frame are 8.6 GB: 0.7 s of PCIe that used to follow the
  // last kernel)
  job.raster = !d->outputs_on_device && d->level == 0 && d->pixel_map == nullptr && job.n_q >= 8
      && static_cast<uint64_t>(job.n_q) * static_cast<uint64_t>(job.n_rays) * sizeof(double) >= (256ull << 20) && !job.geo_load && !job.geo_save && !job.sample_save;
}

// ---- geodesics once per series (bl_set_geodesic_reuse; reference: blacklight.cpp:93-94 against its run loop :178-250, and the
// `first_time` sampling of radiation_integrator.cpp:693-704)
struct KeyWriter {
  std::vector<unsigned char> *out;
  template <typename T>
  void Put(const T &value) {
    const unsigned char *p = reinterpret_cast<const unsigned char *>(&value);
    out->insert(out->end(), p, p + sizeof(T));
  }
};

unsigned long long HashWords(const int32_t *data, size_t count) {
  unsigned long long h = 1469598103934665603ull;
  size_t at = 0;
  for (; at + 2 <= count; at += 2) {
    unsigned long long word;
    std::memcpy(&word, data + at, 8);
    h = (h ^ word) * 1099511628211ull;
    h ^= h >> 29;
  }
  if (at < count) h = (h ^ static_cast<unsigned int>(data[at])) * 1099511628211ull;
  return h;
}

// Everything the sample records of a root-level render and their layout depend on, value by value (no struct padding): two
// renders with equal keys would write the same records. The located samples depend on the grid's geometry and the locate
// step's settings besides.
void BuildReuseKeys(RenderJob &job) {
  bl_ctx *ctx = job.ctx;
  const bl_render_desc *d = job.d;
  const bl_params &p = ctx->params;
  job.geo_key.clear();
  KeyWriter key{&job.geo_key};
  key.Put(ctx->st.bh_m); key.Put(ctx->st.bh_a); key.Put(ctx->st.ray_flat);
  const bl_camera_frame &f = ctx->frame;
  for (const double (*v)[4] : {&f.cam_x, &f.u_con, &f.u_cov, &f.norm_con, &f.norm_con_c, &f.hor_con_c, &f.vert_con_c})
    for (int mu = 0; mu < 4; mu++) key.Put((*v)[mu]);
  key.Put(f.r_horizon); key.Put(f.r_terminate);
  key.Put(p.camera_type); key.Put(p.camera_width); key.Put(p.camera_r); key.Put(p.camera_resolution); key.Put(p.image_normalization);
  key.Put(p.ray_integrator); key.Put(p.ray_step); key.Put(p.ray_tol_abs); key.Put(p.ray_tol_rel); key.Put(p.ray_max_steps); key.Put(p.ray_max_retries);
  key.Put(d->level); key.Put(d->n_rays);
  const unsigned long long map_hash = d->pixel_map != nullptr ? HashWords(d->pixel_map, static_cast<size_t>(d->n_rays)) : 0ull;
  key.Put(d->pixel_map != nullptr ? 1 : 0); key.Put(map_hash);
  // the layout of the records and what the stepper leaves out of them
  key.Put(job.need_time ? 1 : 0); key.Put(job.interleaved ? 1 : 0); key.Put(job.composed ? 1 : 0); key.Put(job.skip_shell ? 1 : 0); key.Put(job.raster ? 1 : 0);
  key.Put(job.skip_shell ? ctx->grid_outer_x1 : 0.0);
  // who steps which rays (the records' order; bl_stats says it)
  key.Put(ctx->tail_policy); key.Put(ctx->switches); key.Put(ctx->overlap_chunks); key.Put(ctx->num_cus);
  key.Put(ctx->scratch_limit);   // (a caller that lowers the cap wants the memory back: the records are integrated again, in as many chunks as it takes)
  job.located_key = job.geo_key;
  KeyWriter located{&job.located_key};
  located.Put(ctx->grid_geometry); located.Put(ctx->undefined_policy); located.Put(job.fast ? 1 : 0); located.Put(job.block_interp ? 1 : 0);
  located.Put(ctx->guard_band); located.Put(p.simulation_interp); located.Put(p.simulation_coord);
}

// The root level's buffers and the other levels' change places (bl_ctx::ResidentGeodesics)
void SwapResidentBuffers(bl_ctx *ctx) {
  bl_ctx::ResidentGeodesics::Buffers &st = ctx->resident.store;
  bl_ctx::ChunkSlot &sl = ctx->slot[0];
  std::swap(st.records_hot, sl.d_records_hot);
  std::swap(st.records_cold, sl.d_records_cold);
  std::swap(st.sample_t, sl.d_sample_t);
  std::swap(st.located, sl.d_located);
  std::swap(st.located_tag, sl.d_located_tag);
  std::swap(st.anchors, sl.d_anchors);
  std::swap(st.ray_kt, ctx->d_ray_kt);
  std::swap(st.ray_factor, ctx->d_ray_factor);
  std::swap(st.ray_sample_num, ctx->d_ray_sample_num);
  std::swap(st.ray_skipped, ctx->d_ray_skipped);
  std::swap(st.ray_rows, ctx->d_ray_rows);
  std::swap(st.ray_flags, ctx->d_ray_flags);
  std::swap(st.ray_out_index, ctx->d_ray_out_index);
  std::swap(st.ray_offset, ctx->d_ray_offset);
  ctx->resident.parked = !ctx->resident.parked;
}

void DropResident(bl_ctx *ctx) {
  bl_ctx::ResidentGeodesics &res = ctx->resident;
  if (res.valid && res.parked) {   // the buffers set aside are the root level's: back to the device
    res.store.Free();
    res.parked = false;
  }
  res.valid = res.located_valid = false;
}

// Which way this render goes: over the resident records (a root-level render of the same camera left them), or integrating its
// own - in scratch set 0 as ever, with the resident records of the root level, if any, set aside first
void DecideReuse(RenderJob &job) {
  bl_ctx *ctx = job.ctx;
  bl_ctx::ResidentGeodesics &res = ctx->resident;
  job.keepable = ctx->geodesic_reuse != 0 && job.d->level == 0 && !job.geo_load;
  if (job.keepable) BuildReuseKeys(job);
  job.reuse = job.keepable && job.allow_reuse && res.valid && res.key == job.geo_key;
  if (job.reuse) {
    if (res.parked) SwapResidentBuffers(ctx);
    job.reuse_located = job.simulation && !job.locate_inside && !job.slow && res.located_valid && res.located_key == job.located_key;
    job.geo_save = false;   // (the render that integrated them wrote the file: geodesic_checkpoint.cpp is called once per run of the program)
  } else if (res.valid) {
    if (job.keepable) DropResident(ctx);            // another camera: this render's records take their place
    else if (!res.parked) SwapResidentBuffers(ctx);   // another level: it works in buffers of its own
  }
}

// After a render that integrated the root level's geodesics in one chunk: what it left is the resident set
void KeepResident(RenderJob &job) {
  bl_ctx *ctx = job.ctx;
  bl_ctx::ResidentGeodesics &res = ctx->resident;
  const bool located_here = job.simulation && !job.locate_inside && !job.slow;
  const unsigned long long *hc = ctx->host_counters;   // scratch set 0's, as CollectChunk read them
  if (job.keepable && !job.reuse) {
    res.valid = false;
    if (job.n_chunks != 1 || job.n_slots != 1) return;   // (the chunks overwrote one another's records: recomputed next time)
    res.valid = true;
    res.parked = false;
    res.key = job.geo_key;
    res.record_capacity = job.record_capacity;
    res.tail_policy = job.park ? BL_TAIL_QUAD : (job.split_long ? BL_TAIL_SPLIT : BL_TAIL_WIDE);
    res.n_parked = job.total_parked;
    res.n_flagged = job.total_flagged;
    std::memcpy(res.counters, hc, sizeof res.counters);
  } else if (!(job.reuse && located_here && !job.reuse_located)) {
    return;
  }
  // (here: a render that integrated the geodesics, or one that located the resident samples on a new geometry)
  res.located_valid = located_here;
  res.located_key = job.located_key;
  for (int c : {BL_CNT_GATHERS, BL_CNT_UNDEFINED, BL_CNT_INTERP_FAILED}) res.counters[c] = located_here ? hc[c] : 0ull;
  res.counters[BL_CNT_REDO] = 0ull;
  for (int c = BL_CNT_COUNT; c < BL_CNT_COUNT + 12; c++) res.counters[c] = 0ull;   // the transfer kernel's statistics, debug counters
}

// ---- scratch: what a sample record costs, how many fit, how many persistent waves trace rays into them
void PlanScratch(RenderJob &job) {
  bl_ctx *ctx = job.ctx;
  const bl_params &p = ctx->params;
  const int n_nu = job.n_nu;
  // per sample record (the arrays indexed by record slot) and per kept sample (the arrays indexed by ray_offset + n: never
  // more than records)
  job.bytes_per_record = sizeof(BlSampleHot) + sizeof(BlSampleCold)
      + ((job.simulation && !job.locate_inside) ? sizeof(BlLocated) + sizeof(unsigned long long) : 0)
      + (job.freq_split ? sizeof(BlFreqInputs) : (ctx->polarized ? 0 : sizeof(double2) * n_nu)) + (job.tau_row ? sizeof(double) * n_nu : 0)
      + (job.composed ? sizeof(double2) : 0)
      + ((job.aux && !job.rows_only) ? sizeof(BlAuxSample) : 0) + (job.need_time ? sizeof(double) : 0) + (job.slow ? sizeof(double) : 0)
      + (ctx->polarized ? sizeof(BlPolSample) + sizeof(BlCoefInputs) + 4 * sizeof(double2) * n_nu + (job.pol_coefficients_inside ? 1 : 0) : 0)
      + (job.coef_split ? sizeof(BlCoefInputs) : 0)
      + (job.matrix_transport ? BL_POL_MATRIX_DOUBLES * sizeof(double) : 0)
      + ((job.block_interp && !job.locate_inside) ? 8 * sizeof(unsigned int) : 0);
  if (job.reuse) {
    // over the resident records: the scratch set as the render that integrated them sized it, no stepper, nothing parked
    job.n_slots = 1;
    job.record_capacity = ctx->resident.record_capacity;
    job.record_gate = static_cast<long long>(job.record_capacity);
    job.geo_grid = 1;
    job.park = job.split_long = false;
    job.park_capacity = 0;
    job.quad_grid = 0;
    return;
  }
  const uint64_t per_slot_fixed = ((job.fast || job.fast_formula || ctx->polarized) ? job.redo_capacity * sizeof(unsigned long long) : 0) + BL_CNT_TOTAL * sizeof(unsigned long long);
  const bool park_every_ray = job.park && (ctx->switches & BL_SWITCH_QUAD_EVERY_RAY) != 0;
  const uint64_t per_ray = (2 + (job.geo_load ? 0 : BL_RAY_START_FIELDS) + (park_every_ray ? BL_PARK_DOUBLES : 0)) * sizeof(double) + (job.skip_shell ? 2 : 1) * sizeof(int) + 1 + 2 * sizeof(long long);
  // The budget is capped by what the device can actually give: 90 % of (free memory + what this context already holds
  // from earlier renders).
  uint64_t budget = ctx->scratch_limit;
  {
    size_t free_bytes = 0, total_bytes = 0;
    if (hipMemGetInfo(&free_bytes, &total_bytes) == hipSuccess) {
      // (the root level's resident records, set aside, are not this render's to use: they stay out of the budget)
      const uint64_t held = ctx->slot[0].Bytes() + ctx->slot[1].Bytes() + ctx->RayBytes();
      const uint64_t available = static_cast<uint64_t>(0.9 * static_cast<double>(free_bytes + held));
      if (available < budget) budget = available;
    }
  }
  const int waves_per_cu = bl_geodesic_occupancy(p.ray_integrator, job.need_time ? 1 : 0, ctx->st.bh_a == 0.0 ? 1 : 0, job.skip_shell ? 1 : 0);
  // No more lanes than half the rays: a lane that traces one ray only leaves its wave idling behind the longest of 64 rays, and
  // with fewer waves per SIMD each of them is faster - an eighth of the benchmark frame (131 072 rays) takes 4.3 ms on 1 024 waves,
  // 4.7 to 5.3 ms on 2 048 (and the coefficient kernel 5.0 instead of 5.2 ms over the more compact records)
  const long long max_grid = std::min<long long>(static_cast<long long>(ctx->num_cus) * waves_per_cu, std::max<long long>(1, (job.n_rays + 127) / 128));
  // (the waves of bl_geodesic_quad_kernel take blocks of record slots as well: a wave per SIMD)
  // (one to a SIMD: with three - as many as fit its registers - every ray runs at a third of the speed, the longest ones too, and
  // configuration 2 takes 94 ms instead of 67)
  const long long quad_waves = job.park ? static_cast<long long>(ctx->num_cus) * 4
      : (job.split_long ? 64ll * 4 : 0);   // (split: at most 64 compute units, sized below)
  const uint64_t worst_case = static_cast<uint64_t>(job.n_rays) * job.max_steps + static_cast<uint64_t>(max_grid + quad_waves) * BL_RECORD_BLOCK;
  const uint64_t fixed = per_ray * static_cast<uint64_t>(job.n_rays);
  auto capacity_for = [&](int n_slots) -> uint64_t {
    const uint64_t overhead = fixed + n_slots * per_slot_fixed;
    if (budget <= overhead) return 0;
    return std::min<uint64_t>((budget - overhead) / (static_cast<uint64_t>(n_slots) * job.bytes_per_record), worst_case);
  };
  // One scratch set; two of half the size each under bl_set_overlap() when one set cannot be sure to take the whole call,
  // so that the geodesic kernel of chunk c + 1 runs while chunk c is being shaded.
  job.n_slots = 1;
  uint64_t capacity = capacity_for(1);
  if (ctx->overlap_chunks && capacity < worst_case) {
    job.n_slots = 2;
    capacity = capacity_for(2);
  }
  // Kernels index a scratch set's records with 32 bits, and a grid-stride loop runs a few strides past the last record: the
  // capacity stays 2^22 records below 2^32. Clamped BEFORE the persistent grid and the reservation gate are derived from it
  // (they were once computed from the unclamped value: a gate beyond the buffers EnsureScratch allocates).
  capacity = std::min<uint64_t>(capacity, (1ull << 32) - (1ull << 22));
  // Persistent waves: every lane in flight holds ray_max_steps record slots until its ray ends, so no more lanes than the
  // buffer can cover at once. (A chunk still takes about capacity / samples-per-ray rays: a finished ray gives back what it
  // did not emit and the lanes that were refused ask again. Fewer waves than that would only trace the same rays more
  // slowly - 500 instead of 2 048 waves took the 64-frequency exact frame's geodesic stage from 37 to 183 ms.)
  const uint64_t per_wave = BL_RECORD_BLOCK + 64ull * static_cast<uint64_t>(job.max_steps);
  const long long grid = std::max<long long>(1, std::min<long long>(max_grid, static_cast<long long>((capacity - std::min<uint64_t>(capacity, static_cast<uint64_t>(quad_waves) * BL_RECORD_BLOCK)) / per_wave)));
  const long long gate = static_cast<long long>(capacity) - (grid + quad_waves) * BL_RECORD_BLOCK;
  if (gate < job.max_steps)
    throw Failure{BL_E_ARG, "Scratch budget too small: the sample records of a single ray (ray_max_steps of them) do not fit (bl_set_scratch_limit)."};
  job.record_capacity = static_cast<size_t>(capacity);
  job.record_gate = gate;
  job.geo_grid = static_cast<int>(grid);
  job.geo_waves_per_cu = waves_per_cu;
  job.quad_grid = static_cast<int>(quad_waves);
  // a lane parks at most one ray (its wave ends), unless every ray is parked
  job.park_capacity = job.park ? (park_every_ray ? static_cast<size_t>(job.n_rays) : static_cast<size_t>(grid) * 64) : 0;
  // (the split is decided once for all rays of the call: only where one chunk is sure to take them all)
  if (job.split_long && (job.n_slots != 1 || capacity < worst_case)) job.split_long = false;
  if (job.split_long) {
    // Which rays. Alone in a wave a ray of the benchmark camera takes 3.2 ms at b = 5.20 M, 2.3 ... 2.9 ms between 4.9 and 5.18, 2.3 ms
    // at 5.23 and 1.8 ms at 5.3 (tools/gpu_ray_length_by_radius.py): the band reaches 0.03 M beyond the critical curve and inwards
    // as far as the quad stepper's compute units hold it in ONE round of quads (a second round doubles its time) - a quad per ray,
    // 16 per wave, a wave per SIMD. The compute units: an eighth of the device, bits 0 ... num_cus / 8 - 1 of the mask - on MI355X
    // one CU of every shader engine of every XCD (tools/ubench/cu_mask_probe.hip), which leaves the other stepper's share of every
    // shader engine equal; other counts were measured and lose (uneven shader engines fill unevenly: docs/notebook.md section 5k).
    // How densely the call's rays cover the ring is counted on the host from every 1 / stride-th of them.
    const double centre = 5.196152422706632 * ctx->st.bh_m;
    const double scale = ctx->st.bh_m * p.camera_width / p.camera_resolution, half = 0.5 * p.camera_resolution - 0.5;
    const double ref_lo = centre - 0.3 * ctx->st.bh_m, ref_hi = centre + 0.1 * ctx->st.bh_m;
    const long long stride = std::max<long long>(1, job.n_rays / 32768);
    long long seen = 0, inside = 0;
    for (long long ray = 0; ray < job.n_rays; ray += stride, seen++) {
      const long long pixel = job.d->pixel_map != nullptr ? job.d->pixel_map[ray] : ray;
      const double u = (static_cast<double>(pixel % p.camera_resolution) - half) * scale, v = (static_cast<double>(pixel / p.camera_resolution) - half) * scale;
      const double b = std::sqrt(u * u + v * v);
      inside += (b >= ref_lo && b <= ref_hi) ? 1 : 0;
    }
    const double per_width = static_cast<double>(inside) * static_cast<double>(job.n_rays) / static_cast<double>(std::max<long long>(seen, 1)) / (ref_hi - ref_lo);
    job.split_cus = std::max(1, ctx->num_cus / 8);
    const double outer = 0.03 * ctx->st.bh_m;
    // (rounds of quads: one where the stage is a few rays per lane long - a second round would end after the other stepper -, more
    // where the other stepper has many rays per lane to get through and the quad stepper's compute units would stand idle meanwhile)
    const int rounds = job.n_rays > 2ll * 256 * ctx->num_cus ? 2 : 1;   // (a quarter of the frame: 14.0 ms with two, 14.6 with one, 14.5 with three)
    double width = per_width > 0.0 ? 0.9 * 64.0 * job.split_cus * rounds / per_width : 0.0;   // of the band
    width = std::min(width, 0.45 * ctx->st.bh_m);
    if (width >= 0.08 * ctx->st.bh_m) {
      job.split_b_lo = centre - (width - outer);
      job.split_b_hi = centre + outer;
    } else {
      job.split_long = false;   // no ring in these rays, or too many rays on it for an eighth of the device
    }
  }
  if (job.split_long) {
    job.park_capacity = static_cast<size_t>(job.n_rays);
    job.quad_grid = job.split_cus * 4;
  }
}

void EnsureScratchOnce(RenderJob &job);
// The budget counts what the context already holds as available (PlanScratch); arrays of an earlier render in another mode - per-
// frequency transfer records where this one wants per-sample factors - are not among those this render grows. When an allocation
// fails, everything the scratch sets hold goes back to the device and the render's own arrays are allocated afresh.
void EnsureScratch(RenderJob &job) {
  try {
    EnsureScratchOnce(job);
  } catch (const Failure &) {
    (void)hipGetLastError();
    if (job.reuse) throw ReuseImpossible{};   // (scratch set 0 holds the resident records: the call is planned again without them)
    DropResident(job.ctx);                    // (... set aside: they make room)
    job.ctx->slot[0].Free();
    job.ctx->slot[1].Free();
    EnsureScratchOnce(job);
  }
}

void EnsureScratchOnce(RenderJob &job) {
  bl_ctx *ctx = job.ctx;
  const size_t cap = job.record_capacity;
  const size_t n_nu = static_cast<size_t>(job.n_nu);
  for (int k = 0; k < job.n_slots; k++) {
    bl_ctx::ChunkSlot &sl = ctx->slot[k];
    if (job.interleaved) {
      sl.d_records_hot.Ensure(2 * cap);
    } else {
      sl.d_records_hot.Ensure(cap);
      sl.d_records_cold.Ensure(cap);
    }
    if (job.simulation && !job.locate_inside) {
      sl.d_located.Ensure(cap);
      sl.d_located_tag.Ensure(cap);
    }
    if (job.freq_split) sl.d_freq_inputs.Ensure(cap);   // instead of the transfer records
    else if (!ctx->polarized) sl.d_transfer.Ensure(cap * n_nu);   // (polarized runs: the eight coefficients of a sample side by side, d_pol_coeffs)
    if (job.composed) sl.d_composed.Ensure(cap);
    if (job.park || job.split_long) sl.d_parked.Ensure(job.park_capacity * BL_PARK_DOUBLES);
    if (job.tau_row) sl.d_tau_inc.Ensure(cap * n_nu);
    sl.d_counters.Ensure(BL_CNT_TOTAL);
    if (job.aux && !job.rows_only) sl.d_aux.Ensure(cap);   // (rows_only: nobody writes or reads the 96-byte records)
    if (job.need_time) sl.d_sample_t.Ensure(cap);
    if (job.slow) sl.d_slow_frac.Ensure(cap);
    if (ctx->polarized) {
      sl.d_pol_samples.Ensure(cap);
      if (job.matrix_transport) sl.d_pol_matrix.Ensure(cap * BL_POL_MATRIX_DOUBLES);
      sl.d_pol_coeffs.Ensure(cap * n_nu * 4);
      sl.d_coef_inputs.Ensure(cap);
      if (job.pol_coefficients_inside) sl.d_have_flags.Ensure(cap);
    }
    if (job.coef_split) sl.d_coef_inputs.Ensure(cap);
    if (job.block_interp && !job.locate_inside) sl.d_anchors.Ensure(cap * 8);   // (locate step inside: the exact pass keeps a sample's anchors in registers)
    if (job.fast || job.fast_formula || ctx->polarized) sl.d_redo.Ensure(job.redo_capacity);   // polarized runs: the samples whose frame bl_polarized_frame_kernel builds
  }
  const size_t n_rays = static_cast<size_t>(job.n_rays);
  ctx->d_ray_kt.Ensure(n_rays);
  ctx->d_ray_factor.Ensure(n_rays);
  ctx->d_ray_sample_num.Ensure(n_rays);
  if (job.skip_shell) ctx->d_ray_skipped.Ensure(n_rays);
  if (job.composed) ctx->d_ray_rows.Ensure(n_rays);
  ctx->d_ray_flags.Ensure(n_rays);
  ctx->d_ray_out_index.Ensure(n_rays);
  ctx->d_ray_offset.Ensure(n_rays);
  if (!job.geo_load) ctx->d_ray_start.Ensure(n_rays * BL_RAY_START_FIELDS);
  EnsureRenderResources(ctx);
}

// ---- inputs of the call that live in HBM: frequencies, pixel map, block list; output buffers (the caller's, or staging)
void StageInputsAndOutputs(RenderJob &job) {
  bl_ctx *ctx = job.ctx;
  const bl_render_desc *d = job.d;
  const bl_params &p = ctx->params;
  hipStream_t stream = ctx->stream;
  const long long n_rays = job.n_rays;
  if (ctx->caller_stream_set) {
    // bl_set_caller_stream: whatever the caller queued on its stream up to now (fills of the output buffers, a collective still
    // reading the previous frame out of them) is ahead of everything this call queues (both of its streams start behind `stream`) - a wait on the device, none on the host
    if (ctx->caller_event == nullptr) Check(hipEventCreateWithFlags(&ctx->caller_event, hipEventDisableTiming), "hipEventCreate");
    Check(hipEventRecord(ctx->caller_event, ctx->caller_stream), "event on the caller's stream");
    Check(hipStreamWaitEvent(stream, ctx->caller_event, 0), "stream wait");
  }
  ctx->d_freq.Ensure(job.n_nu);
  Check(hipMemcpyAsync(ctx->d_freq.ptr, ctx->frequencies.data(), job.n_nu * sizeof(double), hipMemcpyHostToDevice, stream), "freq upload");
  if (d->pixel_map != nullptr) {
    ctx->d_pixel_map.Ensure(n_rays);
    Check(hipMemcpyAsync(ctx->d_pixel_map.ptr, d->pixel_map, n_rays * sizeof(int), hipMemcpyHostToDevice, stream), "pixel_map upload");
    job.d_pixel_map = ctx->d_pixel_map.ptr;
  }
  if (d->level > 0) {
    ctx->d_block_locs.Ensure(static_cast<size_t>(d->n_blocks) * 2);
    Check(hipMemcpyAsync(ctx->d_block_locs.ptr, d->block_locs, static_cast<size_t>(d->n_blocks) * 2 * sizeof(int), hipMemcpyHostToDevice, stream), "block_locs upload");
    job.d_block_locs = ctx->d_block_locs.ptr;
  }
  job.image = d->image;
  job.cam_pos = d->camera_pos;
  job.cam_dir = d->camera_dir;
  job.out_num = d->sample_num;
  job.out_flags = d->sample_flags;
  if (!d->outputs_on_device) {
    ctx->d_image.Ensure(static_cast<size_t>(job.n_q) * n_rays);
    job.image = ctx->d_image.ptr;
    if (d->sample_num != nullptr) { ctx->d_out_sample_num.Ensure(n_rays); job.out_num = ctx->d_out_sample_num.ptr; }
    if (d->sample_flags != nullptr) { ctx->d_out_flags.Ensure(n_rays); job.out_flags = ctx->d_out_flags.ptr; }
    if (d->camera_pos != nullptr) { ctx->d_camera_pos.Ensure(static_cast<size_t>(n_rays) * 4); job.cam_pos = ctx->d_camera_pos.ptr; }
    if (d->camera_dir != nullptr) { ctx->d_camera_dir.Ensure(static_cast<size_t>(n_rays) * 4); job.cam_dir = ctx->d_camera_dir.ptr; }
  }
  if (ctx->polarized || job.geo_save) {   // the camera tetrad projection (and the checkpoint) need every ray's initial position and momentum
    if (job.cam_pos == nullptr) { ctx->d_camera_pos.Ensure(static_cast<size_t>(n_rays) * 4); job.cam_pos = ctx->d_camera_pos.ptr; }
    if (job.cam_dir == nullptr) { ctx->d_camera_dir.Ensure(static_cast<size_t>(n_rays) * 4); job.cam_dir = ctx->d_camera_dir.ptr; }
  }
  if (job.geo_load && (job.cam_pos != nullptr || job.cam_dir != nullptr)) {   // camera_pos / camera_dir come from the file as well
    if (ctx->caller_stream_set) Check(hipStreamSynchronize(ctx->caller_stream), "caller's stream");   // (blocking copies below: no stream orders them)
    std::vector<double> rows(static_cast<size_t>(n_rays) * 4);
    for (int which = 0; which < 2; which++) {
      double *target = which == 0 ? job.cam_pos : job.cam_dir;
      if (target == nullptr) continue;
      const std::vector<double> &source = which == 0 ? ctx->checkpoint->camera_pos : ctx->checkpoint->camera_dir;
      for (long long ray = 0; ray < n_rays; ray++) {
        const size_t m = d->pixel_map != nullptr ? static_cast<size_t>(d->pixel_map[ray]) : static_cast<size_t>(ray);
        for (int mu = 0; mu < 4; mu++) rows[4 * ray + mu] = source[4 * m + mu];
      }
      Check(hipMemcpy(target, rows.data(), rows.size() * sizeof(double), hipMemcpyHostToDevice), "checkpoint upload");
    }
  }
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
      }
    }
    rp.fill_present = job.fill_present ? 1 : 0;
    ctx->d_render_params.Ensure(1);
    Check(hipMemcpyAsync(ctx->d_render_params.ptr, &rp, sizeof(BlRenderDevice), hipMemcpyHostToDevice, stream), "render parameter upload");
    Check(hipStreamSynchronize(stream), "render parameter upload");   // rp is a local
    job.render_out = d->render;
    if (!d->outputs_on_device) {
      ctx->d_render.Ensure(static_cast<size_t>(ctx->render_num_images) * 3 * n_rays);
      job.render_out = ctx->d_render.ptr;
    }
  }
}

// ---- kernel arguments common to all chunks: the geodesic kernel's
void BuildTraceArgs(RenderJob &job) {
  bl_ctx *ctx = job.ctx;
  const bl_render_desc *d = job.d;
  const bl_params &p = ctx->params;
  BlTraceArgs &ta = job.ta;
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
  ta.skip_low = std::numeric_limits<double>::infinity();
  ta.skip_high = 0.0;
  if (job.skip_shell) {
    const double a = ctx->st.bh_a;
    ta.skip_low = std::sqrt(ctx->grid_outer_x1 * ctx->grid_outer_x1 * (1.0 + 1.0e-9) + a * a) * (1.0 + 1.0e-12);
    ta.skip_high = p.camera_r * (1.0 - 1.0e-9);
  }
  ta.ray_step = p.ray_step;
  ta.ray_tol_abs = p.ray_integrator == BL_INTEGRATOR_DP ? p.ray_tol_abs : 0.0;
  ta.ray_tol_rel = p.ray_integrator == BL_INTEGRATOR_DP ? p.ray_tol_rel : 0.0;
  ta.ray_max_steps = job.max_steps;
  ta.ray_max_retries = p.ray_integrator == BL_INTEGRATOR_DP ? p.ray_max_retries : 0;
  ta.n_rays_total = job.n_rays;
  ta.swizzle_tiles = (d->level == 0 && d->pixel_map == nullptr && p.camera_resolution % 8 == 0 && job.n_rays == job.level_pixels && !job.raster)
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
  ta.pixel_map = job.d_pixel_map;
  ta.block_locs = job.d_block_locs;
  ta.record_capacity = static_cast<long long>(job.record_capacity);
  ta.record_gate = job.record_gate;
  ta.camera_pos = job.cam_pos;
  ta.camera_dir = job.cam_dir;
  ta.ray_start_stride = job.n_rays;
}

// Power-law and kappa-distribution constants of the coefficient formulas (simulation_coefficients.cpp:54-193); pow / exp / log
// and K_nu are the pinned ones, tgamma the host libm's
void FillElectronConstants(RenderJob &job, BlPlasmaDevice &pl, BlShadeCold &cold) {
  bl_ctx *ctx = job.ctx;
  const bl_params &p = ctx->params;
  pl.power_frac = p.plasma_power_frac;
  pl.plasma_p = 0.0;
  pl.power_jj = pl.power_aa = 0.0;
  if (p.plasma_power_frac != 0.0) {
    // simulation_coefficients.cpp:54-66 (unpolarized part)
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
    // simulation_coefficients.cpp:82-193 for a polarized run
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
}

// ---- the locate / coefficient kernels'
void BuildShadeArgs(RenderJob &job) {
  bl_ctx *ctx = job.ctx;
  const bl_params &p = ctx->params;
  hipStream_t stream = ctx->stream;
  BlShadeArgs &sa = job.sa;
  sa.st = ctx->st;
  BlShadeCold cold;
  std::memset(static_cast<void *>(&cold), 0, sizeof cold);   // (padding too: the block is compared byte for byte below)
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
  sa.plasma.fallback_nan = p.fallback_nan;   // (formula mode too: a flagged ray's coefficients are NaN there as well, formula_coefficients.cpp:51-59)
  if (job.simulation) {
    BlPlasmaDevice &pl = sa.plasma;
    pl.d_unit = p.simulation_rho_cgs;                       // simulation_coefficients.cpp:237-239
    pl.e_unit = pl.d_unit * kC * kC;
    pl.b_unit = blm_sqrt(4.0 * kPi * pl.e_unit);
    pl.plasma_mu = p.plasma_mu;
    pl.plasma_ne_ni = p.plasma_ne_ni;
    pl.plasma_rat_low = p.plasma_rat_low;
    pl.plasma_rat_high = p.plasma_rat_high;
    pl.plasma_thermal_frac = ctx->plasma_thermal_frac;
    FillElectronConstants(job, pl, cold);
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
    pl.kappa_frac_zero = p.plasma_kappa_frac == 0.0 ? 1 : 0;
    pl.kappa_unpolarized = (p.plasma_kappa_frac != 0.0 && !ctx->polarized) ? 1 : 0;
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
      pl.any_cell_cut = 0;
      for (int c = 0; c < 14; c++) {
        const bool active = cuts[c] >= 0.0;
        if (active) pl.cut_mask |= 1 << c;
        if (active) pl.any_cell_cut = 1;
        // the fast kernels compare in code units (rho, rho for n_e, p, k T_e for Theta_e, |b| in code units; sigma and 1 / beta have
        // none): the thresholds are scaled once here, the guard band is relative and scales with them
        const double n_e_per_rho = pl.d_unit / (p.plasma_mu * kMp * (1.0 + 1.0 / p.plasma_ne_ni));
        const double to_code[7] = {1.0 / pl.d_unit, 1.0 / n_e_per_rho, 1.0 / pl.e_unit, kMe * kC * kC, 1.0 / pl.b_unit, 1.0, 1.0};
        const double scaled = cuts[c] * to_code[c >> 1];
        cold.fast_cut[c] = active ? scaled : 0.0;
        cold.fast_cut_lo[c] = active ? scaled * (1.0 - ctx->guard_band) : 0.0;
        cold.fast_cut_hi[c] = active ? scaled * (1.0 + ctx->guard_band) : 0.0;
      }
      {
        const bool use_p = p.plasma_use_p != 0;
        const double g0 = use_p ? 1.0 : 1.0 / (ctx->grid_meta.plasma_gamma - 1.0);
        const double g1 = use_p ? 1.0 : 1.0 / (ctx->grid_meta.plasma_gamma_i - 1.0);
        const double g2 = use_p ? 1.0 : 1.0 / (ctx->grid_meta.plasma_gamma_e - 1.0);
        const double n_e_per_rho = pl.d_unit / (p.plasma_mu * kMp * (1.0 + 1.0 / p.plasma_ne_ni));
        const double nu_c_over_b = kE * pl.b_unit / (2.0 * kPi * kMe * kC);
        sa.fast_k[0] = (1.0 + p.plasma_ne_ni) * p.plasma_mu * kMp * pl.e_unit / pl.d_unit * g0;
        sa.fast_k[1] = p.plasma_rat_high * g1;
        sa.fast_k[2] = p.plasma_rat_low * g1;
        sa.fast_k[3] = p.plasma_ne_ni * g2;
        sa.fast_k[4] = (kMe * kC * kC) * (kMe * kC * kC) * 4.5 / nu_c_over_b;
        sa.fast_k[5] = ctx->plasma_thermal_frac * n_e_per_rho * kE * kE * nu_c_over_b / kC * (kSqrt2 * kPi / 27.0);
        sa.fast_k[6] = n_e_per_rho;
        sa.fast_k[7] = nu_c_over_b;
      }
      sa.fast_angle_band = std::max(1.0e-12, ctx->guard_band > 1.0e-8 ? ctx->guard_band : 0.0);   // (the debug switch widens both kinds of band)
    }
    sa.grid = ctx->grid_dev;
    sa.lds_table_bytes = ctx->lds_table_bytes;
    sa.undefined_edge = (ctx->undefined_policy & BL_UNDEFINED_EDGE) ? 1 : 0;
    // Polarized runs in the tolerant tier keep the exact tier's per-frequency coefficient kernel: the reference's polarized step
    // amplifies last-place differences of the coefficients by up to ten orders of magnitude in optically and Faraday thick
    // configurations (docs/notebook.md section 5h), so only bit-identical coefficients keep Stokes V within the tier's tolerance
    // everywhere. The tolerant coefficient kernel (106 -> 59 ms per 1024^2 frame) is there for the asking.
    sa.tolerant = job.fast ? 1 : 0;
  } else {
    BlFormulaDevice &fm = sa.formula;
    fm.r0 = p.formula_r0; fm.h = p.formula_h; fm.l0 = p.formula_l0; fm.q = p.formula_q; fm.nup = p.formula_nup;
    fm.cn0 = p.formula_cn0; fm.alpha = p.formula_alpha; fm.a = p.formula_a; fm.beta = p.formula_beta;
  }
  sa.samples_renormalised = job.geo_load ? 1 : 0;
  sa.general_locate = (ctx->switches & BL_SWITCH_GENERAL_LOCATE) ? 1 : 0;
  // (uploaded when it differs from what the device holds: a frame loop uploads it once and waits for nothing here)
  if (ctx->shade_cold_host.size() != sizeof(BlShadeCold) || std::memcmp(ctx->shade_cold_host.data(), &cold, sizeof(BlShadeCold)) != 0) {
    ctx->d_shade_cold.Ensure(1);
    ctx->shade_cold_host.clear();   // (what the device holds is unknown until the copy has completed)
    Check(hipMemcpyAsync(ctx->d_shade_cold.ptr, &cold, sizeof(BlShadeCold), hipMemcpyHostToDevice, stream), "shade parameter upload");
    Check(hipStreamSynchronize(stream), "shade parameter upload");
    ctx->shade_cold_host.assign(reinterpret_cast<const unsigned char *>(&cold), reinterpret_cast<const unsigned char *>(&cold) + sizeof(BlShadeCold));
  }
  sa.cold = ctx->d_shade_cold.ptr;
  sa.frequencies = ctx->d_freq.ptr;
  sa.n_nu = job.n_nu;
  sa.ray_max_steps = job.max_steps;
  sa.x_unit = kGGMsun * ctx->frame.mass_msun / (kC * kC);   // unpolarized.cpp:42
  sa.aux_need_coefficients = (p.image_light || p.image_emission || p.image_tau || ctx->aux_images.image_emission_ave
                              || ctx->aux_images.image_tau_int) ? 1 : 0;   // simulation_coefficients.cpp:389
  sa.aux_need_length = (ctx->aux_images.image_length || job.fill_present) ? 1 : 0;
  sa.aux_record_unused = job.rows_only ? 1 : 0;
  for (int mu = 0; mu < 4; mu++) sa.cam_x[mu] = ctx->frame.cam_x[mu];
  sa.tag_in_record = job.fast ? 1 : 0;
  sa.freq_split = job.freq_split ? 1 : 0;
  sa.coef_split = job.coef_split ? 1 : 0;
  sa.redo_capacity = (job.fast || job.fast_formula || ctx->polarized) ? job.redo_capacity : 0;

  job.snapshot_time = job.slow ? p.slow_t_start + p.slow_dt * ctx->snapshot : 0.0;   // simulation_reader.cpp:214
  if (job.slow) {
    const int chunk_size = p.slow_chunk_size;
    std::vector<unsigned long long> table(3 * static_cast<size_t>(chunk_size) + 4, 0ull);
    for (int n = 0; n < chunk_size; n++) {
      const bl_ctx::SlowSlice &slice = ctx->slow_slices[n];
      table[n] = reinterpret_cast<unsigned long long>(slice.cells.ptr);
      table[chunk_size + n] = reinterpret_cast<unsigned long long>(slice.kappa.ptr);
      std::memcpy(&table[2 * static_cast<size_t>(chunk_size) + n], &slice.time, sizeof(double));
    }
    ctx->d_slow_table.Ensure(table.size());
    Check(hipMemcpy(ctx->d_slow_table.ptr, table.data(), table.size() * sizeof(unsigned long long), hipMemcpyHostToDevice), "slow-light table upload");
    ctx->d_ray_extrap.Ensure(job.n_rays);
    Check(hipMemsetAsync(ctx->d_ray_extrap.ptr, 0, job.n_rays * sizeof(unsigned int), stream), "slow-light flags reset");
    sa.slow.n = chunk_size;
    sa.slow.interp = p.slow_interp ? 1 : 0;
    sa.slow.snapshot_time = job.snapshot_time;
    sa.slow.cells = reinterpret_cast<const float *const *>(ctx->d_slow_table.ptr);
    sa.slow.kappa = reinterpret_cast<const float *const *>(ctx->d_slow_table.ptr + chunk_size);
    sa.slow.times = reinterpret_cast<const double *>(ctx->d_slow_table.ptr + 2 * static_cast<size_t>(chunk_size));
    sa.slow.extrap_max = ctx->d_slow_table.ptr + 3 * static_cast<size_t>(chunk_size);
  }
  if (ctx->polarized) {
    for (int c = 0; c < 7; c++) sa.power_pol[c] = ctx->power_pol[c];
    sa.plasma_gamma_min = p.plasma_power_frac != 0.0 ? p.plasma_gamma_min : 0.0;
  }
  // Locate kernel: 256-thread workgroups. Alone it runs 4 waves per SIMD; beside the next chunk's geodesic kernel
  // (bl_set_overlap) one workgroup per CU, so that whichever of the two kernels is dispatched first cannot fill the
  // register file and lock the other out.
  job.locate_grid_alone = ctx->num_cus * 4 * 4;
  job.locate_grid_shared = ctx->num_cus;
  job.shade_grid = ctx->num_cus * 2 * 4;  // 256-thread workgroups, 2 waves per SIMD, x4 for tail balance
}

// ---- the transfer kernels'
void BuildTransferArgs(RenderJob &job) {
  bl_ctx *ctx = job.ctx;
  const bl_params &p = ctx->params;
  BlTransferArgs &xa = job.xa;
  xa.frequencies = ctx->d_freq.ptr;
  xa.n_nu = job.n_nu;
  xa.ray_max_steps = job.max_steps;
  xa.fallback_nan = p.fallback_nan;
  xa.model_type = p.model_type;
  xa.affine = (job.fast || job.fast_formula) ? 1 : 0;
  xa.lane_transfer = (ctx->switches & BL_SWITCH_LANE_TRANSFER) ? 1 : 0;
  xa.n_rays_total = job.n_rays;
  xa.image = job.image;
  xa.out_sample_num = job.out_num;
  xa.out_flags = job.out_flags;
  xa.aux_images = ctx->aux_images;
  xa.aux_images.polarized_rows_only = job.rows_only ? 1 : 0;
  xa.x_unit = kGGMsun * ctx->frame.mass_msun / (kC * kC);
  xa.t_unit = xa.x_unit / kC;   // unpolarized.cpp:43
  xa.render_params = ctx->render_num_images > 0 ? ctx->d_render_params.ptr : nullptr;
  xa.render = job.render_out;
  if (ctx->polarized) {
    xa.camera_pos = job.cam_pos;
    xa.camera_dir = job.cam_dir;
    xa.st = ctx->st;
    xa.simulation_coord = p.simulation_coord == BL_COORD_FMKS ? BL_COORD_SKS : p.simulation_coord;
    xa.rotation_split = p.image_rotation_split ? 1 : 0;
    for (int mu = 0; mu < 4; mu++) {
      xa.cam_u_con[mu] = ctx->frame.u_con[mu];
      xa.cam_u_cov[mu] = ctx->frame.u_cov[mu];
      xa.cam_vert_con_c[mu] = ctx->frame.vert_con_c[mu];
    }
  }
}

// Point the three argument blocks at scratch set k and at the rays [begin, begin + rays)
void BindChunk(RenderJob &job, int k, long long begin, int rays) {
  bl_ctx *ctx = job.ctx;
  bl_ctx::ChunkSlot &sl = ctx->slot[k];
  BlTraceArgs &ta = job.ta;
  BlShadeArgs &sa = job.sa;
  BlTransferArgs &xa = job.xa;
  ta.chunk_begin = begin;
  ta.chunk_rays = rays;
  // The halves of a sample record (position + id | momentum + length): side by side in one array where every reader wants both
  // (the fused tolerant kernel: 64 contiguous bytes per lane for the geodesic kernel's scattered stores), in two arrays where
  // the locate kernel reads positions only
  ta.records_hot = sl.d_records_hot.ptr;
  ta.records_cold = job.interleaved ? reinterpret_cast<BlSampleCold *>(sl.d_records_hot.ptr + 1) : sl.d_records_cold.ptr;
  ta.record_stride = job.interleaved ? 2 : 1;
  ta.sample_t = job.need_time ? sl.d_sample_t.ptr : nullptr;
  ta.counters = sl.d_counters.ptr;
  ta.ray_kt = ctx->d_ray_kt.ptr + begin;
  ta.ray_factor = ctx->d_ray_factor.ptr + begin;
  ta.ray_sample_num = ctx->d_ray_sample_num.ptr + begin;
  ta.ray_skipped = job.skip_shell ? ctx->d_ray_skipped.ptr + begin : nullptr;
  ta.segment_rows = job.composed ? 1 : 0;
  ta.ray_rows = job.composed ? ctx->d_ray_rows.ptr + begin : nullptr;
  ta.parked = (job.park || job.split_long) ? sl.d_parked.ptr : nullptr;
  ta.split_b_lo = ta.split_b_hi = 0.0;
  if (job.split_long) {
    ta.split_b_lo = job.split_b_lo;
    ta.split_b_hi = job.split_b_hi;
  }
  ta.park_capacity = static_cast<int>(std::min<size_t>(job.park_capacity, 0x7fffffff));
  ta.park_below = 0;           // (a wave parks its rays once the queue is dry and none of its lanes holds ... see bl_geodesic.hip; the
  ta.park_after = 0;           //  other thresholds were measurement knobs of rounds 4 and 5, left at the values that won)
  ta.park_quiet = 1 << 30;
  ta.park_age = job.max_steps / 8;
  ta.quad_first_round = ctx->num_cus * 4;
  ta.park_always = (job.park && (ctx->switches & BL_SWITCH_QUAD_EVERY_RAY)) ? 1 : 0;
  ta.ray_flags = ctx->d_ray_flags.ptr + begin;
  ta.ray_out_index = ctx->d_ray_out_index.ptr + begin;
  ta.ray_offset = ctx->d_ray_offset.ptr + begin;
  ta.ray_start = job.geo_load ? nullptr : ctx->d_ray_start.ptr + begin;
  sa.records_hot = ta.records_hot;
  sa.records_cold = ta.records_cold;
  sa.record_stride = ta.record_stride;
  sa.located = (job.simulation && !job.locate_inside) ? sl.d_located.ptr : nullptr;
  sa.located_tag = (job.simulation && !job.locate_inside) ? sl.d_located_tag.ptr : nullptr;
  sa.freq_inputs = job.freq_split ? sl.d_freq_inputs.ptr : nullptr;
  sa.counters_in = sl.d_counters.ptr;
  sa.counters = sl.d_counters.ptr;
  sa.ray_kt = ta.ray_kt;
  sa.ray_factor = ta.ray_factor;
  sa.ray_offset = ta.ray_offset;
  sa.ray_flags = ta.ray_flags;
  sa.transfer = ctx->polarized ? nullptr : sl.d_transfer.ptr;
  sa.composed = job.composed ? sl.d_composed.ptr : nullptr;
  sa.tau_inc = job.tau_row ? sl.d_tau_inc.ptr : nullptr;
  sa.aux = (job.aux && !job.rows_only) ? sl.d_aux.ptr : nullptr;
  sa.sample_t = ta.sample_t;
  sa.coef_inputs = (job.coef_split || ctx->polarized) ? sl.d_coef_inputs.ptr : nullptr;
  sa.have_flags = job.pol_coefficients_inside ? sl.d_have_flags.ptr : nullptr;
  sa.anchors = (job.block_interp && !job.locate_inside) ? sl.d_anchors.ptr : nullptr;
  sa.redo_list = (job.fast || job.fast_formula || ctx->polarized) ? sl.d_redo.ptr : nullptr;
  if (job.slow) {
    sa.slow.frac = sl.d_slow_frac.ptr;
    sa.slow.ray_extrap = ctx->d_ray_extrap.ptr + begin;
  }
  xa.chunk_rays = rays;
  xa.counters = sl.d_counters.ptr;
  xa.transfer = ctx->polarized ? sl.d_pol_coeffs.ptr : sl.d_transfer.ptr;
  xa.ja_stride = ctx->polarized ? 4 : 1;
  xa.composed = sa.composed;
  xa.ray_rows = ta.ray_rows;
  xa.tau_inc = sa.tau_inc;
  xa.tau_row = ctx->aux_images.offset_tau;
  xa.freq_inputs = sa.freq_inputs;
  xa.ray_sample_num = ta.ray_sample_num;
  xa.ray_skipped = ta.ray_skipped;
  xa.ray_flags = ta.ray_flags;
  xa.ray_out_index = ta.ray_out_index;
  xa.ray_offset = ta.ray_offset;
  xa.ray_factor = ta.ray_factor;
  xa.stats = sl.d_counters.ptr + BL_CNT_COUNT;
  xa.aux = sa.aux;
  if (ctx->polarized) {
    sa.pol_samples = sl.d_pol_samples.ptr;
    sa.pol_coeffs = sl.d_pol_coeffs.ptr;
    xa.pol_samples = sl.d_pol_samples.ptr;
    xa.pol_coeffs = sl.d_pol_coeffs.ptr;
    xa.pol_matrix = job.matrix_transport ? sl.d_pol_matrix.ptr : nullptr;
  }
}

// LoadGeodesics(): the chunk's sample records come from the file instead of the geodesic kernel. The file holds them
// far -> near (ReverseGeodesics) with the renormalised momentum; records are near -> far, so sample n of a ray is entry
// num - 1 - n, and len = -sample_len. Takes as many of the rays [begin, begin + rays) as the record buffers hold and
// returns how many.
long long LoadChunkFromCheckpoint(RenderJob &job, int k, long long begin, int rays) {
  bl_ctx *ctx = job.ctx;
  const bl_render_desc *d = job.d;
  bl_ctx::ChunkSlot &sl = ctx->slot[k];
  const bl_ctx::Checkpoint &ck = *ctx->checkpoint;
  const size_t steps = static_cast<size_t>(ck.num_steps);
  size_t total = 0;
  int taken = 0;
  for (; taken < rays; taken++) {
    const long long ray = begin + taken;
    const size_t m = d->pixel_map != nullptr ? static_cast<size_t>(d->pixel_map[ray]) : static_cast<size_t>(ray);
    const size_t num = static_cast<size_t>(ck.sample_num[m]);
    if (total + num > static_cast<size_t>(job.record_gate)) break;
    total += num;
  }
  if (taken == 0) throw Failure{BL_E_ARG, "Scratch budget too small for the samples of one checkpointed ray (bl_set_scratch_limit)."};
  std::vector<BlSampleHot> hot;
  std::vector<BlSampleCold> cold;
  std::vector<double> sample_t, ray_kt(taken), ray_factor(taken);
  std::vector<int> ray_num(taken);
  std::vector<unsigned char> ray_flags(taken);
  std::vector<long long> ray_out(taken), ray_offset(taken);
  hot.reserve(total);
  cold.reserve(total);
  sample_t.reserve(total);
  for (int q = 0; q < taken; q++) {
    const long long ray = begin + q;
    const size_t m = d->pixel_map != nullptr ? static_cast<size_t>(d->pixel_map[ray]) : static_cast<size_t>(ray);
    const int num = ck.sample_num[m];
    ray_kt[q] = ck.camera_dir[4 * m];
    ray_factor[q] = ck.factors[m];
    ray_num[q] = num;
    ray_flags[q] = ck.flags[m];
    ray_out[q] = ray;
    ray_offset[q] = static_cast<long long>(hot.size());
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
  const unsigned long long n_loaded = hot.size(), n_taken = static_cast<unsigned long long>(taken);
  if (n_loaded > 0) {
    Check(hipMemcpy(sl.d_records_hot.ptr, hot.data(), n_loaded * sizeof(BlSampleHot), hipMemcpyHostToDevice), "checkpoint upload");
    Check(hipMemcpy(sl.d_records_cold.ptr, cold.data(), n_loaded * sizeof(BlSampleCold), hipMemcpyHostToDevice), "checkpoint upload");
    Check(hipMemcpy(sl.d_sample_t.ptr, sample_t.data(), n_loaded * sizeof(double), hipMemcpyHostToDevice), "checkpoint upload");
  }
  Check(hipMemcpy(ctx->d_ray_kt.ptr + begin, ray_kt.data(), taken * sizeof(double), hipMemcpyHostToDevice), "checkpoint upload");
  Check(hipMemcpy(ctx->d_ray_factor.ptr + begin, ray_factor.data(), taken * sizeof(double), hipMemcpyHostToDevice), "checkpoint upload");
  Check(hipMemcpy(ctx->d_ray_sample_num.ptr + begin, ray_num.data(), taken * sizeof(int), hipMemcpyHostToDevice), "checkpoint upload");
  Check(hipMemcpy(ctx->d_ray_flags.ptr + begin, ray_flags.data(), taken, hipMemcpyHostToDevice), "checkpoint upload");
  Check(hipMemcpy(ctx->d_ray_out_index.ptr + begin, ray_out.data(), taken * sizeof(long long), hipMemcpyHostToDevice), "checkpoint upload");
  Check(hipMemcpy(ctx->d_ray_offset.ptr + begin, ray_offset.data(), taken * sizeof(long long), hipMemcpyHostToDevice), "checkpoint upload");
  Check(hipMemcpy(sl.d_counters.ptr + BL_CNT_RECORDS, &n_loaded, sizeof n_loaded, hipMemcpyHostToDevice), "checkpoint upload");
  Check(hipMemcpy(sl.d_counters.ptr + BL_CNT_NEXT_RAY, &n_taken, sizeof n_taken, hipMemcpyHostToDevice), "checkpoint upload");
  return taken;
}

// SaveGeodesics(), first half: bring a chunk's records back while its scratch set still holds them
void SaveChunkRecords(RenderJob &job, int k, long long begin, int rays) {
  bl_ctx *ctx = job.ctx;
  bl_ctx::ChunkSlot &sl = ctx->slot[k];
  CheckpointSave &save = job.save;
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
  Check(hipMemcpy(ray_kt.data(), ctx->d_ray_kt.ptr + begin, rays * sizeof(double), hipMemcpyDeviceToHost), "checkpoint download");
  Check(hipMemcpy(ray_factor.data(), ctx->d_ray_factor.ptr + begin, rays * sizeof(double), hipMemcpyDeviceToHost), "checkpoint download");
  Check(hipMemcpy(ray_num.data(), ctx->d_ray_sample_num.ptr + begin, rays * sizeof(int), hipMemcpyDeviceToHost), "checkpoint download");
  Check(hipMemcpy(ray_flags.data(), ctx->d_ray_flags.ptr + begin, rays, hipMemcpyDeviceToHost), "checkpoint download");
  Check(hipMemcpy(ray_out.data(), ctx->d_ray_out_index.ptr + begin, rays * sizeof(long long), hipMemcpyDeviceToHost), "checkpoint download");
  if (save.sample_num.empty()) {
    save.sample_num.assign(job.n_rays, 0);
    save.flags.assign(job.n_rays, 0);
    save.factors.assign(job.n_rays, 0.0);
    save.offset.assign(job.n_rays, 0);
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

// SaveGeodesics(), second half (geodesic_checkpoint.cpp:28-59)
void WriteGeodesicCheckpoint(RenderJob &job) {
  bl_ctx *ctx = job.ctx;
  const bl_params &p = ctx->params;
  const CheckpointSave &save = job.save;
  const long long n_rays = job.n_rays;
  std::vector<double> camera_pos(static_cast<size_t>(n_rays) * 4), camera_dir(static_cast<size_t>(n_rays) * 4);
  Check(hipMemcpy(camera_pos.data(), job.cam_pos, camera_pos.size() * sizeof(double), hipMemcpyDeviceToHost), "checkpoint download");
  Check(hipMemcpy(camera_dir.data(), job.cam_dir, camera_dir.size() * sizeof(double), hipMemcpyDeviceToHost), "checkpoint download");
  std::ofstream out(p.checkpoint_geodesic_file.s, std::ios_base::out | std::ios_base::binary);
  if (!out.is_open()) throw Failure{BL_E_INPUT, "Could not open geodesic checkpoint file."};
  const bl_camera_frame &f = ctx->frame;
  const double *vectors[7] = {f.cam_x, f.u_con, f.u_cov, f.norm_con, f.norm_con_c, f.hor_con_c, f.vert_con_c};
  for (const double *v : vectors) out.write(reinterpret_cast<const char *>(v), 4 * sizeof(double));
  const int n_pix = static_cast<int>(n_rays);
  const int n_nu = job.n_nu;
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

// SaveSampling(), first half: a chunk's located samples, while its scratch set still holds them, as the reference keeps them -
// sample_inds (MeshBlock, k, j, i of the nearest cell or of the lower corner; eight of them with inter-block interpolation),
// sample_fracs (f_k, f_j, f_i), sample_nan, sample_fallback (simulation_sampling.cpp:205-216, :377-384, :427-549) - by pixel
// and by the reversed sample index of ReverseGeodesics.
void SaveChunkSampling(RenderJob &job, int k, long long begin, int rays) {
  bl_ctx *ctx = job.ctx;
  const bl_params &p = ctx->params;
  bl_ctx::ChunkSlot &sl = ctx->slot[k];
  RenderJob::SampleSave &out = job.sampling;
  const BlGridDevice &g = ctx->grid_dev;
  unsigned long long n_written = 0;
  Check(hipMemcpy(&n_written, sl.d_counters.ptr + BL_CNT_RECORDS, sizeof n_written, hipMemcpyDeviceToHost), "checkpoint download");
  const size_t stride = static_cast<size_t>(job.interleaved ? 2 : 1);
  std::vector<BlSampleHot> hot(n_written * stride);
  std::vector<BlLocated> located(n_written);
  std::vector<unsigned long long> tags(n_written);
  std::vector<unsigned int> anchors(job.block_interp ? n_written * 8 : 0);
  std::vector<int> ray_num(rays);
  std::vector<unsigned char> ray_flags(rays);
  std::vector<long long> ray_out(rays);
  if (n_written > 0) {
    Check(hipMemcpy(hot.data(), sl.d_records_hot.ptr, hot.size() * sizeof(BlSampleHot), hipMemcpyDeviceToHost), "checkpoint download");
    Check(hipMemcpy(located.data(), sl.d_located.ptr, n_written * sizeof(BlLocated), hipMemcpyDeviceToHost), "checkpoint download");
    if (!job.fast) Check(hipMemcpy(tags.data(), sl.d_located_tag.ptr, n_written * sizeof(unsigned long long), hipMemcpyDeviceToHost), "checkpoint download");
    if (job.block_interp) Check(hipMemcpy(anchors.data(), sl.d_anchors.ptr, anchors.size() * sizeof(unsigned int), hipMemcpyDeviceToHost), "checkpoint download");
  }
  Check(hipMemcpy(ray_num.data(), ctx->d_ray_sample_num.ptr + begin, rays * sizeof(int), hipMemcpyDeviceToHost), "checkpoint download");
  Check(hipMemcpy(ray_flags.data(), ctx->d_ray_flags.ptr + begin, rays, hipMemcpyDeviceToHost), "checkpoint download");
  Check(hipMemcpy(ray_out.data(), ctx->d_ray_out_index.ptr + begin, rays * sizeof(long long), hipMemcpyDeviceToHost), "checkpoint download");
  if (out.sample_num.empty()) {
    out.per_sample = job.block_interp ? 32 : 4;
    out.sample_num.assign(job.n_rays, 0);
    out.offset.assign(job.n_rays, 0);
  }
  std::vector<size_t> slot_offset(rays);
  for (int q = 0; q < rays; q++) {
    const size_t m = static_cast<size_t>(ray_out[q]);
    const size_t num = static_cast<size_t>(ray_num[q]);
    out.sample_num[m] = ray_num[q];
    out.offset[m] = slot_offset[q] = out.nan.size();
    out.inds.resize(out.inds.size() + num * out.per_sample, 0);
    if (p.simulation_interp) out.fracs.resize(out.fracs.size() + num * 3, 0.0);
    out.nan.resize(out.nan.size() + num, 0);
    out.fallback.resize(out.fallback.size() + num, 0);
  }
  // a cell of the HBM arrays as the reference names it: MeshBlock of the file, then k, j, i inside the block
  const size_t block_cells = static_cast<size_t>(g.nb[0]) * g.nb[1] * g.nb[2];
  auto name_cell = [&](unsigned int cell, int32_t *dst) {
    if (g.n_blocks > 0) {   // cells kept by MeshBlock
      const size_t b = cell / block_cells, rest = cell % block_cells;
      dst[0] = static_cast<int32_t>(b);
      dst[1] = static_cast<int32_t>(rest / g.stride_plane);
      dst[2] = static_cast<int32_t>(rest % g.stride_plane / g.stride_row);
      dst[3] = static_cast<int32_t>(rest % g.stride_row);
    } else {                // equal blocks merged into one array (one block: itself)
      const int i = static_cast<int>(cell % g.n[0]), j = static_cast<int>(cell / g.n[0] % g.n[1]), kk = static_cast<int>(cell / (static_cast<size_t>(g.n[0]) * g.n[1]));
      const int at = ((kk / g.nb[2]) * ctx->merged_blocks[1] + j / g.nb[1]) * ctx->merged_blocks[0] + i / g.nb[0];
      dst[0] = ctx->merged_block_at.empty() ? 0 : ctx->merged_block_at[at];
      dst[1] = kk % g.nb[2];
      dst[2] = j % g.nb[1];
      dst[3] = i % g.nb[0];
    }
  };
  for (unsigned long long r = 0; r < n_written; r++) {
    const BlSampleHot &h = hot[r * stride];
    if (h.ray == BL_DEAD_RAY) continue;
    const int num = ray_num[h.ray];
    if (static_cast<int>(h.n) >= num) continue;
    const size_t at = slot_offset[h.ray] + static_cast<size_t>(num - 1 - static_cast<int>(h.n));
    if (p.fallback_nan && ray_flags[h.ray] != 0) {   // a poorly terminated geodesic samples NaN everywhere (:211-216)
      out.nan[at] = 1;
      continue;
    }
    unsigned long long tag = tags[r];
    if (job.fast) std::memcpy(&tag, &located[r].ph, sizeof tag);   // tolerant tier: the tag rides in the azimuth's slot
    const int status = static_cast<int>(tag >> 32) & 0xff;
    if (status == 2) {                 // off the grid (:377-384)
      (p.fallback_nan ? out.nan : out.fallback)[at] = 1;
    } else if (status == 3 || status == 4) {   // nearest cell | lower corner of the trilinear stencil
      name_cell(static_cast<unsigned int>(tag), &out.inds[at * out.per_sample]);
    } else if (status == 6) {          // inter-block interpolation: the eight anchors (:541)
      for (int c = 0; c < 8; c++) name_cell(anchors[r * 8 + c], &out.inds[at * out.per_sample + 4 * c]);
    }
    if (p.simulation_interp && (status == 4 || status == 6)) {
      out.fracs[3 * at] = located[r].f_k;
      out.fracs[3 * at + 1] = located[r].f_j;
      out.fracs[3 * at + 2] = located[r].f_i;
    }
  }
}

// SaveSampling(), second half (sample_checkpoint.cpp:22-46): four Arrays (file_io.cpp:65-76: five int32 extents, fastest first,
// then the data). Entries the reference never writes - beyond a pixel's samples, cut samples, samples off the grid - are
// whatever its allocator held there; zeros here.
void WriteSampleCheckpoint(RenderJob &job) {
  bl_ctx *ctx = job.ctx;
  const bl_params &p = ctx->params;
  const RenderJob::SampleSave &sv = job.sampling;
  std::ofstream out(p.checkpoint_sample_file.s, std::ios_base::out | std::ios_base::binary);
  if (!out.is_open()) throw Failure{BL_E_INPUT, "Could not open sample checkpoint file."};
  const int n_pix = static_cast<int>(job.n_rays);
  int num_steps = 0;
  for (int32_t num : sv.sample_num) num_steps = std::max(num_steps, static_cast<int>(num));
  auto write_rows = [&](const auto &packed, int per_sample) {
    using T = typename std::decay<decltype(packed)>::type::value_type;
    std::vector<T> row(static_cast<size_t>(num_steps) * per_sample);
    for (int m = 0; m < n_pix; m++) {
      std::fill(row.begin(), row.end(), T(0));
      const size_t first = sv.offset[m] * per_sample, count = static_cast<size_t>(sv.sample_num[m]) * per_sample;
      std::copy(packed.begin() + first, packed.begin() + first + count, row.begin());
      out.write(reinterpret_cast<const char *>(row.data()), static_cast<std::streamsize>(row.size() * sizeof(T)));
    }
  };
  if (job.block_interp) {
    const int dims[5] = {4, 8, num_steps, n_pix, 1};
    out.write(reinterpret_cast<const char *>(dims), sizeof dims);
  } else {
    WriteCheckpointHeader<int32_t>(out, 4, num_steps, n_pix);
  }
  write_rows(sv.inds, sv.per_sample);
  if (p.simulation_interp) {
    WriteCheckpointHeader<double>(out, 3, num_steps, n_pix);
    write_rows(sv.fracs, 3);
  }
  WriteCheckpointHeader<uint8_t>(out, num_steps, n_pix, 1);
  write_rows(sv.nan, 1);
  WriteCheckpointHeader<uint8_t>(out, num_steps, n_pix, 1);
  write_rows(sv.fallback, 1);
  if (!out) throw Failure{BL_E_INPUT, "Could not write sample checkpoint file."};
  ctx->sample_checkpoint_saved = true;
}

// ---- a chunk, first half: the geodesic stage on stream_geo into scratch set k (the set must be free)
void LaunchGeodesicStage(RenderJob &job, int k, long long begin, int rays, hipStream_t stream_geo) {
  bl_ctx *ctx = job.ctx;
  bl_ctx::ChunkSlot &sl = ctx->slot[k];
  hipEvent_t *e = SlotEvents(job, k);
  BindChunk(job, k, begin, rays);
  Check(hipMemsetAsync(sl.d_counters.ptr, 0, BL_CNT_TOTAL * sizeof(unsigned long long), stream_geo), "counter reset");
  Check(hipEventRecord(e[0], stream_geo), "event");
  job.in_flight[k].busy = true;
  job.in_flight[k].begin = begin;
  job.in_flight[k].rays = rays;
  job.in_flight[k].done = -1;
  if (job.geo_load) {
    Check(hipStreamSynchronize(stream_geo), "kernel execution");   // the counters are reset before the host writes two of them
    job.in_flight[k].done = LoadChunkFromCheckpoint(job, k, begin, rays);
  } else {
    if (job.split_long) {
      // the band's rays parked; then the two steppers side by side on disjoint compute units, and the stage ends with both
      hipEvent_t *ev = SlotEvents(job, k);
      Check(bl_launch_split_long(&job.ta, stream_geo), "split kernel launch");
      Check(hipEventRecord(ev[7], stream_geo), "event");
      Check(hipStreamWaitEvent(ctx->stream_most, ev[7], 0), "stream wait");
      Check(hipStreamWaitEvent(ctx->stream_few, ev[7], 0), "stream wait");
      // (as many waves per SIMD as the unsplit launch would have had: a wave that shares its SIMD steps its rays more slowly)
      const int per_simd = std::max(1, (std::min(job.geo_grid, (rays + 63) / 64) + ctx->num_cus * 4 - 1) / (ctx->num_cus * 4));
      const int wide_grid = std::min(std::min(job.geo_grid, (rays + 63) / 64), (ctx->num_cus - job.split_cus) * 4 * per_simd);
      const int pad = per_simd == 1 ? ctx->split_lds_pad : 0;
      Check(bl_launch_geodesic(&job.ta, ctx->params.ray_integrator, std::max(wide_grid, 1), ctx->stream_most, pad), "geodesic kernel launch");
      Check(bl_launch_geodesic_quad(&job.ta, job.quad_grid, ctx->stream_few, ctx->split_lds_pad), "geodesic quad kernel launch");
      Check(hipEventRecord(ev[8], ctx->stream_most), "event");
      Check(hipEventRecord(ev[9], ctx->stream_few), "event");
      Check(hipStreamWaitEvent(stream_geo, ev[8], 0), "stream wait");
      Check(hipStreamWaitEvent(stream_geo, ev[9], 0), "stream wait");
    } else
    Check(bl_launch_geodesic(&job.ta, ctx->params.ray_integrator, std::min(job.geo_grid, (rays + 63) / 64), stream_geo, 0), "geodesic kernel launch");
    // the rays it parked, sixteen to a wave, a wave per SIMD (waves that find none end at once)
    if (job.park) Check(bl_launch_geodesic_quad(&job.ta, job.quad_grid, stream_geo, 0), "geodesic quad kernel launch");
  }
  Check(hipEventRecord(e[1], stream_geo), "event");
}

// ---- second half: locate, coefficients, transfer on the shading stream, counters to the host
void LaunchShadingStage(RenderJob &job, int k, bool geodesic_beside, hipStream_t stream) {
  bl_ctx *ctx = job.ctx;
  const bl_params &p = ctx->params;
  bl_ctx::ChunkSlot &sl = ctx->slot[k];
  hipEvent_t *e = SlotEvents(job, k);
  BindChunk(job, k, job.in_flight[k].begin, job.in_flight[k].rays);
  BlShadeArgs &sa = job.sa;
  BlTransferArgs &xa = job.xa;
  auto coefficient_kernel = [&]() {
    if (job.fast) Check(bl_launch_shade_fast(&sa, job.shade_grid, stream), "coefficient kernel launch");
    else if (job.fast_formula) Check(bl_launch_shade_formula_fast(&sa, ctx->num_cus * 4 * 4, stream), "coefficient kernel launch");
    else if (job.exact_fused) Check(bl_launch_shade_exact2(&sa, job.shade_grid, stream), "coefficient kernel launch");
    else if (job.pol_fused) Check(bl_launch_shade_polarized2(&sa, job.shade_grid, stream), "coefficient kernel launch");
    else Check(bl_launch_shade(&sa, p.model_type, job.shade_grid, stream), "coefficient kernel launch");
  };
  Check(hipStreamWaitEvent(stream, e[1], 0), "stream wait");
  Check(hipEventRecord(e[2], stream), "event");
  if (job.simulation && !job.locate_inside && !job.reuse_located)
    Check(bl_launch_locate(&sa, geodesic_beside ? job.locate_grid_shared : job.locate_grid_alone, ctx->lds_table_bytes, stream), "locate kernel launch");
  Check(hipEventRecord(e[3], stream), "event");
  coefficient_kernel();
  // The transport matrices - memory - on the second stream beside the per-frequency coefficient kernel - arithmetic: both read what
  // bl_shade_polarized2_kernel left, neither reads the other. (The coefficient kernel's workgroups fill the device first, so the
  // matrices overlap its last quarter only: 276 -> 270 ms per 1024^2 frame, 1.10 -> 1.08 s at 2048^2 adaptive; a smaller grid for the
  // coefficient kernel or a priority stream for the matrices move the split, not the sum. BLACKLIGHT_AMD_POLARIZED_OVERLAP=0: in sequence.)
  // (one scratch set: with two, the second stream carries the next chunk's geodesic stage, and the matrices would queue behind it)
  // (with the coefficients evaluated inside the coefficient kernel there is nothing left beside which to build them: in sequence)
  const bool matrices_beside = ctx->polarized && job.matrix_transport && job.n_slots == 1 && ctx->stream_geo != stream && !job.pol_coefficients_inside;
  const int polcoef_grid = ctx->num_cus * 20;
  if (matrices_beside) {
    // (the frames of the samples without coefficients first: the matrices read them)
    Check(bl_launch_polarized_coefficients_parts(&sa, polcoef_grid, 2, stream), "polarized frame kernel launch");
    Check(hipEventRecord(e[10], stream), "event");
    Check(hipStreamWaitEvent(ctx->stream_geo, e[10], 0), "stream wait");
    Check(bl_launch_transport_matrices(&xa, ctx->num_cus, ctx->stream_geo), "transport matrix kernel launch");
    Check(hipEventRecord(e[11], ctx->stream_geo), "event");
  }
  if (ctx->polarized) Check(bl_launch_polarized_coefficients_parts(&sa, polcoef_grid, job.pol_coefficients_inside ? 2 : (matrices_beside ? 0 : 1), stream), "polarized coefficient kernel launch");
  if (job.coef_split) Check(bl_launch_coefficients_freq(&sa, ctx->num_cus * 16, stream), "per-frequency coefficient kernel launch");
  Check(hipEventRecord(e[4], stream), "event");
  Check(job.aux ? bl_launch_transfer_aux(&xa, stream)
                : (job.freq_split ? bl_launch_transfer_freq(&xa, stream) : (job.composed ? bl_launch_transfer_composed(&xa, stream) : bl_launch_transfer(&xa, stream))),
        "transfer kernel launch");
  if (job.tau_row) Check(bl_launch_tau(&xa, stream), "optical-depth kernel launch");
  if (ctx->polarized && matrices_beside) {
    Check(hipStreamWaitEvent(stream, e[11], 0), "stream wait");
    Check(bl_launch_transfer_polarized_rays(&xa, stream), "polarized transfer kernel launch");
  } else if (ctx->polarized) {
    Check(job.matrix_transport ? bl_launch_transfer_polarized_matrix(&xa, ctx->num_cus, stream) : bl_launch_transfer_polarized(&xa, stream),
          "polarized transfer kernel launch");
  }
  Check(hipEventRecord(e[5], stream), "event");
  Check(hipMemcpyAsync(ctx->host_counters + static_cast<size_t>(k) * BL_CNT_TOTAL, sl.d_counters.ptr, BL_CNT_TOTAL * sizeof(unsigned long long),
                       hipMemcpyDeviceToHost, stream), "counter download");
  Check(hipEventRecord(e[6], stream), "event");
}

// Rays of the chunk on scratch set k that the geodesic stage covered (waits for that stage)
long long WaitGeodesicStage(RenderJob &job, int k, hipStream_t stream_geo) {
  RenderJob::InFlight &fl = job.in_flight[k];
  if (fl.done >= 0) return fl.done;
  Check(hipStreamSynchronize(stream_geo), "kernel execution");
  unsigned long long taken = 0;
  Check(hipMemcpy(&taken, job.ctx->slot[k].d_counters.ptr + BL_CNT_NEXT_RAY, sizeof taken, hipMemcpyDeviceToHost), "counter download");
  fl.done = static_cast<long long>(std::min<unsigned long long>(taken, static_cast<unsigned long long>(fl.rays)));
  return fl.done;
}

// ---- wait for the chunk on scratch set k, add its counters and times to the totals, free the set
void CollectChunk(RenderJob &job, int k) {
  bl_ctx *ctx = job.ctx;
  const bl_params &p = ctx->params;
  RenderJob::InFlight &fl = job.in_flight[k];
  if (!fl.busy) return;
  hipEvent_t *e = SlotEvents(job, k);
  Check(hipEventSynchronize(e[6]), "kernel execution");
  const unsigned long long *hc = ctx->host_counters + static_cast<size_t>(k) * BL_CNT_TOTAL;
  float ms = 0.0f;
  Check(hipEventElapsedTime(&ms, e[0], e[1]), "event time"); job.ms_geo += ms;
  Check(hipEventElapsedTime(&ms, e[2], e[3]), "event time"); job.ms_locate += ms;
  if (job.split_long && ctx->debug_counters) {
    float a = 0, b = 0, c = 0;
    (void)hipEventElapsedTime(&a, e[0], e[7]);
    (void)hipEventElapsedTime(&b, e[0], e[8]);
    (void)hipEventElapsedTime(&c, e[0], e[9]);
    std::fprintf(stderr, "split long: rays parked by %.3f ms, wide stepper done at %.3f ms, quad stepper at %.3f ms; %llu rays with b in [%.3f, %.3f] M on %d CUs\n", a, b, c, hc[BL_CNT_PARKED],
                 job.split_b_lo, job.split_b_hi, job.split_cus);
  }
  Check(hipEventElapsedTime(&ms, e[3], e[4]), "event time"); job.ms_shade += ms;
  Check(hipEventElapsedTime(&ms, e[4], e[5]), "event time"); job.ms_transfer += ms;
  fl.busy = false;
  if (fl.done < 0) fl.done = static_cast<long long>(std::min<unsigned long long>(hc[BL_CNT_NEXT_RAY], static_cast<unsigned long long>(fl.rays)));
  if (hc[BL_CNT_OVERFLOW] != 0) throw Failure{BL_E_DEVICE, "Sample record buffer overflow."};
  if (hc[BL_CNT_INTERP_FAILED] != 0) throw Failure{BL_E_INPUT, "Grid interpolation failed."};   // simulation_sampling.cpp:1319
  if (hc[BL_CNT_UNDEFINED] != 0 && !(ctx->undefined_policy & BL_UNDEFINED_EDGE)) {
    if (p.simulation_coord == BL_COORD_FMKS)
      throw Failure{BL_E_UNSUPPORTED, "FMKS sampling reached the last polar zone of the last azimuthal plane (or the last entry of the "
                                      "coordinate table), where the reference reads past its arrays (simulation_sampling.cpp:405-415, "
                                      ":809-819): no defined result to reproduce. bl_set_undefined_policy(BL_UNDEFINED_EDGE) uses the edge cell instead."};
    throw Failure{BL_E_UNSUPPORTED, "Inter-block interpolation reached an upper edge of the last MeshBlock, where the reference reads past the end "
                                    "of its cell-centre arrays (simulation_sampling.cpp:520-522): no defined result to reproduce. "
                                    "bl_set_undefined_policy(BL_UNDEFINED_EDGE) mirrors the last cell centre about the block's face instead."};
  }
  job.total_undefined += hc[BL_CNT_UNDEFINED];
  job.total_records += hc[BL_CNT_RECORDS];
  job.total_gathers += hc[BL_CNT_GATHERS];
  job.total_parked += std::min<unsigned long long>(hc[BL_CNT_PARKED] + hc[BL_CNT_PARKED_YOUNG], job.park_capacity);
  if (job.fast || job.fast_formula) job.total_redo += hc[BL_CNT_REDO];   // (polarized runs use the list for something else: bl_polarized_frame_kernel)
  job.total_samples += hc[BL_CNT_COUNT + 0];
  job.total_flagged += hc[BL_CNT_COUNT + 1];
  job.max_num = std::max<unsigned long long>(job.max_num, hc[BL_CNT_COUNT + 2]);
  for (int c = 0; c < 8; c++) job.debug_counters[c] += hc[BL_CNT_DEBUG + c];
  job.n_chunks++;
}

void DownloadChunk(RenderJob &job, long long begin, long long count);
hipError_t DownloadColumns(const RenderJob &job, long long begin, long long count, int threads);

// ---- all chunks of the call
void RunChunks(RenderJob &job) {
  bl_ctx *ctx = job.ctx;
  hipStream_t stream = ctx->stream;
  // one scratch set: one stream, chunks back to back
  hipStream_t stream_geo = job.n_slots == 2 ? ctx->stream_geo : stream;
  hipEvent_t ev_begin = ctx->events[2 * kEventsPerChunk], ev_end = ctx->events[2 * kEventsPerChunk + 1];
  Check(hipEventRecord(ev_begin, stream), "event");                // the uploads above were queued on `stream`
  if (stream_geo != stream) Check(hipStreamWaitEvent(stream_geo, ev_begin, 0), "stream wait");
  if (job.reuse) {
    // Shade the resident records again: the per-ray rows and the records are where the stepper left them; the counters go back to
    // what they were when it ended (and when the locate kernel ended, if its samples are kept too). The caller's camera_pos /
    // camera_dir - a pure function of the pixel - are written again by the kernel that wrote them then.
    bl_ctx::ChunkSlot &sl = ctx->slot[0];
    hipEvent_t *e = SlotEvents(job, 0);
    BindChunk(job, 0, 0, static_cast<int>(job.n_rays));
    if (job.cam_pos != nullptr || job.cam_dir != nullptr) Check(bl_launch_ray_init(&job.ta, ctx->params.ray_integrator, stream), "ray start kernel launch");
    unsigned long long *staged = ctx->host_counters + BL_CNT_TOTAL;   // (pinned; the second set's half: a reuse render has one set)
    std::memcpy(staged, ctx->resident.counters, BL_CNT_TOTAL * sizeof(unsigned long long));
    if (!job.reuse_located)
      for (int c : {BL_CNT_GATHERS, BL_CNT_UNDEFINED, BL_CNT_INTERP_FAILED}) staged[c] = 0ull;
    Check(hipMemcpyAsync(sl.d_counters.ptr, staged, BL_CNT_TOTAL * sizeof(unsigned long long), hipMemcpyHostToDevice, stream), "counter upload");
    Check(hipEventRecord(e[0], stream), "event");
    Check(hipEventRecord(e[1], stream), "event");
    job.in_flight[0].busy = true;
    job.in_flight[0].begin = 0;
    job.in_flight[0].rays = static_cast<int>(job.n_rays);
    job.in_flight[0].done = job.n_rays;
    LaunchShadingStage(job, 0, false, stream);
    if (job.sample_save) {
      Check(hipStreamSynchronize(stream), "kernel execution");
      SaveChunkSampling(job, 0, 0, static_cast<int>(job.n_rays));
    }
    CollectChunk(job, 0);
    Check(hipEventRecord(ev_end, stream), "event");
    Check(hipStreamSynchronize(stream), "kernel execution");
    return;
  }
  if (!job.geo_load) {
    // start states of every ray of the call, once (bl_ray_init_kernel; the geodesic kernel of each chunk reads its share)
    BindChunk(job, 0, 0, static_cast<int>(job.n_rays));
    Check(bl_launch_ray_init(&job.ta, ctx->params.ray_integrator, stream_geo), "ray start kernel launch");
  }
  const long long n_rays = job.n_rays;
  auto no_progress = []() {
    return Failure{BL_E_ARG, "Scratch budget too small: no ray fits the sample record buffers (bl_set_scratch_limit)."};
  };
  long long begin = 0;
  for (int c = 0; begin < n_rays; c++) {
    const int k = c % job.n_slots;
    const int rays = static_cast<int>(n_rays - begin);
    CollectChunk(job, k);   // the chunk that used this scratch set before (two chunks back when there are two sets)
    LaunchGeodesicStage(job, k, begin, rays, stream_geo);
    long long done;
    if (job.n_slots == 2) {
      // Two sets: chunk c + 1's geodesic kernel is to run beside chunk c's shading, so this chunk's extent is fetched as
      // soon as its geodesic kernel ends, and its shading goes to the other stream without waiting for anything else.
      done = WaitGeodesicStage(job, k, stream_geo);
      LaunchShadingStage(job, k, begin + done < n_rays, stream);
      if (job.geo_save || job.sample_save) Check(hipStreamSynchronize(stream), "kernel execution");
      if (job.geo_save) SaveChunkRecords(job, k, begin, static_cast<int>(done));
      if (job.sample_save) SaveChunkSampling(job, k, begin, static_cast<int>(done));
    } else {
      LaunchShadingStage(job, k, false, stream);
      if (job.geo_save || job.sample_save) Check(hipStreamSynchronize(stream), "kernel execution");
      if (job.geo_save) SaveChunkRecords(job, k, begin, static_cast<int>(WaitGeodesicStage(job, k, stream_geo)));
      if (job.sample_save) SaveChunkSampling(job, k, begin, static_cast<int>(WaitGeodesicStage(job, k, stream_geo)));
      CollectChunk(job, k);
      done = job.in_flight[k].done;
    }
    if (done <= 0) throw no_progress();
    // (the split is planned for calls one chunk is sure to take - PlanScratch - but the band's reservations are counted twice for a
    // moment, and a frame whose rays all use every step they may can see the gate close on that: the marked rays of a second chunk
    // would be lost. Rendered again with one stepper instead.)
    if (job.split_long && done < rays) throw SplitIncomplete{};
    if (job.raster && job.n_slots == 1 && (job.chunk_downloads || begin + done < n_rays) && job.download_status.size() < 4000) {
      job.chunk_downloads = true;
      DownloadChunk(job, begin, done);   // (the chunk is complete: CollectChunk has waited for its kernels)
    } else if (job.chunk_downloads) {
      Check(DownloadColumns(job, begin, done, 1), "download of a chunk's outputs");
    }
    begin += done;
  }
  const int oldest = job.n_chunks % job.n_slots;   // chunks are collected in order: the next one to collect sits on this set
  for (int k = 0; k < job.n_slots; k++) CollectChunk(job, (oldest + k) % job.n_slots);
  Check(hipEventRecord(ev_end, stream), "event");
  Check(hipStreamSynchronize(stream_geo), "kernel execution");
  Check(hipStreamSynchronize(stream), "kernel execution");
}

// ---- results to the caller's host memory
// Memory the runtime can copy into without staging (hipHostMalloc - bl_host_alloc - or hipHostRegister)
bool IsPinnedHost(const void *p) {
  hipPointerAttribute_t attr{};
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();   // (plain malloc memory is "invalid value" to the runtime)
    return false;
  }
  return attr.type == hipMemoryTypeHost;
}

// `rows` rows of `width` bytes each, device -> host, both sides with a pitch. Pinned destination: one copy at the link's rate. Pageable:
// the runtime stages such a copy through a buffer of its own at ~16 GB/s per calling thread (measured: 537 MB of image rows in
// 33.7 ms, round 5); `threads` host threads, each with a share of the rows (or of the bytes of a single row), overlap their stagings.
hipError_t CopyToHost(int device, void *dst, size_t dst_pitch, const void *src, size_t src_pitch, size_t width, size_t rows, int threads) {
  if (width == 0 || rows == 0) return hipSuccess;
  if (dst_pitch == width && src_pitch == width) {   // contiguous: one long row, cut anywhere
    width *= rows;
    rows = 1;
    dst_pitch = src_pitch = width;
  }
  const size_t total = width * rows;
  if (IsPinnedHost(dst) || threads <= 1 || total < (32ull << 20)) return hipMemcpy2D(dst, dst_pitch, src, src_pitch, width, rows, hipMemcpyDeviceToHost);
  std::vector<hipError_t> status(threads, hipSuccess);
  std::vector<std::thread> workers;
  for (int t = 0; t < threads; t++) {
    workers.emplace_back([=, &status]() {
      status[t] = hipSetDevice(device);
      if (status[t] != hipSuccess) return;
      if (rows == 1) {   // shares of the bytes, on 4 KiB boundaries
        const size_t share = ((width + threads - 1) / threads + 4095) / 4096 * 4096;
        const size_t first = std::min(width, share * t), last = std::min(width, share * (t + 1));
        if (last > first) status[t] = hipMemcpy(static_cast<char *>(dst) + first, static_cast<const char *>(src) + first, last - first, hipMemcpyDeviceToHost);
      } else {
        const size_t first = rows * t / threads, last = rows * (t + 1) / threads;
        if (last > first)
          status[t] = hipMemcpy2D(static_cast<char *>(dst) + first * dst_pitch, dst_pitch, static_cast<const char *>(src) + first * src_pitch, src_pitch, width, last - first,
                                  hipMemcpyDeviceToHost);
      }
    });
  }
  for (std::thread &w : workers) w.join();
  for (hipError_t e : status)
    if (e != hipSuccess) return e;
  return hipSuccess;
}

// The outputs of the rays [begin, begin + count) of the call - columns of every row - with `threads` host threads
hipError_t DownloadColumns(const RenderJob &job, long long begin, long long count, int threads) {
  bl_ctx *ctx = job.ctx;
  const bl_render_desc *d = job.d;
  const size_t n_rays = static_cast<size_t>(job.n_rays), first = static_cast<size_t>(begin), n = static_cast<size_t>(count);
  hipError_t err = hipSuccess;
  auto rows = [&](void *dst, const void *src, size_t element, size_t n_rows) {   // [n_rows][n_rays] arrays of `element` bytes
    if (dst == nullptr || err != hipSuccess) return;
    err = CopyToHost(ctx->device, static_cast<char *>(dst) + first * element, n_rays * element, static_cast<const char *>(src) + first * element, n_rays * element, n * element, n_rows,
                     threads);
  };
  if (job.n_q > 0) rows(d->image, job.image, sizeof(double), static_cast<size_t>(job.n_q));
  rows(d->sample_num, job.out_num, sizeof(int), 1);
  rows(d->sample_flags, job.out_flags, 1, 1);
  rows(d->camera_pos, job.cam_pos, 32, 1);   // [n_rays][4]
  rows(d->camera_dir, job.cam_dir, 32, 1);
  if (ctx->render_num_images > 0) rows(d->render, job.render_out, sizeof(double), static_cast<size_t>(ctx->render_num_images) * 3);
  return err;
}

// A finished chunk's columns on their way while the next chunk renders (RenderJob::raster): a host thread of its own, since a copy into
// pageable memory blocks the thread that asks for it
void DownloadChunk(RenderJob &job, long long begin, long long count) {
  job.download_status.push_back(hipSuccess);
  hipError_t *status = &job.download_status.back();
  const RenderJob *const_job = &job;
  const int device = job.ctx->device;
  job.downloads.emplace_back([=]() {
    *status = hipSetDevice(device);
    if (*status == hipSuccess) *status = DownloadColumns(*const_job, begin, count, 1);
  });
}

void DownloadOutputs(RenderJob &job) {
  const bl_render_desc *d = job.d;
  if (d->outputs_on_device) return;
  if (job.chunk_downloads) {   // every chunk went as it finished
    for (std::thread &t : job.downloads)
      if (t.joinable()) t.join();
    for (hipError_t e : job.download_status) Check(e, "download of a chunk's outputs");
    return;
  }
  Check(DownloadColumns(job, 0, job.n_rays, 1), "download of the outputs");   // (pageable memory: several threads bring nothing - measured, 4 threads 41.6 ms against 33.7 for 537 MB: the pages' first touch is what it waits for)
}

// bl_stats of the call, and the reference's warning about rays that ended unexpectedly (geodesics.cpp:389-394)
void FinishStats(RenderJob &job) {
  bl_ctx *ctx = job.ctx;
  const bl_params &p = ctx->params;
  bl_stats st{};
  st.n_rays = job.n_rays;
  st.n_chunks = job.n_chunks;
  st.launches_geodesic = job.reuse ? 0 : job.n_chunks;
  st.launches_locate = (job.simulation && !job.locate_inside && !job.reuse_located) ? job.n_chunks : 0;
  st.geodesics_reused = job.reuse ? 1 : 0;
  st.sampling_reused = job.reuse_located ? 1 : 0;
  st.launches_shade = job.n_chunks;
  st.launches_transfer = job.n_chunks;
  st.n_samples = static_cast<int64_t>(job.total_samples);
  st.n_samples_emitted = static_cast<int64_t>(job.total_records);
  st.n_gathers = static_cast<int64_t>(job.total_gathers);
  st.n_flagged = static_cast<int64_t>(job.total_flagged);
  st.max_sample_num = static_cast<int32_t>(job.max_num);
  const double bytes_per_gather = (job.simulation && !p.simulation_interp) ? 32.0 : 256.0;
  st.algorithmic_bytes = bytes_per_gather * static_cast<double>(job.total_gathers) + 13.0 * static_cast<double>(job.n_rays);
  st.ms_geodesic = (job.geo_load || job.reuse) ? 0.0f : job.ms_geo;   // nothing was integrated
  st.ms_locate = job.ms_locate;
  st.ms_shade = job.ms_shade;
  st.ms_transfer = job.ms_transfer;
  st.ms_total = job.ms_geo + job.ms_locate + job.ms_shade + job.ms_transfer;
  float ms_wall = 0.0f;
  Check(hipEventElapsedTime(&ms_wall, ctx->events[2 * kEventsPerChunk], ctx->events[2 * kEventsPerChunk + 1]), "event time");
  st.ms_wall = ms_wall;
  st.arithmetic = (job.fast || job.fast_formula || job.tolerant_polarized) ? BL_ARITH_TOLERANT : BL_ARITH_EXACT;
  st.n_deferred = static_cast<int64_t>(job.total_redo);
  st.n_undefined = static_cast<int64_t>(job.total_undefined);
  st.switches = ctx->switches;
  st.fused_variant = job.fused2 ? 2 : (job.exact_fused ? 3 : (job.pol_fused ? 4 : 0));
  st.n_parked = static_cast<int64_t>(job.reuse ? ctx->resident.n_parked : job.total_parked);
  st.composed_maps = job.composed ? 1 : 0;
  st.tail_policy = job.reuse ? ctx->resident.tail_policy : (job.park ? BL_TAIL_QUAD : (job.split_long ? BL_TAIL_SPLIT : BL_TAIL_WIDE));
  ctx->stats = st;
  if (ctx->debug_counters) {   // kernels built with -DBL_GEO_STATS fill these
    std::fprintf(stderr, "debug counters:");
    for (int k = 0; k < 8; k++) std::fprintf(stderr, " %llu", job.debug_counters[k]);
    std::fprintf(stderr, "\n");
  }
  if (job.total_flagged > 0 && !job.reuse)   // (geodesics.cpp:389-394: raised where the geodesics are integrated - once per series)
    Warn(ctx, std::to_string(job.total_flagged) + " out of " + std::to_string(job.n_rays) + " geodesics terminate unexpectedly.");
  if (job.total_undefined > 0)   // BL_UNDEFINED_EDGE (this text has no counterpart in the reference)
    Warn(ctx, std::to_string(job.total_undefined) + " samples lie where the reference reads past its arrays; the edge cell was used for them.");
}

// Slow light (simulation_sampling.cpp:553-617): pixels whose samples fall outside the window of files
void SlowLightMessages(RenderJob &job) {
  bl_ctx *ctx = job.ctx;
  const bl_params &p = ctx->params;
  const long long n_rays = job.n_rays;
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
    message << "Snapshot " << ctx->snapshot << " at time " << job.snapshot_time << " requires " << degree << " extrapolation "
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

}  // namespace

namespace {
// BL_TAIL_SPLIT: two streams whose CU masks partition the device - bit c of a mask = compute unit c may run the stream's kernels -
// `cus` compute units for bl_geodesic_quad_kernel, the rest for bl_geodesic_kernel. False when the runtime refuses.
// The pair belongs to the PROCESS, one per (device, cus), and is never destroyed: contexts borrow it. (Destroying a CU-masked
// stream leaves its queue in the runtime's list - ROCr 7.0 as torch ships it: the next device allocation that has to trim scratch,
// GpuAgent::Trim -> AqlQueue::AsyncReclaimMainScratch, walks into it. Found by the -m gpu suite, 300 contexts into the run. Two
// contexts of one device that render at once share the pair; their kernels queue behind each other there, which orders nothing
// that the events of each call do not order already.)
bool EnsureSplitStreams(bl_ctx *ctx, int cus) {
  if (ctx->stream_few != nullptr && ctx->split_cus_made == cus) return true;
  static std::mutex table_lock;
  static std::map<std::pair<int, int>, std::pair<hipStream_t, hipStream_t>> table;
  std::lock_guard<std::mutex> guard(table_lock);
  auto found = table.find({ctx->device, cus});
  if (found == table.end()) {
    const int words = (ctx->num_cus + 31) / 32;
    std::vector<uint32_t> few(words, 0u), most(words, 0u);
    for (int c = 0; c < ctx->num_cus; c++) (c < cus ? few : most)[c / 32] |= 1u << (c % 32);
    hipStream_t stream_few = nullptr, stream_most = nullptr;
    if (hipExtStreamCreateWithCUMask(&stream_few, static_cast<uint32_t>(words), few.data()) != hipSuccess
        || hipExtStreamCreateWithCUMask(&stream_most, static_cast<uint32_t>(words), most.data()) != hipSuccess) {
      (void)hipGetLastError();
      return false;   // (a first stream that was made stays: see above)
    }
    found = table.emplace(std::make_pair(ctx->device, cus), std::make_pair(stream_few, stream_most)).first;
  }
  ctx->stream_few = found->second.first;
  ctx->stream_most = found->second.second;
  ctx->split_cus_made = cus;
  return true;
}
}  // namespace

namespace blhost {
void EnsureStreams(bl_ctx *ctx) {
  if (ctx->stream == nullptr) Check(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking), "hipStreamCreate");
  if (ctx->stream_geo == nullptr) Check(hipStreamCreateWithFlags(&ctx->stream_geo, hipStreamNonBlocking), "hipStreamCreate");
}
}  // namespace blhost

extern "C" int bl_render(bl_ctx *ctx, const bl_render_desc *d) {
  if (ctx == nullptr || d == nullptr) return BL_E_ARG;
  std::lock_guard<std::mutex> render_guard(ctx->render_lock);   // (bl_set_grid on another host thread stages the next snapshot meanwhile: bl_api.hip)
  auto drain = [ctx]() {   // leave no chunk half collected behind: a later call starts from idle streams
    if (ctx->stream_few != nullptr) (void)hipStreamSynchronize(ctx->stream_few);
    if (ctx->stream_most != nullptr) (void)hipStreamSynchronize(ctx->stream_most);
    if (ctx->stream_geo != nullptr) (void)hipStreamSynchronize(ctx->stream_geo);
    if (ctx->stream != nullptr) (void)hipStreamSynchronize(ctx->stream);
  };
  try {
    // (planned again at most twice: without the split stepper after a chunk that closed its gate early, without the resident
    // records after scratch beside them could not be had)
    bool allow_split = true, allow_reuse = true;
    for (int attempt = 0;; attempt++) {
      RenderJob job;
      job.ctx = ctx;
      job.d = d;
      job.allow_split = allow_split;
      job.allow_reuse = allow_reuse;
      try {
        PlanJob(job);
        Check(hipSetDevice(ctx->device), "hipSetDevice");
        EnsureStreams(ctx);
        DecideReuse(job);
        PlanScratch(job);
        if (job.split_long && !EnsureSplitStreams(ctx, job.split_cus)) {
          if (ctx->tail_policy == BL_TAIL_SPLIT) throw Failure{BL_E_DEVICE, "BL_TAIL_SPLIT: the runtime gave no stream with a CU mask (hipExtStreamCreateWithCUMask)."};
          ctx->split_unavailable = true;   // BL_TAIL_AUTO: the one-stepper path, from now on
          job.split_long = false;
          job.park_capacity = 0;
          job.quad_grid = 0;
        }
        EnsureScratch(job);
        StageInputsAndOutputs(job);
        BuildTraceArgs(job);
        BuildShadeArgs(job);
        BuildTransferArgs(job);
        RunChunks(job);
        if (job.geo_save) WriteGeodesicCheckpoint(job);
        if (job.sample_save) WriteSampleCheckpoint(job);
        DownloadOutputs(job);
        FinishStats(job);
        KeepResident(job);
        if (job.slow) SlowLightMessages(job);
        break;
      } catch (const SplitIncomplete &) {
        drain();
        if (job.keepable) DropResident(ctx);
        if (!allow_split || attempt >= 2) throw Failure{BL_E_DEVICE, "A chunk of rays ended early twice."};
        allow_split = false;
      } catch (const ReuseImpossible &) {
        drain();
        (void)hipGetLastError();
        DropResident(ctx);
        if (!allow_reuse || attempt >= 2) throw Failure{BL_E_DEVICE, "Scratch memory for the render could not be allocated."};
        allow_reuse = false;
      } catch (const Failure &) {
        // (what scratch set 0 holds is no longer known to be a whole set of records)
        if (job.keepable) DropResident(ctx);
        throw;
      }
    }
  } catch (const Failure &failure) {
    drain();
    return Fail(ctx, failure);
  }
  return BL_OK;
}

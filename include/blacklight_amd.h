/* blacklight_amd.h - C-ABI of the MI355X-native hot path of blacklight (c-white/blacklight).
 *
 * The reference has no plugin / FFI boundary: its hot path is entered by three C++ member calls
 * made from main() (reference src/blacklight.cpp:93-94, 203-204, 221):
 *
 *     double GeodesicIntegrator::Integrate()                      geodesic_integrator.cpp:194
 *     bool   RadiationIntegrator::Integrate(int, double*, ...)    radiation_integrator.cpp:676
 *     double GeodesicIntegrator::AddGeodesics(RadiationIntegrator*) geodesic_integrator.cpp:236
 *
 * with inputs = the InputReader fields copied in the two constructors
 * (geodesic_integrator.cpp:26-104, radiation_integrator.cpp:30-357) plus the read-only grid arrays
 * owned by SimulationReader (simulation_sampling.cpp:38-78), and outputs = image[level](n_q,n_pix),
 * sample_num, sample_flags, camera_pos/dir (output_writer.cpp:116-124, 172-246).
 *
 * This header is the drop-in replacement for exactly that surface, as plain C: a parameter block
 * that mirrors InputReader key-for-key (bl_params), a borrowed view of the grid (bl_grid_desc), and
 * one call per adaptive level (bl_render) that runs camera + geodesics + sampling + coefficients +
 * transfer on the GPU. No C++ or torch types cross the boundary; every function returns 0 on
 * success or a BL_E_* code, with the message text (identical to the reference's where the
 * reference has one) available from bl_last_error().
 */
#ifndef BLACKLIGHT_AMD_H_
#define BLACKLIGHT_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BL_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ error codes */
enum {
  BL_OK = 0,
  BL_E_INPUT = 1,      /* malformed .input file or bad value: message = reference text            */
  BL_E_MISSING = 2,    /* a required key is absent (reference: std::bad_optional_access)          */
  BL_E_UNSUPPORTED = 3,/* valid reference configuration outside the hot-path scope of this build  */
  BL_E_DEVICE = 4,     /* HIP runtime failure or no gfx950 device                                 */
  BL_E_ARG = 5,        /* bad argument to the C-ABI itself                                        */
  BL_E_STATE = 6       /* call out of order (e.g. simulation render before bl_set_grid)           */
};

/* ------------------------------------------------------------------ enumerations
 * Same members, same order as reference src/blacklight.hpp:36-46. */
enum { BL_MODEL_SIMULATION = 0, BL_MODEL_FORMULA = 1 };
enum { BL_OUTPUT_NPZ = 0, BL_OUTPUT_NPY = 1, BL_OUTPUT_RAW = 2 };
enum { BL_SIMFMT_ATHENA = 0, BL_SIMFMT_ATHENAK = 1, BL_SIMFMT_IHARM3D = 2, BL_SIMFMT_HARM3D = 3 };
enum { BL_COORD_CKS = 0, BL_COORD_SKS = 1, BL_COORD_FMKS = 2 };
enum { BL_CAMERA_PLANE = 0, BL_CAMERA_PINHOLE = 1 };
enum { BL_TERMINATE_PHOTON = 0, BL_TERMINATE_MULTIPLICATIVE = 1, BL_TERMINATE_ADDITIVE = 2 };
enum { BL_INTEGRATOR_DP = 0, BL_INTEGRATOR_RK4 = 1, BL_INTEGRATOR_RK2 = 2 };
enum { BL_SPACING_LIN_FREQ = 0, BL_SPACING_LIN_WAVE = 1, BL_SPACING_LOG = 2 };
enum { BL_NORM_CAMERA = 0, BL_NORM_INFINITY = 1 };
enum { BL_PLASMA_TI_TE_BETA = 0, BL_PLASMA_CODE_KAPPA = 1 };

#define BL_STR_LEN 512
#define BL_MAX_REGIONS 32

/* ------------------------------------------------------------------ parameter block
 * One entry per key of the reference's .input grammar (src/input_reader/input_reader.cpp:101-407),
 * in the same order. X(kind, name):
 *   B bool (int32 0/1)   I int32   D double   F float   S string   E enum (int32)
 *   G double given in degrees in the file, stored in radians as val * pi / 180
 *     (input_reader.cpp:185-201, 389)
 * camera_th additionally sets camera_pole when the written value is exactly 0 or 180
 * (input_reader.cpp:492-500).
 */
#define BL_PARAM_LIST(X) \
  X(E, model_type) X(I, num_threads) \
  X(E, output_format) X(S, output_file) X(B, output_camera) \
  X(B, checkpoint_geodesic_save) X(B, checkpoint_geodesic_load) X(S, checkpoint_geodesic_file) \
  X(B, checkpoint_sample_save) X(B, checkpoint_sample_load) X(S, checkpoint_sample_file) \
  X(E, simulation_format) X(S, simulation_file) X(B, simulation_multiple) X(I, simulation_start) \
  X(I, simulation_end) X(E, simulation_coord) X(D, simulation_a) X(D, simulation_m_msun) \
  X(D, simulation_rho_cgs) X(S, simulation_kappa_name) X(B, simulation_interp) \
  X(B, simulation_block_interp) \
  X(D, formula_mass) X(D, formula_spin) X(D, formula_r0) X(D, formula_h) X(D, formula_l0) \
  X(D, formula_q) X(D, formula_nup) X(D, formula_cn0) X(D, formula_alpha) X(D, formula_a) \
  X(D, formula_beta) \
  X(E, camera_type) X(D, camera_r) X(G, camera_th) X(G, camera_ph) X(D, camera_urn) \
  X(D, camera_uthn) X(D, camera_uphn) X(D, camera_k_r) X(D, camera_k_th) X(D, camera_k_ph) \
  X(G, camera_rotation) X(D, camera_width) X(I, camera_resolution) X(B, camera_pole) \
  X(B, ray_flat) X(E, ray_terminate) X(D, ray_factor) X(E, ray_integrator) X(D, ray_step) \
  X(I, ray_max_steps) X(I, ray_max_retries) X(D, ray_tol_abs) X(D, ray_tol_rel) \
  X(B, image_light) X(I, image_num_frequencies) X(D, image_frequency) X(D, image_frequency_start) \
  X(D, image_frequency_end) X(E, image_frequency_spacing) X(E, image_normalization) \
  X(B, image_polarization) X(B, image_rotation_split) X(B, image_time) X(B, image_length) \
  X(B, image_lambda) X(B, image_emission) X(B, image_tau) X(B, image_lambda_ave) \
  X(B, image_emission_ave) X(B, image_tau_int) X(B, image_crossings) \
  X(I, render_num_images) \
  X(B, slow_light_on) X(B, slow_interp) X(I, slow_chunk_size) X(D, slow_t_start) X(D, slow_dt) \
  X(I, slow_num_images) X(I, slow_offset) \
  X(I, adaptive_max_level) X(I, adaptive_block_size) X(I, adaptive_frequency_num) \
  X(D, adaptive_val_cut) X(D, adaptive_val_frac) X(D, adaptive_abs_grad_cut) \
  X(D, adaptive_abs_grad_frac) X(D, adaptive_rel_grad_cut) X(D, adaptive_rel_grad_frac) \
  X(D, adaptive_abs_lapl_cut) X(D, adaptive_abs_lapl_frac) X(D, adaptive_rel_lapl_cut) \
  X(D, adaptive_rel_lapl_frac) X(I, adaptive_num_regions) \
  X(D, plasma_mu) X(D, plasma_ne_ni) X(E, plasma_model) X(B, plasma_use_p) X(D, plasma_gamma) \
  X(D, plasma_gamma_i) X(D, plasma_gamma_e) X(D, plasma_rat_low) X(D, plasma_rat_high) \
  X(D, plasma_power_frac) X(D, plasma_p) X(D, plasma_gamma_min) X(D, plasma_gamma_max) \
  X(D, plasma_kappa_frac) X(D, plasma_kappa) X(D, plasma_w) \
  X(D, cut_rho_min) X(D, cut_rho_max) X(D, cut_n_e_min) X(D, cut_n_e_max) X(D, cut_p_gas_min) \
  X(D, cut_p_gas_max) X(D, cut_theta_e_min) X(D, cut_theta_e_max) X(D, cut_b_min) X(D, cut_b_max) \
  X(D, cut_sigma_min) X(D, cut_sigma_max) X(D, cut_beta_inverse_min) X(D, cut_beta_inverse_max) \
  X(B, cut_omit_near) X(B, cut_omit_far) X(D, cut_omit_in) X(D, cut_omit_out) \
  X(G, cut_midplane_theta) X(D, cut_midplane_z) X(B, cut_plane) \
  X(D, cut_plane_origin_x) X(D, cut_plane_origin_y) X(D, cut_plane_origin_z) \
  X(D, cut_plane_normal_x) X(D, cut_plane_normal_y) X(D, cut_plane_normal_z) \
  X(B, fallback_nan) X(F, fallback_rho) X(F, fallback_pgas) X(F, fallback_kappa)

typedef struct bl_str { char s[BL_STR_LEN]; } bl_str;

#define BL_PT_B int32_t
#define BL_PT_I int32_t
#define BL_PT_E int32_t
#define BL_PT_D double
#define BL_PT_G double
#define BL_PT_F float
#define BL_PT_S bl_str
#define BL_X_FIELD(kind, name) BL_PT_##kind name;
#define BL_X_INDEX(kind, name) BL_P_##name,

enum { BL_PARAM_LIST(BL_X_INDEX) BL_P_COUNT };

/* False-colour rendering (rendering.cpp): bounds of this build and enumerations */
#define BL_MAX_RENDER_IMAGES 4
#define BL_MAX_RENDER_FEATURES 16
enum { BL_RENDER_FILL = 0, BL_RENDER_THRESH = 1, BL_RENDER_RISE = 2, BL_RENDER_FALL = 3 };
enum { BL_RENDER_HAS_QUANTITY = 1, BL_RENDER_HAS_TYPE = 2, BL_RENDER_HAS_MIN = 4, BL_RENDER_HAS_MAX = 8,
       BL_RENDER_HAS_THRESH = 16, BL_RENDER_HAS_TAU_SCALE = 32, BL_RENDER_HAS_OPACITY = 64, BL_RENDER_HAS_XYZ = 128 };

typedef struct bl_params {
  /* has[i] != 0 iff field i (enum BL_P_*) was given: the std::optional of the reference */
  uint8_t has[((BL_P_COUNT + 7) / 8) * 8];
  BL_PARAM_LIST(BL_X_FIELD)
  /* adaptive_region_<n>_{level,x_min,x_max,y_min,y_max}  (src/input_reader/adaptive_reader.cpp) */
  int32_t adaptive_region_has[BL_MAX_REGIONS];     /* bit 0..4 = level, x_min, x_max, y_min, y_max */
  int32_t adaptive_region_level[BL_MAX_REGIONS];
  double adaptive_region_x_min[BL_MAX_REGIONS];
  double adaptive_region_x_max[BL_MAX_REGIONS];
  double adaptive_region_y_min[BL_MAX_REGIONS];
  double adaptive_region_y_max[BL_MAX_REGIONS];
  /* render_<i>_num_features, render_<i>_<f>_{quantity,type,min,max,thresh,tau_scale,opacity,rgb|xyz}
   * (src/input_reader/render_reader.cpp); indices beyond render_num_images / num_features are ignored
   * as there, indices beyond BL_MAX_RENDER_* are an error of this build */
  int32_t render_num_features_has[BL_MAX_RENDER_IMAGES];
  int32_t render_num_features[BL_MAX_RENDER_IMAGES];
  int32_t render_has[BL_MAX_RENDER_IMAGES][BL_MAX_RENDER_FEATURES];   /* BL_RENDER_HAS_* bits */
  int32_t render_quantity[BL_MAX_RENDER_IMAGES][BL_MAX_RENDER_FEATURES];   /* cell value index 0..6 */
  int32_t render_type[BL_MAX_RENDER_IMAGES][BL_MAX_RENDER_FEATURES];       /* BL_RENDER_* */
  double render_min[BL_MAX_RENDER_IMAGES][BL_MAX_RENDER_FEATURES];
  double render_max[BL_MAX_RENDER_IMAGES][BL_MAX_RENDER_FEATURES];
  double render_thresh[BL_MAX_RENDER_IMAGES][BL_MAX_RENDER_FEATURES];
  double render_tau_scale[BL_MAX_RENDER_IMAGES][BL_MAX_RENDER_FEATURES];
  double render_opacity[BL_MAX_RENDER_IMAGES][BL_MAX_RENDER_FEATURES];
  double render_x[BL_MAX_RENDER_IMAGES][BL_MAX_RENDER_FEATURES];      /* XYZ colour (D65), rgb keys converted */
  double render_y[BL_MAX_RENDER_IMAGES][BL_MAX_RENDER_FEATURES];
  double render_z[BL_MAX_RENDER_IMAGES][BL_MAX_RENDER_FEATURES];
} bl_params;

/* Zero a parameter block (nothing present). */
BL_API void bl_params_clear(bl_params *p);
BL_API size_t bl_params_sizeof(void);
/* Parse one "key = value" assignment exactly as a line of the .input file would be
 * (whitespace stripped, '#' comments removed); err receives "Error: ...\n" text on failure. */
BL_API int bl_params_set_line(bl_params *p, const char *line, char *err, size_t err_len);
/* Parse a whole .input file: replaces InputReader::Read() (input_reader.cpp:72-428).
 * *num_runs receives the run count (1 unless simulation_multiple). */
BL_API int bl_params_read_file(bl_params *p, const char *path, int *num_runs, char *err,
                               size_t err_len);
/* Typed read-back by key name (used by the Python host layer and the tests). */
BL_API int bl_params_get(const bl_params *p, const char *key, double *value, int *present);
BL_API int bl_params_get_string(const bl_params *p, const char *key, char *out, size_t out_len);

/* ------------------------------------------------------------------ grid view
 * What RadiationIntegrator::ObtainGridData() takes from SimulationReader
 * (simulation_sampling.cpp:26-95). All pointers are host pointers borrowed for the duration of
 * bl_set_grid(); layouts are the reference's Array<T> layouts (n1 fastest):
 *   prim  float  [n_var][n_blocks][n_k][n_j][n_i]     simulation_reader.cpp:767-780
 *   x1f   double [n_blocks][n_i+1], x1v double [n_blocks][n_i]  (x2*, x3* alike)
 * Coordinates are the values the reader hands over, i.e. file float32 promoted to double
 * (simulation_reader.cpp:615-620) after its angular-range fix (:724-758). */
typedef struct bl_grid_desc {
  int32_t n_blocks, n_i, n_j, n_k, n_var;
  const float *prim;
  const double *x1f, *x2f, *x3f, *x1v, *x2v, *x3v;
  int32_t ind_rho, ind_pgas, ind_kappa, ind_uu1, ind_uu2, ind_uu3, ind_bb1, ind_bb2, ind_bb3;
  double plasma_gamma, plasma_gamma_i, plasma_gamma_e; /* possibly modified by the reader */
  /* MeshBlock table, read only with simulation_block_interp = true (simulation_sampling.cpp:36-39, :84-93):
   * refinement level of each block, its logical location (i, j, k) on that level, and the number of cells
   * of the root grid in x^3 (RootGridSize[2]). NULL / 0 otherwise. */
  const int32_t *levels;      /* [n_blocks] */
  const int32_t *locations;   /* [n_blocks][3] */
  int32_t n_3_root;
  /* simulation_coord = fmks (iharm3d FMKS / MMKS grids; simulation_sampling.cpp:66-73, :190-198, :396-456): x1f ... x3v
   * stay in the file's native coordinates (x^1 = log r, x^2 in [0, 1], x^3 = phi, one block, uniform), and a sample's
   * position on them comes from the reader's look-up table from spherical Kerr-Schild (r, theta) to (x^1, x^2)
   * (SimulationReader::GenerateSKSMap, simulation_geometry.cpp:321-419) within the grid's bounds in (r, theta, phi)
   * (simulation_bounds, :46-57). NULL / 0 for every other coordinate system. */
  const double *sks_map;      /* [2][sks_map_n2][sks_map_n1]: x^1, then x^2 */
  int32_t sks_map_n1, sks_map_n2;
  double sks_map_r_in, sks_map_dr, sks_map_dtheta;
  double simulation_bounds[6];   /* r_min, r_max, theta_min, theta_max, phi_min, phi_max */
} bl_grid_desc;

/* ------------------------------------------------------------------ snapshot reader (host only)
 * SimulationReader for simulation_format = athena: constructor checks (simulation_reader.cpp:36-159),
 * Read() (:211-861: file name for `snapshot` via FormatFilename :870-904, Time, Levels, LogicalLocations,
 * coordinates with the angular-range fix :722-758, VerifyVariablesAthena :1141-1216, "prim" + "B" cell
 * data :761-781) over the subset of HDF5 the reference decodes by hand (hdf5_format_*.cpp: superblock 0,
 * version-1 B-trees / heaps / object headers / attributes, contiguous version-3 layouts). The snapshot
 * owns the arrays bl_snapshot_grid() points to; hand that view to bl_set_grid(), then close the snapshot.
 * err receives "Error: ...\n" (reference texts). Each call reads its file completely; the reference
 * re-uses block layout and coordinates of the first file for later files of a series.
 * simulation_format = athenak: the AthenaK binary dump reader (simulation_reader.cpp:915-1131, :434-589) behind the same calls.
 * simulation_format = iharm3d / harm3d: modified Kerr-Schild (MKS) dumps read with simulation_coord = sks (coordinates and
 * vector components converted like SimulationReader does); iharm3d FMKS / MMKS dumps read with simulation_coord = fmks
 * (native coordinates kept, bl_grid_desc::sks_map built). With slow_light_on use bl_slow_light_read(). */
typedef struct bl_snapshot bl_snapshot;
BL_API int bl_snapshot_open(const bl_params *p, int snapshot, bl_snapshot **out, char *err, size_t err_len);
BL_API const bl_grid_desc *bl_snapshot_grid(const bl_snapshot *s);
BL_API double bl_snapshot_time(const bl_snapshot *s);              /* file attribute "Time"           */
BL_API const char *bl_snapshot_warnings(const bl_snapshot *s);     /* "Warning: ...\n" lines           */
/* How many leading bytes of bl_snapshot_warnings() stem from SimulationReader's constructor checks (the reference
 * prints those before RadiationIntegrator's constructor warnings - bl_warnings() after bl_init() - and the file's
 * own warnings after them). */
BL_API size_t bl_snapshot_setup_warning_bytes(const bl_snapshot *s);
BL_API const char *bl_snapshot_file(const bl_snapshot *s);         /* file name actually opened       */
/* MeshBlock table: returns n_blocks; *levels -> [n_blocks], *locations -> [n_blocks][3] */
BL_API int bl_snapshot_blocks(const bl_snapshot *s, const int32_t **levels, const int32_t **locations);
BL_API void bl_snapshot_close(bl_snapshot *s);
/* Open file number `file_number` of the series (simulation_file with its {Nd} field filled in), whatever
 * simulation_multiple / slow_light_on say: the building block of bl_slow_light_read(). */
BL_API int bl_snapshot_open_number(const bl_params *p, int file_number, bl_snapshot **out, char *err, size_t err_len);

/* ------------------------------------------------------------------ camera frame
 * The seven 4-vectors GeodesicIntegrator::InitializeCamera() derives (camera.cpp:61-380,
 * geodesic_integrator.hpp) and the frequency list (:30-50). Filled by bl_init. */
typedef struct bl_camera_frame {
  double cam_x[4], u_con[4], u_cov[4], norm_con[4], norm_con_c[4], hor_con_c[4], vert_con_c[4];
  double bh_m, bh_a, r_horizon, r_terminate, mass_msun;
} bl_camera_frame;

/* ------------------------------------------------------------------ one render call
 * Replaces, for one adaptive level, InitializeCamera/AugmentCamera + IntegrateGeodesics* +
 * ReverseGeodesics + CalculateSimulationSampling + SampleSimulation + Calculate*Coefficients +
 * IntegrateUnpolarizedRadiation / IntegratePolarizedRadiation (image rows 4 l + (I, Q, U, V) with
 * image_polarization). The refinement decision between levels
 * (radiation_adaptive.cpp) stays with the caller, as in blacklight.cpp:196-233. */
typedef struct bl_render_desc {
  int32_t level;              /* 0 = root camera; L>0 = refined blocks at res * 2^L              */
  int32_t n_blocks;           /* level>0: number of blocks                                        */
  const int32_t *block_locs;  /* level>0: host [n_blocks][2] = (block_v, block_u), camera.cpp:457  */
  int64_t n_rays;             /* rays to trace in this call                                       */
  const int32_t *pixel_map;   /* optional host [n_rays]: ray r is pixel pixel_map[r] of the level's
                                 pixel array (multi-GPU tiling); NULL = pixels 0..n_rays-1        */
  int32_t outputs_on_device;  /* 0: pointers below are host memory; 1: device memory (HBM)        */
  double *image;              /* [n_q][n_rays] f64, row order = reference image offsets            */
  int32_t *sample_num;        /* [n_rays] or NULL                                                  */
  uint8_t *sample_flags;      /* [n_rays] or NULL                                                  */
  double *camera_pos;         /* [n_rays][4] or NULL                                               */
  double *camera_dir;         /* [n_rays][4] or NULL                                               */
  double *render;             /* render_num_images > 0: [render_num_images][3][n_rays] XYZ, else NULL
                                 (RadiationIntegrator::Render, rendering.cpp:25-179)                  */
} bl_render_desc;

typedef struct bl_stats {
  int64_t n_rays;             /* rays traced by the last bl_render                                 */
  int64_t n_samples;          /* kept samples (sum of sample_num)                                  */
  int64_t n_samples_emitted;  /* samples written by the geodesic kernel (before truncation)        */
  int64_t n_gathers;          /* S_in: samples that read the grid (8 var x 8 corners, or x1)        */
  int64_t n_flagged;          /* rays with sample_flags set                                        */
  int32_t max_sample_num;     /* geodesic_num_steps                                                */
  int32_t n_chunks;
  double algorithmic_bytes;   /* 256 (or 32) * n_gathers + 13 * n_rays                */
  float ms_geodesic, ms_shade, ms_transfer, ms_total; /* HIP-event kernel time, last bl_render: geodesic,
                                 coefficient (bl_shade_kernel) and transfer kernels; ms_total = sum of all four    */
  int32_t launches_geodesic, launches_shade, launches_transfer;
  float ms_locate;            /* locate kernel (simulation mode)                                      */
  float ms_wall;              /* first kernel start to last kernel end; less than ms_total when the geodesic
                                 kernel of one chunk overlaps the shading of the previous one          */
  int32_t launches_locate;
  int32_t arithmetic;         /* BL_ARITH_EXACT or BL_ARITH_TOLERANT: the tier the last bl_render ran in          */
  int64_t n_deferred;         /* tolerant tier: samples whose cut decision was left to the exact kernel           */
  int64_t n_undefined;        /* BL_UNDEFINED_EDGE: samples where the reference reads past its arrays (edge cell used)     */
  uint32_t switches;          /* BL_SWITCH_* bits active in this context (measurement switches, below); 0 in production  */
  int32_t fused_variant;      /* the locate step inside the coefficient kernel: 2 = bl_shade_fused2_kernel (tolerant tier),
                                 3 = bl_shade_exact2_kernel (exact tier), 4 = bl_shade_polarized2_kernel (polarized runs),
                                 0 = a locate kernel of its own ran  */
  int64_t n_parked;           /* rays whose last steps ran with a ray per quad of lanes (bl_geodesic_quad_kernel)                 */
  int32_t composed_maps;      /* 1: the tolerant tier composed the affine transfer maps of neighbouring samples (intensities equal from
                                 run to run to rounding, ~1e-15, not bit for bit); 0: every image row of this render is bit-reproducible
                                 (always so in the exact tier and under bl_set_reproducible)                                         */
  int32_t tail_policy;        /* BL_TAIL_* the render ran with (after BL_TAIL_AUTO was resolved)                                     */
  int32_t geodesics_reused;   /* 1: the sample records of an earlier render of the same camera were shaded again - no ray was
                                 integrated (launches_geodesic = 0, ms_geodesic = 0); bl_set_geodesic_reuse                              */
  int32_t sampling_reused;    /* 1: ... and the located samples too (same grid geometry: no locate kernel ran, launches_locate = 0)     */
} bl_stats;

/* Measurement switches: environment variables BLACKLIGHT_AMD_<NAME>, read ONCE by bl_init (never during a render) and echoed in
 * bl_stats.switches, so that a benchmark line says which kernels it timed. They select an alternative kernel or layout with the
 * same results (A/B runs, tests of the general paths); none of them changes an image. */
#define BL_SWITCH_TENSOR_TRANSPORT (1u << 0)                /* polarized, tolerant tier: the exact tier's tensor transport      */
#define BL_SWITCH_SPLIT_RECORDS (1u << 1)                   /* sample records as two arrays of 32-byte halves                   */
#define BL_SWITCH_RECORD_EVERY_STEP (1u << 2)               /* steps in the empty shell around the grid leave records too       */
#define BL_SWITCH_GENERAL_LOCATE (1u << 4)                  /* bl_locate_kernel where bl_locate_plain_kernel applies; a refined
                                                               mesh's tables searched in HBM where they would be staged in LDS */
#define BL_SWITCH_LANE_TRANSFER (1u << 5)                   /* one lane per ray where bl_transfer_quad_kernel applies           */
#define BL_SWITCH_NO_FUSED_LOCATE (1u << 6)                 /* a locate kernel + bl_shade_fast_kernel / bl_shade_exact_kernel   */
#define BL_SWITCH_SAMPLE_RECORDS (1u << 8)                  /* tolerant tier: one transfer record per sample where composed maps apply */
#define BL_SWITCH_QUAD_EVERY_RAY (1u << 11)                 /* every ray parked before its first step: all stepping in bl_geodesic_quad_kernel */
/* (Eight switches. Rounds 3 - 5 had eight more for experiments the measurements buried - a second pre-fused2 kernel, pre-gathered
 * cell bricks, the coefficient kernel beside a chunk's last rays, repacked tails - and for what bl_set_tail_policy now says; their
 * numbers are in docs/notebook.md, their code in the history.) */

typedef struct bl_ctx bl_ctx;

/* Validate parameters exactly like the two reference constructors, compute the camera frame and
 * frequency list on the host, select the device (device = -1: current device) and allocate
 * nothing large yet. device = BL_DEVICE_NONE gives a host-only context (parameter validation, camera
 * frame, bl_adaptive_refine, bl_write_output); bl_set_grid / bl_render on it fail with BL_E_DEVICE. */
#define BL_DEVICE_NONE (-2)
BL_API int bl_init(const bl_params *p, int device, bl_ctx **out);
/* HIP devices visible to this process (0: none). One context per device and one host thread per context is how one process
 * drives several GPUs: bl_render is thread-safe across contexts (bin/blacklight_amd with BLACKLIGHT_AMD_DEVICES does that). */
BL_API int bl_device_count(void);
/* The grid into its HBM layout; once per snapshot. Equal blocks of one level tiling a box are merged into one [k][j][i][8 floats]
 * array; other sets of non-overlapping equal-sized blocks (mesh refinement) stay [block][k][j][i][8] behind a lattice of block
 * boundaries. Overlapping blocks are refused. The variable planes are uploaded as they lie and interleaved on the device (a 256^3
 * snapshot: 33 ms, the PCIe copy of its 537 MB).
 * Beside a render: when the geometry handed over is, to the bit, the one in place (the next snapshot of a series) bl_set_grid may be
 * called on another host thread while bl_render of the same context runs: the cells go up on a stream of their own into a second
 * cell array, and the call returns once that render has ended and the arrays have changed places - the new cells are what the NEXT
 * bl_render reads. One bl_set_grid at a time per context; any other change of the grid waits for the render and excludes it. */
BL_API int bl_set_grid(bl_ctx *ctx, const bl_grid_desc *g);
/* ---- slow light (slow_light_on = true): the reader keeps a window of slow_chunk_size files, latest first
 * (simulation_reader.cpp:211-303), all on the geometry of the first; every sample reads the slice(s) around
 * its own coordinate time (simulation_sampling.cpp:296-349, :736-786, :840-912).
 * bl_slow_light_read(ctx, snapshot): SimulationReader::Read(snapshot) for slow light - advances the window to
 *   cover camera time slow_t_start + snapshot * slow_dt, reading only new files (bl_snapshot_open_number),
 *   shifting the slices already in HBM, and selects `snapshot` for the next bl_render. Reference error and
 *   warning texts ("... would require significant extrapolation beyond file N.").
 * The three calls below are what it is made of, for callers with their own reader:
 * bl_set_grid_slice(ctx, n, g, time): slice n of the window (prim[n], time[n]).
 * bl_shift_grid_slices(ctx, count): slice n <- slice n - count for n >= count (device pointers move, no copy).
 * bl_set_snapshot(ctx, snapshot): image index; camera time and the texts of bl_render's extrapolation
 *   warnings / errors (simulation_sampling.cpp:577-617) follow from it. */
BL_API int bl_slow_light_read(bl_ctx *ctx, int snapshot);
BL_API int bl_set_grid_slice(bl_ctx *ctx, int slice, const bl_grid_desc *g, double time);
BL_API int bl_shift_grid_slices(bl_ctx *ctx, int count);
BL_API int bl_set_snapshot(bl_ctx *ctx, int snapshot);
/* Number of image rows n_q and their offsets (radiation_integrator.cpp:436-520). */
BL_API int bl_image_num_quantities(const bl_ctx *ctx);
/* Number of false-colour renderings bl_render produces (render_num_images; 0 in formula mode). */
BL_API int bl_render_num_images(const bl_ctx *ctx);
BL_API int bl_camera_frame_get(const bl_ctx *ctx, bl_camera_frame *out);
BL_API int bl_frequencies(const bl_ctx *ctx, double *out, int n);
/* Arithmetic tier. BL_ARITH_TOLERANT (what a new context starts in, unless the environment says BLACKLIGHT_AMD_ARITHMETIC=exact): the
 * tolerance north_star grants for intensities ("within a stated fp64 tolerance", per-pixel L-infinity < 1e-6: every pixel relative to
 * its own intensity; measured 2e-14 per pixel - 6e-15 of the image maximum - on the benchmark frame's 1 048 576 pixels and against the
 * reference's own windows of it, asserted at 1e-10 per pixel: tests/test_gpu_window_1024.py, test_gpu_configs_at_size.py) is used between the sampled primitives and the transfer record of a sample -
 * fused multiply-adds, lighter exp / expm1 / cbrt, the fluid-frame angle and frequency as invariants instead of through the tetrad
 * of simulation_coefficients.cpp:398-455, transport matrices in polarized runs. Ray-step counts, flags, cell indices, NaN masks and
 * every cut decision are those of the exact tier. Configurations the tier has no kernel for run in exact arithmetic regardless -
 * bl_stats.arithmetic reports the tier that ran. BL_ARITH_EXACT: every operation in the reference's order with the pinned math
 * library - images equal the reference's bit for bit (tier-B goldens), at 0.57 of the tolerant tier's speed on the benchmark frame.
 * (Rounds 1 - 4 started contexts in the exact tier; the parity suite pins it through the environment variable, tests/conftest.py.) */
/* What to do with a sample for which the reference reads past the end of one of its arrays (no bounds checks in its Array):
 * inter-block interpolation at an upper edge of the file's last MeshBlock (simulation_sampling.cpp:520-522), FMKS sampling in
 * the last polar zone of the last azimuthal plane (:412-415 with :809-819). BL_UNDEFINED_REFUSE (default): bl_render fails with
 * BL_E_UNSUPPORTED - there is no defined result to reproduce. BL_UNDEFINED_EDGE: such samples use the edge of the data that
 * exists (block centre mirrored about the upper face; the zone's own row), and bl_render warns with their number. Every
 * sample the reference defines is unaffected by the choice. */
#define BL_UNDEFINED_REFUSE 0
#define BL_UNDEFINED_EDGE 1
/* BL_UNDEFINED_KAPPA (may be or-ed with BL_UNDEFINED_EDGE): kappa-distribution electrons (plasma_kappa_frac != 0) in an unpolarized
 * run. The reference's absorptivity reads kappa_aa_high_i (simulation_coefficients.cpp:652), which it sets for polarized runs only
 * (:108-121): whatever its allocator left there decides its image. Without this flag bl_render refuses such a run (BL_E_UNSUPPORTED);
 * with it the constant has its polarized definition, (3 / kappa)^4.75 + 0.6, and bl_render says so in a warning. No reference image
 * exists to compare with: the checker is the oracle under the same definition. */
#define BL_UNDEFINED_KAPPA 2
BL_API int bl_set_undefined_policy(bl_ctx *ctx, int policy);
#define BL_ARITH_EXACT 0
#define BL_ARITH_TOLERANT 1
BL_API int bl_set_arithmetic(bl_ctx *ctx, int mode);
/* Tolerant tier only (the exact tier is always bit-reproducible). on = 0 (default): the coefficient kernel composes the affine
 * transfer maps of a ray's neighbouring samples before they leave it (a quarter of the transfer records); which samples are
 * composed together follows the order in which the persistent geodesic waves emitted them, so two renders of one frame - or a
 * frame and its tiles - agree to rounding (~1e-15 of the image maximum), not bit for bit. on = 1: one transfer record per
 * sample, applied in ray order: tolerant images are then bit-identical from run to run and however a frame is cut into
 * bl_render calls, ranks or chunks, like the reference's are across thread counts (blacklight.cpp:196-233 is one deterministic
 * loop). Costs ~1.5 % of the benchmark frame. bl_stats.composed_maps says which way the last render went. */
BL_API int bl_set_reproducible(bl_ctx *ctx, int on);
/* Who steps the rays a chunk of geodesics waits for longest (per-ray independence, geodesics.cpp:109-324; the results are
 * bit-identical whichever way, only the time differs). BL_TAIL_WIDE: the persistent one-ray-per-lane stepper alone.
 * BL_TAIL_QUAD: it parks rays that outlive their neighbours and bl_geodesic_quad_kernel finishes them with a ray per quad of
 * lanes (1.5 - 1.65 x faster per ray; pays where a few rays run for thousands of steps - formula-mode frames - and costs where
 * they do not). BL_TAIL_SPLIT: the rays of a plane camera whose impact parameter lies within 0.12 M of the photon ring's
 * 3 sqrt(3) M - the longest of a frame by a factor of two - are given to bl_geodesic_quad_kernel before their first step, on a
 * stream whose CU mask keeps the other stepper off its compute units (hipExtStreamCreateWithCUMask); pays where the geodesic
 * stage waits for single rays - a rank's share of a tiled frame: an eighth of the benchmark frame 8.0 -> 7.2 ms - and needs a
 * non-rotating hole (the critical curve is a circle), the Dormand-Prince stepper and a call whose rays fit one chunk.
 * BL_TAIL_AUTO (default): QUAD in formula mode; SPLIT where it applies and the call has at most eight rays per lane of the
 * device (a share of a frame tiled over two or more GPUs); else WIDE. bl_stats.tail_policy says what the last render did. */
#define BL_TAIL_AUTO 0
#define BL_TAIL_WIDE 1
#define BL_TAIL_QUAD 2
#define BL_TAIL_SPLIT 3
BL_API int bl_set_tail_policy(bl_ctx *ctx, int policy);
/* Ordering against the caller's own GPU work. bl_render runs on streams of its own (non-blocking: the NULL stream does not order
 * them) and returns when its outputs are complete, so nothing the caller does AFTER the call needs ordering. What the caller
 * queued BEFORE it - a fill of the output buffers, an RCCL gather still reading the previous frame out of them - does:
 * with enabled != 0, every later bl_render first makes its streams wait (on the device, hipStreamWaitEvent: no host wait) for
 * all work queued on `stream` (a hipStream_t; NULL is the NULL stream) up to the moment of the call. */
BL_API int bl_set_caller_stream(bl_ctx *ctx, void *stream, int enabled);
/* Geodesics once per series. The reference integrates the root camera's geodesics ONCE, before its loop over snapshots
 * (blacklight.cpp:93-94 against :178-250), and - while the mesh does not change and slow light is off - locates the samples on the
 * grid once (`first_time`, radiation_integrator.cpp:693-704); per snapshot it only reads the cells, evaluates the coefficients and
 * integrates. on != 0 (default): a root-level bl_render whose rays fit one chunk leaves its sample records (64 B per sample: 48 GB
 * for the 1024^2 benchmark camera) and per-ray rows in HBM; the next root-level bl_render of this context with the SAME camera -
 * same parameters, pixel map, record layout, tail policy - after a new bl_set_grid / bl_slow_light_read shades those records again
 * instead of launching the stepper (bl_stats.geodesics_reused = 1, launches_geodesic = 0), and where a locate kernel of its own
 * ran, skips that too when the grid's geometry is bit for bit the one it located the samples on (bl_stats.sampling_reused).
 * Renders of refined levels in between (the adaptive loop of every snapshot) leave the root level's records alone: they work in
 * buffers of their own. The records go when the camera changes, when a render fails, when memory for another render cannot be
 * had otherwise, and with bl_free. Images are the same bits either way in the exact tier and under bl_set_reproducible (the records
 * ARE what the stepper would write again); with composed maps, equal to rounding like any two renders. The reference's warning
 * about geodesics that end unexpectedly is raised by the render that integrated them, once. on = 0: every render integrates its
 * rays (what a benchmark of the whole pipeline wants: bench.py's headline). */
BL_API int bl_set_geodesic_reuse(bl_ctx *ctx, int on);
/* Host memory the device copies into at the link's rate (pinned: hipHostMalloc). bl_render recognises such buffers among the pointers
 * of bl_render_desc - and any the caller pinned itself (hipHostRegister) - and downloads into them with one asynchronous-engine copy
 * (537 MB of image rows: ~10 ms) where pageable memory goes through the runtime's staging buffer (~16 GB/s per thread; bl_render
 * then splits the copy over four host threads). Large results in many rows (eight image rows or more, a quarter of a GiB or more)
 * leave chunk by chunk while the next chunk renders, whichever kind of memory receives them. NULL when the allocation fails or
 * the context is host-only: fall back to malloc. Free with bl_host_free (NULL is fine). */
BL_API void *bl_host_alloc(bl_ctx *ctx, size_t bytes);
BL_API void bl_host_free(bl_ctx *ctx, void *p);
/* Cap on scratch HBM (bytes) used for per-sample records; default four fifths of the device's memory (MI355X: 230 GB of 288). */
BL_API int bl_set_scratch_limit(bl_ctx *ctx, uint64_t bytes);
/* on != 0: when a render needs several chunks, run the geodesic kernel of chunk c + 1 on a second stream
 * beside the shading kernels of chunk c (two scratch sets of half the budget). Off by default: measured
 * on MI355X it changes the frame time by less than 1 %, and per-kernel times are cleaner off. */
BL_API int bl_set_overlap(bl_ctx *ctx, int on);
BL_API int bl_render(bl_ctx *ctx, const bl_render_desc *d);
/* Diagnostics (not part of the reference's interface): apply one device math function element-wise to
 * host arrays x (and y for two-argument functions; may be NULL otherwise) and return the results in out.
 * op: 0 exp, 1 expm1, 2 log, 3 cbrt, 4 sin, 5 cos, 6 acos, 7 atan, 8 atan2(x, y), 9 pow(x, y),
 * 10 hypot (blmath.h), 11 bl_hypot_g, 12 bl_sqrt_g, 13 bl_div_g(x, y) (bl_geometry.h), 14 sqrt, 15 x / y,
 * 16 sincos -> sin, 17 sincos -> cos; 38 bl_pow_neg_fifth (the step controller's x^(-1/5), blmath.h). Used by the tests to compare the
 * device arithmetic with the host's. */
BL_API int bl_debug_math(bl_ctx *ctx, int op, int64_t n, const double *x, const double *y, double *out);
/* Tolerant tier, tests only: relative half-width of the band around an active cell cut threshold inside which the cut
 * decision of a sample is left to the exact kernel (default 1e-9; the tolerant arithmetic is good to ~1e-13). A wide band
 * defers many samples, which exercises the list and its overflow path. ops 20-27 of bl_debug_math are the tolerant
 * tier's exp, expm1, cbrt, reciprocal, reciprocal square root and K_0, K_1, K_2. */
BL_API int bl_debug_set_guard_band(bl_ctx *ctx, double relative_width);
/* The measurement switches of this context (BL_SWITCH_*, above) set by the program instead of the environment bl_init read:
 * what tests and A/B tools use between two renders of one context. */
BL_API int bl_debug_set_switches(bl_ctx *ctx, uint32_t switches);
BL_API int bl_get_stats(const bl_ctx *ctx, bl_stats *out);
/* Text of the last failure on this context ("Error: ...\n"), or "" */
BL_API const char *bl_last_error(const bl_ctx *ctx);
/* Text of the last failure of a call that has no context (bl_init) */
BL_API const char *bl_last_global_error(void);
/* Warnings raised by the last call, newline separated, reference wording ("Warning: ...\n") */
BL_API const char *bl_warnings(const bl_ctx *ctx);
BL_API void bl_warnings_clear(bl_ctx *ctx);   /* forget the warnings collected so far */
BL_API void bl_free(bl_ctx *ctx);

/* ------------------------------------------------------------------ host steps between / after renders
 * Kept on the host exactly like the reference does (tiny, serial per block). */
#define BL_MAX_LEVELS 16

/* RadiationIntegrator::CheckAdaptiveRefinement + EvaluateBlock (radiation_adaptive.cpp:19-312) for the
 * level just rendered, followed by the block bookkeeping of GeodesicIntegrator::AugmentCamera
 * (camera.cpp:445-458): which blocks refine, and the (block_v, block_u) list of the next level in the
 * reference's order (parents in order, children (2v,2u), (2v,2u+1), (2v+1,2u), (2v+1,2u+1)).
 *   block_locs    [n_blocks][2] of this level; NULL at level 0 (root blocks, row-major)
 *   image         host [n_q][n_pix of this level] as filled by bl_render
 *   refine_flags  out [n_blocks]
 *   next_locs     out [4 * n_refined][2], may be NULL
 * Returns BL_OK; *n_refined = 0 means the adaptive loop is complete. */
BL_API int bl_adaptive_refine(const bl_ctx *ctx, int level, int n_blocks, const int32_t *block_locs,
                              const double *image, uint8_t *refine_flags, int32_t *n_refined,
                              int32_t *next_locs);

/* OutputWriter::Write (output_writer.cpp:169-274): npz / npy / raw exactly in the reference's layout
 * (numpy_format.cpp, zip_format.cpp, raw_format.cpp), format and array selection from the parameters. */
typedef struct bl_output_level {
  int32_t n_blocks;            /* levels > 0 */
  const int32_t *block_locs;   /* levels > 0: [n_blocks][2] */
  const double *image;         /* host [n_q][n_pix of the level] */
  const double *camera;        /* output_camera: positions (plane) or directions (pinhole) [n_pix][4] */
  const double *render;        /* render_num_images > 0: host [render_num_images][3][n_pix of the level] */
} bl_output_level;
typedef struct bl_output_desc {
  int32_t adaptive_num_levels; /* levels beyond the root that were rendered */
  bl_output_level level[BL_MAX_LEVELS + 1];
  int32_t snapshot;            /* run index, for output_file with {Nd} when simulation_multiple */
} bl_output_desc;
BL_API int bl_write_output(bl_ctx *ctx, const char *path_override, const bl_output_desc *d);

/* Library self-description: "gfx950;hip" etc. Lets a loader verify the HIP path is the one built. */
BL_API const char *bl_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* BLACKLIGHT_AMD_H_ */

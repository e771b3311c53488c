// bl_device.h - data layout shared by the HIP kernels (bl_geodesic.hip, bl_shade.hip, bl_shade_fast.hip, bl_coefficients_freq.hip, bl_transfer.hip, bl_polarized.hip) and their host driver
// (bl_api.hip).
//
// HBM layout of one render chunk (all sizes for C rays, ray_max_steps = S, n_nu frequencies):
//
//   grid cells    float  [n_k][n_j][n_i][8]      32 B per cell: rho, pgas, uu1, uu2, uu3, bb1, bb2,
//                                                bb3 interleaved, i fastest, so the 8 corners of a
//                                                trilinear fetch are four 64-B segments
//   face / centre double x{1,2,3}f[n+1], x{1,2,3}v[n]  + per-axis bucket tables for the cell search
//   records       BlSampleHot, BlSampleCold [<= C*S]  2 x 32 B per emitted sample (position + id | momentum +
//                                                length), written by the geodesic kernel in wave-contiguous
//                                                runs; the locate kernel reads the first, the coefficient
//                                                kernel both (coalesced)
//   located       BlLocated [<= C*S] + tag [<= C*S]   32 + 8 B per sample: trilinear fractions + azimuth | first
//                                                cell + status, locate kernel -> coefficient kernel
//   transfer      double2 [C][S][n_nu]           (a, b) of the per-sample affine update
//                                                I <- a * (I + b), written by the shading kernel,
//                                                replayed far -> near by the transfer kernel
//   per ray       k_t, 1/nu_local, sample_num, flags, output index
#ifndef BLACKLIGHT_AMD_BL_DEVICE_H_
#define BLACKLIGHT_AMD_BL_DEVICE_H_

#include <stdint.h>

#include "bl_camera.h"
#include "bl_geometry.h"

// One emitted geodesic sample (geodesic_pos/dir/len entries of the reference, geodesics.cpp:250-293), before the
// per-sample momentum renormalisation of :352-371, in two 32-byte halves kept in two arrays: the locate kernel
// reads positions and ids only, and with the halves interleaved in one 64-byte record it fetched the whole line
// for half of it (24 GB instead of 12 GB per launch of the benchmark).
struct alignas(16) BlSampleHot {
  double x, y, z;     // position (CKS)
  uint32_t ray;       // chunk-local ray slot, 0xFFFFFFFF = dead slot (sample dropped by truncation)
  uint32_t n;         // sample index along the ray in integration order
};
struct alignas(16) BlSampleCold {
  double kx, ky, kz;  // covariant spatial momentum, not yet renormalised
  double len;         // affine step length as stored by the integrator (negative: camera -> source)
};
static_assert(sizeof(BlSampleHot) == 32 && sizeof(BlSampleCold) == 32, "record halves must be 32 bytes");

// One located sample, written by the locate kernel at the index of its sample record and read once by the
// coefficient kernel (simulation mode): where the sample sits on the grid. 32 bytes of fractions and azimuth
// (BlLocated) plus an 8-byte tag in its own array: bits 0..31 linear index of cell (k_m, j_m, i_m) resp. of the
// nearest cell, bits 32..39 SampleStatus, bits 40.. time slice (slow light). r^2, which the coefficient kernel
// needs again, is not handed over: with zero spin it is x^2 + y^2 + z^2, otherwise one hypot.
struct alignas(16) BlLocated {
  double f_i, f_j, f_k;   // trilinear fractions (kSampleInterp)
  double ph;              // unwrapped spherical Kerr-Schild azimuth (needed again by the Jacobian)
};
static_assert(sizeof(BlLocated) == 32, "located sample must be 32 bytes");

#define BL_DEAD_RAY 0xFFFFFFFFu
// Record slots a wave of the geodesic kernel reserves at a time (one global atomic per block); the
// record buffers hold one spare block per launched wave on top of chunk_rays * ray_max_steps.
#define BL_RECORD_BLOCK 1024
// Marker for "optically thick: I <- b" in the transfer record (exp(-dtau) is never negative)
#define BL_THICK_MARK (-1.0)
// ... and in the tolerant tier's affine records (a, c) of I <- a I + c: a = -0.0. An optically thick step's intensity REPLACES what lies
// behind it (unpolarized.cpp:103-104), a NaN included, so the transfer kernels must tell it from a thin step whose a = 1 + expm1(-tau)
// has rounded to +0 (tau > 37.4), behind which a NaN stays a NaN as in the exact tier (NaN x exp(-tau)). No other a is negative.
#define BL_AFFINE_THICK (-0.0)
#define BL_IS_AFFINE_THICK(a) (__double2hiint(a) < 0)
// A row of composed maps (BlShadeArgs::composed) that stands for per-sample records instead: a = -(number of samples) <= -1 (no
// map has such an a: a composed a lies in [0, 1], the thick mark is -0.0), c = the bit pattern of the first sample's record index;
// the transfer kernel replays transfer[first ... first + number) in its place.
#define BL_COMPOSED_EXPANDED(a) ((a) <= -1.0)

enum BlCounter {
  BL_CNT_NEXT_RAY = 0,      // work queue head of the geodesic kernel
  BL_CNT_RECORDS = 1,       // sample records allocated
  BL_CNT_GATHERS = 2,       // samples that read the grid (S_in)
  BL_CNT_OVERFLOW = 3,      // record buffer overflow flag
  BL_CNT_UNDEFINED = 4,     // inter-block interpolation: samples at an upper edge of the last MeshBlock
  BL_CNT_INTERP_FAILED = 5, // inter-block interpolation: samples for which no anchor block exists
  BL_CNT_REDO = 6,          // tolerant tier: samples left to the exact coefficient kernel (redo list entries)
  BL_CNT_COMMITTED = 7,     // geodesic kernel: record slots spoken for - ray_max_steps per ray in flight, what it emitted per finished ray
  BL_CNT_SAMPLES = 8,       // kept samples of the finished rays: where the next ray's per-sample rows start (BlTraceArgs::ray_offset)
  BL_CNT_PARKED = 9,        // rays bl_geodesic_kernel parked for bl_geodesic_quad_kernel (BlTraceArgs::parked)
  BL_CNT_COUNT = 10
};
// per scratch set: the counters above, four transfer statistics, eight debug counters (kernels built with -DBL_GEO_STATS)
#define BL_CNT_DEBUG (BL_CNT_COUNT + 4)
#define BL_CNT_QUAD_NEXT (BL_CNT_COUNT + 12)   // work queue head of the launch that finishes the parked rays: parked rays handed out
#define BL_CNT_PARKED_YOUNG (BL_CNT_COUNT + 14)   // parked rays with fewer than BlTraceArgs::park_age samples so far (BL_CNT_PARKED: the others)
#define BL_CNT_TOTAL (BL_CNT_COUNT + 15)

// dynamic LDS the refined instantiation of bl_shade_fused2_kernel may take (one 512-lane workgroup to a compute unit of 160 KiB)
#define BL_FUSED_REFINED_LDS (150 * 1024)
// ... and the tables of bl_locate_kernel<kRefined> (one 1 024-lane workgroup to a compute unit; 16 KiB more for its waves' lists)
#define BL_LOCATE_REFINED_LDS (136 * 1024)
// ... and of the exact second pass behind the fused kernel (256-lane workgroups, two to a compute unit, 8 KiB of anchor rows beside)
#define BL_REDO_TABLES_LDS (48 * 1024)

struct BlGridDevice {
  const float *cells;        // [n_k][n_j][n_i][8]
  const float *kappa;        // [n_k][n_j][n_i] electron entropy (plasma_model = code_kappa), else null
  const double *xf[3];       // faces  (r, theta, phi)
  const double *xv[3];       // centres
  const unsigned short *bucket[3];   // bucket -> first candidate cell
  double bucket_x0[3], bucket_inv_w[3];
  int n_bucket[3];
  // axes whose faces are evenly spaced to 1e-4 of a cell (bit a of uniform_mask): cell = floor((x - cell_x0) * cell_inv_w), which the
  // tolerant locate step takes as its guess and checks against the faces (locate_plain_from_angles)
  int uniform_mask;
  double cell_x0[3], cell_inv_w[3];
  // what bl_shade_fused2_kernel (bl_shade_fused.hip) asks of a grid: radial faces evenly spaced in log r to 1e-4 of a cell
  // (log_uniform; cell = floor((log2 r - log_l0) * log_inv_w), a guess the faces confirm), theta and phi covering the sphere
  // (full_sphere: no sample is off the grid in angle), and the first and last radial face
  int log_uniform, full_sphere;
  float log_l0, log_inv_w;
  double r_face_in, r_face_out;
  int n[3];                  // n_i, n_j, n_k of the (merged) global grid
  int nb[3];                 // cells per block along each axis (= n for a single block)
  int stride_row, stride_plane;   // cells between j- and k-neighbours in `cells` / `kappa`
  // Mesh refinement (blocks of several levels, simulation_sampling.cpp:352-394): `cells` is then
  // [block][n_k][n_j][n_i][8] with nb = block size, and the tables above are unused. n_blocks = 0: merged grid.
  int n_blocks;
  const double *edge[3];     // ascending distinct block boundaries per axis, n_edge[a] + 1 values
  int n_edge[3];
  const int *lattice;        // [n_edge[2]][n_edge[1]][n_edge[0]] -> block covering that box, -1: none
  // Coordinate rows of the blocks, each distinct row once (blocks at one level and position along an axis share theirs: a 256^3 mesh in
  // 64^3 blocks has 4 - 6 rows per axis, 36 - 64 blocks): small enough for the locate kernel to stage in LDS (refined_lds_bytes)
  const double *bxf[3];      // [n_rows[a]][nb[a] + 1] faces
  const double *bxv[3];      // [n_rows[a]][nb[a]] centres
  const int *block_row[3];   // [n_blocks]: a block's row along each axis
  const double *xv_next[3];  // [n_blocks]: first centre of the NEXT block of the file along the axis (what the reference's Array holds behind a
                             // block's last centre, simulation_sampling.cpp:520-522); unused for the last block
  const double *row_guess[3];   // [n_rows[a]][3]: where in a row to start the search for a coordinate s - kind (0: faces evenly spaced, 1: evenly
                                // spaced in log s), origin (the first face, or its log2), cells per unit (of s, or of log2 s); any start gives
                                // the same cell, a good one saves the walk
  int n_rows[3];
  double edge_first[3], edge_last[3];   // edge[a][0], edge[a][n_edge[a]]
  double box_guess[3][3];    // the same three numbers for the block boundaries along each axis
  int refined_lds_bytes;     // > 0: edges, lattice, rows, block table and hash fit the locate kernel's LDS budget (bl_locate_kernel<kRefined>)
  // The locate step inside bl_shade_fused2_kernel<..., kRefined> (bl_shade_fused.hip): a mesh whose boxes are evenly spaced in log r, theta
  // and phi, cover the sphere, and whose rows are evenly spaced likewise. Per box one descriptor: byte offset of the block's cells, LDS
  // offsets of the block's three row chunks (16-byte header with the row's cell guess, then 64 bytes per cell).
  const unsigned int *fused_desc;   // [n_edge[2]][n_edge[1]][n_edge[0]][4]
  int fused_lds_bytes;              // > 0: the kernel takes this mesh (cut table + row chunks + descriptors)
  float box_l0, box_linv;           // box along r = floor((log2 r - box_l0) * box_linv)
  double box_x0[2], box_inv_w[2];   // along theta and phi = floor((x - box_x0) * box_inv_w)
  // Inter-block interpolation (simulation_block_interp; simulation_sampling.cpp:505-546, :1068-1321): the MeshBlock
  // table and a hash from (level, location) to block - the reference scans all blocks for every such lookup
  int block_interp;
  const int *levels;             // [n_blocks]
  const int *locations;          // [n_blocks][3]
  const unsigned long long *hash_keys;   // open addressing, hash_mask + 1 slots, ~0ull = empty
  const int *hash_blocks;
  unsigned int hash_mask;
  int max_level, n_3_level0;     // n_3_level(level) = n_3_level0 << level (:84-93)
  // FMKS grids (simulation_coord = fmks; simulation_sampling.cpp:190-198, :396-456): one block in native coordinates
  // (tables above: only x^3 is searched in them), position in x^1, x^2 from the reader's look-up table
  int fmks;
  const double *sks_map;         // [2][sks_map_n2][sks_map_n1]
  int sks_map_n1, sks_map_n2;
  double sks_map_r_in, sks_map_dr, sks_map_dtheta;
  double fmks_bounds[6];         // r, theta, phi limits of the grid
  double fmks_x1_0, fmks_dx1, fmks_dx2;
};

struct BlPlasmaDevice {
  // units (simulation_coefficients.cpp:237-239)
  double d_unit, e_unit, b_unit;
  double plasma_mu, plasma_ne_ni, plasma_rat_low, plasma_rat_high, plasma_thermal_frac;
  int plasma_use_p;
  int simulation_interp;
  int simulation_coord;
  int fallback_nan;
  int any_cell_cut;          // some cell cut threshold (simulation_coefficients.cpp:361-375) is >= 0
  // power-law electrons (simulation_coefficients.cpp:54-66, :556-584); power_frac = 0: none
  double power_frac, plasma_p, power_jj, power_aa;
  int kappa_frac_zero;       // plasma_kappa_frac == 0 (BlShadeCold::kappa.frac lives in HBM; the launchers choose instantiations by this)
  int code_kappa;            // plasma_model = code_kappa: theta_e from the simulation's electron entropy (:351-358)
  int cut_mask;              // bit c set: cell cut threshold c (BlShadeCold::fast_cut order) is active
  int kappa_unpolarized;     // kappa-distribution electrons in an unpolarized run (BL_UNDEFINED_KAPPA): BlShadeCold::kappa's intensity terms
};

// Kappa-distribution electrons (simulation_coefficients.cpp:82-193; the reference's names without the prefix).
// frac = 0: none. The reference defines every constant in polarized runs only: an unpolarized run reads aa_high_i without ever
// setting it, and is rendered here only under bl_set_undefined_policy(BL_UNDEFINED_KAPPA), with the polarized definition.
struct BlKappaDevice {
  double frac, kappa, w;
  double jj_low, jj_high, jj_x_i, aa_low, aa_high, aa_x_i;
  double jj_low_q, jj_low_v, jj_high_q, jj_high_v, jj_x_q, jj_x_v;
  double aa_low_q, aa_low_v, aa_high_i, aa_high_q, aa_high_v, aa_x_q, aa_x_v;
  double rho_v, rho_frac;
  double rho_q_low[5], rho_q_high[5];   // a, b, c, d, e of the two fits that bracket kappa
  double rho_v_low[2], rho_v_high[2];   // a, b
};

// Rarely used parameters of the shading kernel (optional geometric cuts, cell cut thresholds,
// fallback primitives): kept in HBM behind one pointer and read under wave-uniform flags, so they do
// not occupy SGPRs in the common case where they are all disabled.
struct BlShadeCold {
  int omit_near, omit_far, plane;
  double omit_in, omit_out, midplane_theta, midplane_z;
  double plane_origin[3], plane_normal[3];
  double cam_x[4];
  double cut_rho_min, cut_rho_max, cut_n_e_min, cut_n_e_max, cut_p_gas_min, cut_p_gas_max;
  double cut_theta_e_min, cut_theta_e_max, cut_b_min, cut_b_max, cut_sigma_min, cut_sigma_max;
  double cut_beta_inverse_min, cut_beta_inverse_max;
  float fallback_rho, fallback_pgas, fallback_kappa;
  double plasma_gamma, plasma_gamma_i, plasma_gamma_e;
  BlKappaDevice kappa;
  // Tolerant tier: the active cell cut thresholds in the order rho, n_e, p_gas, theta_e, B, sigma, 1 / beta, each
  // (min, max), and the guard band around each (threshold * (1 -+ 1e-9)) inside which the decision is left to the
  // exact kernel
  double fast_cut[14], fast_cut_lo[14], fast_cut_hi[14];
};

struct BlFormulaDevice {
  double r0, h, l0, q, nup, cn0, alpha, a, beta;
};

struct BlCutsDevice {
  int any_optional;          // any cut besides r > camera_r is active (details in BlShadeCold)
  double camera_r;
};

// Per-sample inputs of the auxiliary images (unpolarized.cpp:113-173), stored [ray][n] like the transfer
// records. Only written and read when an auxiliary image is requested; in that mode the transfer buffer
// holds (j_nu, alpha_nu) per sample and frequency instead of (a, b).
#define BL_NUM_CELL_VALUES 7   // blacklight.hpp:30-33: rho, n_e, p_gas, theta_e, bb, sigma, beta_inv
struct alignas(16) BlAuxSample {
  double delta_lambda;               // sample_len in geometric units
  double length_term;                // sqrt(dl_dlambda_sq) * delta_lambda * x_unit (image_length)
  double t;                          // coordinate time of the sample (image_time)
  double plane;                      // camera_x . x (image_crossings compares its sign between samples)
  double cell[BL_NUM_CELL_VALUES];   // cell_values, NaN when not recorded
  double pad;
};
static_assert(sizeof(BlAuxSample) == 96, "aux sample must be 96 bytes");

// Which auxiliary images are requested, and where their rows start (radiation_integrator.cpp:436-520)
struct BlAuxImages {
  int any;
  int image_light, image_time, image_length, image_lambda, image_emission, image_tau;
  int image_lambda_ave, image_emission_ave, image_tau_int, image_crossings;
  int offset_time, offset_length, offset_lambda, offset_emission, offset_tau;
  int offset_lambda_ave, offset_emission_ave, offset_tau_int, offset_crossings;
  int n_q;
  int polarized;   // rows 4 l + (I, Q, U, V) written by the polarized transfer kernel instead of row l
  int polarized_rows_only;   // polarized run whose only other row is tau (or none), no rendering: the polarized transfer kernel
                             // accumulates tau beside the Stokes parameters, the auxiliary kernel keeps its bookkeeping only
                             // and the coefficient kernel writes no BlAuxSample
};

// False-colour rendering parameters (rendering.cpp; a device buffer, too large for kernel arguments)
struct BlRenderDevice {
  int n_images;
  int fill_present;
  int n_features[BL_MAX_RENDER_IMAGES];
  int quantity[BL_MAX_RENDER_IMAGES][BL_MAX_RENDER_FEATURES];
  int type[BL_MAX_RENDER_IMAGES][BL_MAX_RENDER_FEATURES];
  double min_val[BL_MAX_RENDER_IMAGES][BL_MAX_RENDER_FEATURES], max_val[BL_MAX_RENDER_IMAGES][BL_MAX_RENDER_FEATURES];
  double thresh[BL_MAX_RENDER_IMAGES][BL_MAX_RENDER_FEATURES], tau_scale[BL_MAX_RENDER_IMAGES][BL_MAX_RENDER_FEATURES];
  double opacity[BL_MAX_RENDER_IMAGES][BL_MAX_RENDER_FEATURES];
  double xyz[BL_MAX_RENDER_IMAGES][BL_MAX_RENDER_FEATURES][3];
};

// Kernel arguments: geodesic kernel
struct BlTraceArgs {
  BlSpacetime st;
  BlCameraDevice cam;
  double r_terminate, r_horizon, camera_r;
  double ray_step, ray_tol_abs, ray_tol_rel;
  int ray_max_steps, ray_max_retries;
  long long chunk_begin;      // first traversal index of this chunk
  int chunk_rays;
  long long n_rays_total;
  int swizzle_tiles;          // >0: traversal order walks 8x8 pixel tiles of a swizzle_tiles-wide image
  const int *tile_order;      // device [tiles of the image] or null: order in which the 8x8 tiles are traced
  const int *pixel_map;       // device, or null
  const int *block_locs;      // device, or null
  BlSampleHot *records_hot;
  BlSampleCold *records_cold;
  int record_stride;          // 1: the two halves of a record in arrays of their own; 2: interleaved, 64 bytes per record (records_cold = records_hot + 1)
  double *sample_t;           // optional [record capacity]: coordinate time of each sample (image_time)
  long long record_capacity;
  // The geodesic kernel hands a ray to a lane only while BL_CNT_COMMITTED + ray_max_steps <= record_gate (= capacity minus one
  // block per wave): a chunk is the rays that fit the record buffers as they turn out, not as the worst case would have them.
  // Rays it did not get to (BL_CNT_NEXT_RAY < chunk_rays at the end) are the next chunk's.
  long long record_gate;
  unsigned long long *counters;
  double *ray_kt, *ray_factor;
  int *ray_sample_num;
  unsigned char *ray_flags;
  long long *ray_out_index;
  long long *ray_offset;      // [chunk_rays]: first row of the ray's kept samples in the per-sample arrays (transfer records, ...)
  double *camera_pos, *camera_dir;  // optional [n_rays][4] indexed by output index
  // Start state of every ray of the chunk, written by bl_ray_init_kernel (one ray per lane, every lane busy) and read by the
  // persistent geodesic kernel when it hands a ray to an idle lane: BL_RAY_START_FIELDS rows of ray_start_stride doubles,
  // [field][chunk slot] - t, x, y, z, k_x, k_y, k_z | k_t | r | first-stage derivatives (Dormand-Prince only)
  double *ray_start;
  long long ray_start_stride;
  // Steps in the empty shell between the grid's outer edge and the camera's sphere leave no records (plain images of a spherical
  // Kerr-Schild simulation with fallback values: every sample there is off the grid, has no field and adds nothing, simulation_
  // sampling.cpp:352-394, simulation_coefficients.cpp:394). A step is skipped when the bound |p - y| <= sum_q |r_q| on its dense
  // output puts all its samples there: |y| - d > skip_low (= sqrt(r_out^2 (1 + 1e-9) + a^2); r_KS^2 >= R^2 - a^2) and
  // |y| + d < skip_high (= camera_r (1 - 1e-9); r_KS <= R). skip_low = +inf: every step is recorded. The samples still count
  // (sample_num, ray_max_steps); ray_skipped[slot] is how many of a ray's samples have no record.
  double skip_low, skip_high;
  int *ray_skipped;
  // Composed transfer maps (tolerant tier, bl_shade_fused2_kernel with one map per segment): records carry segment numbers instead
  // of sample numbers, ray_offset counts segments, ray_rows[slot] = the ray's segments (bl_geodesic_kernel's emission loop)
  int segment_rows;
  int *ray_rows;
  // Parked rays (Dormand-Prince, no sample times, no skipped shell; BL_SWITCH_QUAD_TAIL - a measured experiment, off by default).
  // A ray advances one step per pass of its wave through ~6 000 instructions however few of the wave's lanes still hold one, so
  // the last rays of a chunk - a wave with nothing left to refill its idle lanes from - set the kernel's time. Such a wave, once
  // park_below or fewer of its lanes hold a ray, writes those rays to parked[] (BL_PARK_DOUBLES doubles each, below; BL_CNT_PARKED
  // counts them) and ends; bl_geodesic_quad_kernel, launched behind it, finishes them with a ray per QUAD of lanes in two thirds
  // of the instructions per step - the same operations on the same operands, so the same bits (bl_geodesic_quad.hip: what it
  // gains and costs). park_always: every wave parks every ray it is handed before its first step (tests: the whole frame
  // goes through the quad kernel). parked == nullptr: no ray is parked.
  double *parked;
  int park_capacity;
  int park_below;
  int park_after;    // ... or, however many lanes hold one, park_after passes of the wave after it first had nothing to refill a lane from
  // Parked rays are taken up again oldest first: those with park_age samples or more fill parked[] from the front (BL_CNT_PARKED),
  // the others from the back (BL_CNT_PARKED_YOUNG) - a ray that is long so far is likely to be long, and the launch that finishes
  // them ends with its longest ray, which should not be the last to get a quad.
  int park_quiet;    // ... or park_quiet passes without a ray of the wave finishing (its rays are long ones)
  int park_age;
  int quad_first_round;   // bl_geodesic_quad_kernel: its first this many waves (one per SIMD) run at raised priority
  int park_always;
  // Rays predicted long, on compute units of their own from the first moment (BL_TAIL_SPLIT: docs/notebook.md section
  // 5k). bl_split_long_kernel parks - before either stepper starts - the rays of a plane camera whose impact parameter lies in
  // [split_b_lo, split_b_hi] (in units of M: the band around the photon ring's critical curve) and marks their start state (r < 0);
  // bl_geodesic_kernel passes a marked ray over, bl_geodesic_quad_kernel steps the parked ones on a stream whose CU mask the
  // other stream's excludes. split_b_hi = 0: no split.
  double split_b_lo, split_b_hi;
};
#define BL_RAY_START_FIELDS 17
// A parked ray (BlTraceArgs::parked): everything bl_geodesic_kernel holds of a ray between two steps, BL_PARK_DOUBLES doubles
//   0..7 the state t, x, y, z, k_x, k_y, k_z, s | 8..15 the first stage of the next step (FSAL) | 16 k_t | 17 the next step's
//   length | 18 r at the state | 19 r of the last sample | 20 (chunk slot, samples so far) | 21 (retries, index of the
//   truncating sample or -1) | 22 (segments so far, bit 0 previous step failed + bit 1 flagged) | 23 unused
#define BL_PARK_DOUBLES 24

// Kernel arguments: shading kernel
// Polarized transfer (image_polarization): what the polarized transfer kernel needs of every sample besides the
// eight coefficients - position (for the metric and the connection), affine length, k^mu = g^{mu nu} k_nu with the
// renormalised momentum, and rows 1 and 2 of the fluid tetrad (the only rows the Stokes projection and its inverse
// read: polarized.cpp:268-292, :793-813). The coefficient kernel has all of it at hand (simulation_coefficients.cpp:
// 398-431 builds the same tetrad from the same inputs as polarized.cpp:201-265). 128 bytes.
// Tolerant tier, per sample: the transport matrix (3 x 3 block of I, Q, U by rows, then the V number), the sample's length, a spare
#define BL_POL_MATRIX_DOUBLES 12
struct alignas(16) BlPolSample {
  double x[3];
  double delta_lambda;
  double kcon[4];
  double e1[4], e2[4];
};

// Tolerant tier with several frequencies (BlShadeArgs::freq_split): what is left of a sample once everything that does not
// depend on the frequency has been evaluated - the factors of bl_shade_fast_kernel's frequency loop. bl_transfer_freq_kernel
// (one lane per ray and frequency) turns them into the sample's (a, c) and applies it on the spot: no transfer records.
// flag: 0 nothing to add (cut sample, cut cell, no field), 1 coefficients follow, 2 NaN (off the grid with fallback_nan).
// A sample whose cut decision is deferred gets its factors from the exact second pass (bl_shade_kernel<..., kRedo>). 64 bytes.
struct alignas(16) BlFreqInputs {
  double flag;
  double s_1_2, s_1_3, s_1_6;   // x^(1/2), x^(1/3), x^(1/6) of x = nu / nu_s at unit frequency
  double s_planck;              // h nu / (k T_e) at unit frequency
  double s_j;                   // j_nu nu^2 e^(x^(1/3)) / var_c^2
  double s_length;              // delta lambda in cm at unit frequency
  double pad;
};

// Polarized runs: what the per-frequency coefficient formulas need of a sample (simulation_coefficients.cpp:458-698),
// left by the coefficient kernel for bl_polarized_coefficients_kernel, one per sample record. 64 bytes.
struct alignas(16) BlCoefInputs {
  double nu_fluid_over_nu;   // -k_mu u^mu
  double n_e_cgs, nu_c_cgs, theta_e, kb_tt_e_cgs;
  double cos2_theta_b;       // min(cos^2, 1) (:449-452)
  double cos_sign;           // +-1: sign of k.b in the fluid frame (:455)
  double have_coefficients;  // 1: the sample has coefficients; 0: j = alpha = rho = 0 at every frequency
};

// Slow light (slow_light_on): the time slices the reader holds (simulation_reader.cpp:211-303), latest
// first, all on the geometry of BlGridDevice. n = 0: off.
struct BlSlowDevice {
  int n;                           // slow_chunk_size
  int interp;                      // slow_interp: linear in time between two slices, else the nearest slice
  double snapshot_time;            // camera time of this image
  const double *times;             // device [n], descending
  const float *const *cells;       // device [n]: cell array of every slice (layout of BlGridDevice::cells)
  const float *const *kappa;       // device [n] or null
  unsigned int *ray_extrap;        // [chunk rays]: bit e set = some sample needed extrapolation of kind e
  unsigned long long *extrap_max;  // [4]: bit pattern of the largest extrapolation of each kind (doubles >= 0)
  double *frac;                    // [record capacity]: t_frac of a located sample
};

struct BlShadeArgs {
  BlSpacetime st;
  BlCutsDevice cuts;
  BlPlasmaDevice plasma;
  BlFormulaDevice formula;
  BlGridDevice grid;
  BlSlowDevice slow;
  const BlShadeCold *cold;    // device pointer
  const BlSampleHot *records_hot;
  const BlSampleCold *records_cold;
  int record_stride;          // as in BlTraceArgs
  BlLocated *located;         // [record capacity], simulation mode
  unsigned long long *located_tag;   // [record capacity]: cell | status << 32 | time slice << 40
  BlFreqInputs *freq_inputs;         // [sample row] when freq_split
  int coef_split;                    // exact tier, plain images, n_nu >= 4: BlCoefInputs for bl_coefficients_freq_kernel instead of the frequency loop
  int freq_split;                    // tolerant tier, n_nu >= 4: per-sample factors instead of per-frequency transfer records
  int tag_in_record;                 // tolerant tier: the tag is written into BlLocated::ph instead (32 bytes per sample, one stream)
  int lds_table_bytes;        // size of the coordinate tables the locate kernel stages in LDS; 0: searched in HBM
  int samples_renormalised;   // records come from a geodesic checkpoint: momenta as stored, no renormalisation per sample
  int tolerant;               // bl_set_arithmetic(BL_ARITH_TOLERANT): kernels that have a tolerant instantiation use it
  // Composed transfer maps (bl_shade_fused2_kernel): one (a, c) per SEGMENT of a ray (BlTraceArgs::segment_rows) in row
  // ray_offset[ray] + segment of `composed`; `transfer` is then indexed by RECORD and written only for the samples of a wave that
  // holds a deferred sample or an optically thick step (the segment's row then says where they are: BL_COMPOSED_EXPANDED)
  double2 *composed;          // [segment row], or null: one transfer record per sample
  int general_locate;         // measurement switch (bl_stats.switches): the general locate kernel where the plain one applies; refined tables in HBM
  int undefined_edge;         // bl_set_undefined_policy(BL_UNDEFINED_EDGE): samples where the reference reads past its arrays use the edge
  const unsigned long long *counters_in;
  unsigned long long *counters;
  const double *ray_kt, *ray_factor;
  const long long *ray_offset;   // [chunk_rays]: sample n of ray q has row ray_offset[q] + n in transfer, aux, pol_samples, freq_inputs, ...
  const double *frequencies;  // device [n_nu]
  int n_nu;
  int ray_max_steps;
  double x_unit;              // GM/c^2 in cm (unpolarized.cpp:42)
  double2 *transfer;          // [sample row][n_nu]; (a, b), or (j, alpha) in auxiliary mode
  double *tau_inc;            // tolerant tier with an optical-depth image: [sample row][n_nu] alpha x length of every sample, or null
  // auxiliary-image mode only
  BlAuxSample *aux;           // [sample row]
  const double *sample_t;     // [record capacity] or null
  const unsigned char *ray_flags;
  // polarized transfer only (runs in auxiliary-image mode): null otherwise
  BlPolSample *pol_samples;   // [sample row]
  double2 *pol_coeffs;        // [sample row][n_nu][4]: (j_I, alpha_I), (j_Q, j_V), (alpha_Q, alpha_V), (rho_Q, rho_V) - one 64-byte record, half a
                              // cache line, for the lane of the sequential kernels that walks the ray (polarized runs have no `transfer` array)
  BlCoefInputs *coef_inputs;  // [record capacity]: coefficient kernel -> polarized coefficient kernel
  unsigned char *have_flags;  // [record capacity] or null: bl_shade_polarized2_kernel<..., kCoefficients> evaluates the coefficients itself and
                              // says here which records have them (bl_polarized_frame_kernel: the others' frames)
  unsigned int *anchors;      // inter-block interpolation: [record capacity][8] cells of the eight anchors, else null
  double power_pol[7];        // simulation_coefficients.cpp:67-80: jj_q, jj_v, aa_q, aa_v, rho, rho_q, rho_v
  double plasma_gamma_min;
  // tolerant arithmetic tier (bl_shade_fast_kernel) only
  // fast_shade_sample's constants, folded on the host (BuildShadeArgs): [0] k T_e = [0] p / rho x D / ([1] + [2] / beta^2 + [3] D) in code
  // units of p and rho; [4] x = nu / nu_s at unit frequency = s_nu / (|b| sin) / (k T_e)^2 x [4]; [5] j at unit frequency =
  // [5] rho |b| sin / s_nu^2; [6] n_e = [6] rho; [7] nu_c / |b|
  double fast_k[8];
  double fast_angle_band;     // tolerant locate step: theta / phi closer than this to a decision are the exact kernel's (1e-12; wider under bl_debug_set_guard_band)
  unsigned long long *redo_list;       // record indices left to the exact kernel, BL_CNT_REDO entries
  unsigned long long redo_capacity;    // entries the list holds; more than that: the exact kernel shades every record
  int aux_need_coefficients;  // image_light || image_emission || image_tau || image_emission_ave || image_tau_int (:389)
  int aux_need_length;
  int aux_record_unused;      // BlAuxImages::polarized_rows_only: nobody reads the BlAuxSample records
  double cam_x[4];
};

// Kernel arguments: transfer kernel
struct BlTransferArgs {
  // polarized transfer only
  const BlPolSample *pol_samples;
  const double2 *pol_coeffs;
  double *pol_matrix;                      // tolerant tier: [sample row][BL_POL_MATRIX_DOUBLES] (bl_polarized.hip)
  const double *camera_pos, *camera_dir;   // [n_rays_total][4] by output index: initial position, momentum
  BlSpacetime st;
  int simulation_coord, rotation_split;
  double cam_u_con[4], cam_u_cov[4], cam_vert_con_c[4];
  const double2 *transfer;
  int ja_stride;                           // records between the (j, alpha) of consecutive (sample, frequency) pairs as bl_transfer_aux_kernel reads them:
                                           // 1, or 4 in polarized runs, where `transfer` is pol_coeffs
  const double *tau_inc;                   // bl_tau_kernel: [sample row][n_nu], rows summed into image row tau_row + l
  int tau_row;
  const unsigned long long *counters;      // BL_CNT_NEXT_RAY: rays of the chunk the geodesic kernel traced (bl_rays_done)
  const int *ray_sample_num;               // [chunk_rays]: kept samples with a record (= rows) of each ray
  const int *ray_skipped;                  // [chunk_rays] or null: its samples without a record (BlTraceArgs::skip_low)
  const unsigned char *ray_flags;
  const long long *ray_out_index;
  const long long *ray_offset;             // [chunk_rays]: first sample row of each ray
  const double *frequencies;
  int n_nu, ray_max_steps, chunk_rays;
  int fallback_nan, model_type;
  int affine;                 // tolerant tier: records are (a, c) of I <- a I + c instead of (a, b) of I <- a (I + b)
  int lane_transfer;          // measurement switch (bl_stats.switches): one lane per ray where four lanes per ray apply
  const double2 *composed;    // composed transfer maps (BlShadeArgs::composed): [ray_offset + segment], ray_rows[ray] of them per ray; `transfer` by record
  const int *ray_rows;
  const BlFreqInputs *freq_inputs;            // bl_transfer_freq_kernel
  long long n_rays_total;
  double *image;              // [n_q][n_rays_total]; rows 0..n_nu-1 = I_nu
  int *out_sample_num;        // [n_rays_total] or null
  unsigned char *out_flags;   // [n_rays_total] or null
  unsigned long long *stats;  // [0] sum sample_num, [1] flagged rays, [2] max sample_num
  // auxiliary-image mode only
  BlAuxImages aux_images;
  const BlAuxSample *aux;
  const double *ray_factor;
  double x_unit, t_unit;
  const BlRenderDevice *render_params;   // device, or null
  double *render;                        // [n_images][3][n_rays_total], or null
};

#if defined(__HIPCC__) || defined(__HIP__)
// Rays of a chunk the geodesic kernel traced: the rays after them did not fit the record buffers and go to the next chunk
__device__ __forceinline__ int bl_rays_done(const unsigned long long *counters, int chunk_rays) {
  const unsigned long long taken = counters[BL_CNT_NEXT_RAY];
  return taken < (unsigned long long)chunk_rays ? (int)taken : chunk_rays;
}
#endif

#endif  // BLACKLIGHT_AMD_BL_DEVICE_H_

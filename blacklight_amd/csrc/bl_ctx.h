// bl_ctx.h - the context behind the C-ABI handle (include/blacklight_amd.h) and the few helpers the host-side
// translation units of the HIP path share: bl_api.hip (construction, validation, grid upload, settings) and
// bl_render.hip (bl_render: chunk planning, kernel pipeline, geodesic checkpoints, statistics). Internal, hidden visibility.
#ifndef BLACKLIGHT_AMD_BL_CTX_H_
#define BLACKLIGHT_AMD_BL_CTX_H_
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/blacklight_amd.h"
#include "bl_bessel.h"
#include "bl_camera.h"
#include "bl_device.h"
#include "bl_internal.h"

extern "C" hipError_t bl_launch_ray_init(const BlTraceArgs *args, int integrator, hipStream_t stream);
extern "C" hipError_t bl_launch_geodesic(const BlTraceArgs *args, int integrator, int grid, hipStream_t stream, int lds_pad);
extern "C" hipError_t bl_launch_geodesic_quad(const BlTraceArgs *args, int grid, hipStream_t stream, int lds_pad);
extern "C" hipError_t bl_launch_split_long(const BlTraceArgs *args, hipStream_t stream);
extern "C" int bl_geodesic_occupancy(int integrator, int with_time, int spin_zero, int shell);
extern "C" hipError_t bl_launch_locate(const BlShadeArgs *args, int grid, int lds_bytes, hipStream_t stream);
extern "C" hipError_t bl_launch_shade(const BlShadeArgs *args, int model, int grid, hipStream_t stream);
extern "C" hipError_t bl_launch_shade_fast(const BlShadeArgs *args, int grid, hipStream_t stream);
extern "C" int bl_fused2_applicable(const BlGridDevice *grid, int n_nu, long long n_rays);
extern "C" int bl_fused2_refined_applicable(const BlGridDevice *grid, int n_nu, long long n_rays);
extern "C" int bl_polarized2_refined_applicable(const BlGridDevice *grid, long long n_rays);
extern "C" hipError_t bl_launch_transfer_composed(const BlTransferArgs *args, hipStream_t stream);
extern "C" hipError_t bl_launch_shade_exact2(const BlShadeArgs *args, int grid, hipStream_t stream);
extern "C" hipError_t bl_launch_shade_polarized2(const BlShadeArgs *args, int grid, hipStream_t stream);
extern "C" hipError_t bl_launch_shade_formula_fast(const BlShadeArgs *args, int grid, hipStream_t stream);
extern "C" hipError_t bl_launch_polarized_coefficients(const BlShadeArgs *args, int grid, hipStream_t stream);
extern "C" hipError_t bl_launch_polarized_coefficients_parts(const BlShadeArgs *args, int grid, int frames, hipStream_t stream);
extern "C" hipError_t bl_launch_transfer(const BlTransferArgs *args, hipStream_t stream);
extern "C" hipError_t bl_launch_tau(const BlTransferArgs *args, hipStream_t stream);
extern "C" hipError_t bl_launch_transfer_freq(const BlTransferArgs *args, hipStream_t stream);
extern "C" hipError_t bl_launch_coefficients_freq(const BlShadeArgs *args, int grid, hipStream_t stream);
extern "C" hipError_t bl_launch_transfer_aux(const BlTransferArgs *args, hipStream_t stream);
extern "C" hipError_t bl_launch_transfer_polarized(const BlTransferArgs *args, hipStream_t stream);
extern "C" hipError_t bl_launch_transfer_polarized_matrix(const BlTransferArgs *args, int num_cus, hipStream_t stream);
extern "C" hipError_t bl_launch_transport_matrices(const BlTransferArgs *args, int num_cus, hipStream_t stream);
extern "C" hipError_t bl_launch_transfer_polarized_rays(const BlTransferArgs *args, hipStream_t stream);
extern "C" hipError_t bl_launch_debug_math(int op, long long n, const double *x, const double *y, double *out, hipStream_t stream);

namespace blhost {

constexpr double kPi = 3.141592653589793;
constexpr double kC = 2.99792458e10;
constexpr double kGGMsun = 1.32712440018e26;
constexpr double kMp = 1.67262192369e-24;
constexpr double kMe = 9.1093837015e-28;   // (the kernels' values: bl_kernel_util.h)
constexpr double kE = 4.80320425e-10;
constexpr double kSqrt2 = 1.4142135623730951;
constexpr int kNumCellValues = 7;

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

inline void Check(hipError_t err, const char *what) {
  if (err != hipSuccess) throw Failure{BL_E_DEVICE, std::string(what) + ": " + hipGetErrorString(err)};
}

}  // namespace blhost
using namespace blhost;

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
  // BL_SWITCH_SPLIT_LONG: two streams whose CU masks partition the device - split_cus compute units for bl_geodesic_quad_kernel
  // and the rays predicted long, the rest for bl_geodesic_kernel (hipExtStreamCreateWithCUMask)
  hipStream_t stream_few = nullptr, stream_most = nullptr;   // (borrowed from a process-wide table, never destroyed: bl_render.hip)
  int split_cus_made = 0;             // compute units of the quad stepper's stream as it exists (an eighth of the device)
  int split_lds_pad = 39 * 1024;      // LDS a wave of either stepper reserves so that a CU takes four of them, one per SIMD
  bool split_unavailable = false;     // the runtime refused a CU-masked stream: BL_TAIL_AUTO stops asking
  std::vector<hipEvent_t> events;     // kEventsPerChunk per scratch set + begin / end of the render
  unsigned long long *host_counters = nullptr;   // pinned, BL_CNT_TOTAL per scratch set
  uint64_t scratch_limit = 144ull << 30;
  int overlap_chunks = 0;             // bl_set_overlap(): geodesic kernel of chunk c + 1 beside the shading of chunk c
  int arithmetic = BL_ARITH_TOLERANT; // bl_set_arithmetic(); BLACKLIGHT_AMD_ARITHMETIC = exact | tolerant sets what a new context starts with
  int reproducible = 0;               // bl_set_reproducible(): tolerant tier without composed transfer maps
  int tail_policy = BL_TAIL_AUTO;     // bl_set_tail_policy()
  hipStream_t caller_stream = nullptr;   // bl_set_caller_stream(): work queued there before a bl_render call precedes its kernels
  bool caller_stream_set = false;
  hipEvent_t caller_event = nullptr;
  int undefined_policy = BL_UNDEFINED_REFUSE;   // bl_set_undefined_policy(): BL_UNDEFINED_EDGE | BL_UNDEFINED_KAPPA
  bool kappa_warned = false;
  bool debug_counters = false;        // BLACKLIGHT_AMD_DEBUG_COUNTERS (bl_init): print the -DBL_GEO_STATS counters after a render
  unsigned int switches = 0;          // BL_SWITCH_* (include/blacklight_amd.h): the environment as bl_init found it, never read again
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
  // bl_set_grid beside bl_render (another host thread: the next snapshot of a series staged while this one renders). The render holds
  // render_lock from start to end; a bl_set_grid whose geometry is the one in place uploads its cells into d_cells_back on a stream of
  // its own without the lock, then takes it - waiting for the render to end - to let the two cell arrays change places.
  std::mutex render_lock;
  DeviceBuffer<float> d_cells_back;
  hipStream_t stream_upload = nullptr;
  struct CellPlacement {   // how the caller's planes become d_cells, as the full upload decided (UploadCells)
    bool valid = false, code_kappa = false;
    size_t n_cells = 0;
    int nb[3] = {0, 0, 0};
    int order[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool has_origin = false;
    std::vector<unsigned long long> origin;
    unsigned long long row_stride = 0, plane_stride = 0;
  } placement;
  DeviceBuffer<float> d_cell_planes;                 // bl_set_grid: the caller's eight variable planes as uploaded, before bl_interleave_cells_kernel
  DeviceBuffer<unsigned long long> d_block_origin;   // ... and where every block's cells go in a merged array
  DeviceBuffer<float> d_kappa;   // electron entropy per cell (plasma_model = code_kappa)
  DeviceBuffer<double> d_coords;   // x1f x1v x2f x2v x3f x3v packed
  DeviceBuffer<unsigned short> d_buckets;
  DeviceBuffer<int> d_lattice;   // refined mesh: box of the block-boundary lattice -> block
  DeviceBuffer<unsigned int> d_fused_desc;   // ... -> what bl_shade_fused2_kernel<..., kRefined> needs of the block (BlGridDevice::fused_desc)
  DeviceBuffer<double> d_sks_map;   // simulation_coord = fmks: the reader's SKS -> FMKS look-up table
  DeviceBuffer<int> d_block_table;                  // inter-block interpolation: levels, locations, hash blocks
  DeviceBuffer<unsigned long long> d_block_keys;    // ... and hash keys
  BlGridDevice grid_dev{};
  double grid_outer_x1 = 0.0;   // largest outer x1 face of the grid's blocks (bl_set_grid)
  std::vector<int> merged_block_at;   // merged grid: lattice position (k, j, i of the block) -> MeshBlock of the file
  int merged_blocks[3] = {1, 1, 1};   // ... and the lattice's extent
  bool sample_checkpoint_saved = false;   // checkpoint_sample_save: written with the first image only (radiation_integrator.cpp:699-704)
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

  // Per-chunk scratch, two sets (the geodesic kernel fills one while the shading kernels drain the other when bl_set_overlap is
  // on). Every array has one entry per sample record or per kept sample, so a set holds `record_capacity` of each and a chunk
  // is as many rays as the geodesic kernel fits into it (BlTraceArgs::record_gate).
  struct ChunkSlot {
    DeviceBuffer<BlSampleHot> d_records_hot;
    DeviceBuffer<BlSampleCold> d_records_cold;
    DeviceBuffer<BlLocated> d_located;
    DeviceBuffer<unsigned long long> d_located_tag;
    DeviceBuffer<double2> d_transfer;
    DeviceBuffer<double2> d_composed;              // tolerant tier: one affine map per ray segment (BlShadeArgs::composed)
    DeviceBuffer<double> d_parked;                 // rays bl_geodesic_kernel leaves to bl_geodesic_quad_kernel (BlTraceArgs::parked)
    DeviceBuffer<unsigned long long> d_counters;   // BL_CNT_TOTAL
    DeviceBuffer<BlAuxSample> d_aux;               // auxiliary-image mode
    DeviceBuffer<double> d_sample_t;               // image_time, slow light
    DeviceBuffer<double> d_slow_frac;              // slow light: t_frac of every located sample
    DeviceBuffer<BlPolSample> d_pol_samples;       // polarized transfer
    DeviceBuffer<double> d_pol_matrix;             // tolerant tier: 12 doubles per sample
    DeviceBuffer<BlFreqInputs> d_freq_inputs;      // tolerant tier, several frequencies
    DeviceBuffer<double2> d_pol_coeffs;
    DeviceBuffer<unsigned int> d_anchors;          // inter-block interpolation: eight anchor cells per record
    DeviceBuffer<BlCoefInputs> d_coef_inputs;      // polarized runs: coefficient kernel -> polarized coefficient kernel
    DeviceBuffer<unsigned char> d_have_flags;      // ... where bl_shade_polarized2_kernel evaluates the coefficients itself: which records have them
    DeviceBuffer<unsigned long long> d_redo;       // tolerant tier: records left to the exact coefficient kernel
    DeviceBuffer<double> d_tau_inc;                // tolerant tier with an optical-depth image: alpha x length per sample and frequency
    uint64_t Bytes() const {
      return d_records_hot.count * sizeof(BlSampleHot) + d_records_cold.count * sizeof(BlSampleCold) + d_located.count * sizeof(BlLocated)
          + d_located_tag.count * sizeof(unsigned long long) + (d_transfer.count + d_composed.count) * sizeof(double2) + d_parked.count * sizeof(double) + d_aux.count * sizeof(BlAuxSample)
          + (d_sample_t.count + d_slow_frac.count + d_pol_matrix.count + d_tau_inc.count) * sizeof(double) + d_pol_samples.count * sizeof(BlPolSample)
          + d_freq_inputs.count * sizeof(BlFreqInputs) + d_pol_coeffs.count * sizeof(double2) + d_anchors.count * sizeof(unsigned int)
          + d_coef_inputs.count * sizeof(BlCoefInputs) + d_redo.count * sizeof(unsigned long long) + d_have_flags.count;
    }
    void Free() {
      d_have_flags.Free(); d_redo.Free(); d_tau_inc.Free(); d_aux.Free(); d_sample_t.Free(); d_slow_frac.Free(); d_pol_samples.Free(); d_pol_matrix.Free(); d_freq_inputs.Free(); d_pol_coeffs.Free(); d_coef_inputs.Free(); d_anchors.Free();
      d_records_hot.Free(); d_records_cold.Free(); d_located.Free(); d_located_tag.Free(); d_transfer.Free(); d_composed.Free(); d_parked.Free(); d_counters.Free();
    }
  };
  ChunkSlot slot[2];
  // bl_set_geodesic_reuse: the root level's geodesics as its last render left them. The buffers are the ones scratch set 0 and the
  // per-ray arrays below point at (`parked` false) or, while a render of another level works there, set aside in `store`
  // (`parked` true: that render allocates its own, which are set aside in turn when the root level comes back - two sets of
  // buffers that change places, no copy).
  struct ResidentGeodesics {
    bool valid = false, parked = false;
    bool located_valid = false;               // ... and d_located / d_located_tag / d_anchors hold the samples as located on `geometry`
    std::vector<unsigned char> key;           // everything the records depend on (bl_render.hip: GeodesicKey)
    std::vector<unsigned char> located_key;   // ... and the located samples besides (LocatedKey)
    size_t record_capacity = 0;
    int tail_policy = BL_TAIL_WIDE;
    unsigned long long n_parked = 0, n_flagged = 0;
    unsigned long long counters[BL_CNT_TOTAL] = {};   // scratch set 0's counters as the geodesic (and locate) stage left them
    struct Buffers {
      DeviceBuffer<BlSampleHot> records_hot;
      DeviceBuffer<BlSampleCold> records_cold;
      DeviceBuffer<double> sample_t;
      DeviceBuffer<BlLocated> located;
      DeviceBuffer<unsigned long long> located_tag;
      DeviceBuffer<unsigned int> anchors;
      DeviceBuffer<double> ray_kt, ray_factor;
      DeviceBuffer<int> ray_sample_num, ray_skipped, ray_rows;
      DeviceBuffer<unsigned char> ray_flags;
      DeviceBuffer<long long> ray_out_index, ray_offset;
      uint64_t Bytes() const {
        return records_hot.count * sizeof(BlSampleHot) + records_cold.count * sizeof(BlSampleCold) + sample_t.count * sizeof(double) + located.count * sizeof(BlLocated)
            + located_tag.count * sizeof(unsigned long long) + anchors.count * sizeof(unsigned int) + (ray_kt.count + ray_factor.count) * sizeof(double)
            + (ray_sample_num.count + ray_skipped.count + ray_rows.count) * sizeof(int) + ray_flags.count + (ray_out_index.count + ray_offset.count) * sizeof(long long);
      }
      void Free() {
        records_hot.Free(); records_cold.Free(); sample_t.Free(); located.Free(); located_tag.Free(); anchors.Free(); ray_kt.Free(); ray_factor.Free();
        ray_sample_num.Free(); ray_skipped.Free(); ray_rows.Free(); ray_flags.Free(); ray_out_index.Free(); ray_offset.Free();
      }
    } store;
  } resident;
  int geodesic_reuse = 1;             // bl_set_geodesic_reuse()
  unsigned long long grid_geometry = 0;   // hash of the grid's geometry as bl_set_grid last saw it (block table, coordinates, look-up tables)
  // per ray of a bl_render call (indexed by traversal position; a chunk's kernels get pointers to its first ray)
  DeviceBuffer<double> d_ray_kt, d_ray_factor;
  DeviceBuffer<double> d_ray_start;              // BL_RAY_START_FIELDS rows: start state of every ray (bl_ray_init_kernel -> geodesic kernel)
  DeviceBuffer<int> d_ray_sample_num, d_ray_skipped, d_ray_rows;
  DeviceBuffer<unsigned char> d_ray_flags;
  DeviceBuffer<long long> d_ray_out_index, d_ray_offset;
  uint64_t RayBytes() const {
    return (d_ray_kt.count + d_ray_factor.count + d_ray_start.count) * sizeof(double) + (d_ray_sample_num.count + d_ray_skipped.count + d_ray_rows.count) * sizeof(int) + d_ray_flags.count
        + (d_ray_out_index.count + d_ray_offset.count) * sizeof(long long);
  }
  DeviceBuffer<double> d_freq;
  DeviceBuffer<int> d_pixel_map, d_block_locs, d_tile_order;
  int tile_order_res = 0;
  DeviceBuffer<BlShadeCold> d_shade_cold;
  std::vector<unsigned char> shade_cold_host;   // the bytes d_shade_cold holds (BuildShadeArgs uploads on change only)
  // host-output staging
  DeviceBuffer<double> d_image, d_camera_pos, d_camera_dir;
  DeviceBuffer<int> d_out_sample_num;
  DeviceBuffer<unsigned char> d_out_flags;

  // geodesic checkpoint of the root level (geodesic_checkpoint.cpp:28-108): what LoadGeodesics() read, by pixel
  struct Checkpoint {
    int num_steps = 0;
    double frame[7][4] = {};              // cam_x, u_con, u_cov, norm_con, norm_con_c, hor_con_c, vert_con_c as the file has them
    std::vector<double> frequencies;
    std::vector<double> camera_pos, camera_dir, factors;   // [n_pix][4], [n_pix][4], [n_pix]
    std::vector<uint8_t> flags;
    std::vector<int32_t> sample_num;
    std::vector<double> pos, dir, len;   // [n_pix][num_steps][4] x 2, [n_pix][num_steps]: reference order (far -> near)
  };
  // One loaded file serves every context of the process that names it (the command-line driver's one context per GPU: 67 GB at
  // 1024^2 once, not once per device): LoadGeodesicCheckpoint() keeps a process-wide table of weak references.
  std::shared_ptr<const Checkpoint> checkpoint;

  bl_stats stats{};
};

namespace blhost {
inline void Warn(bl_ctx *ctx, const std::string &message) { ctx->warnings += "Warning: " + message + "\n"; }
int Fail(bl_ctx *ctx, const Failure &failure);   // sets bl_last_error (or the global error when ctx is null), returns the code
void EnsureStreams(bl_ctx *ctx);
}  // namespace blhost

#endif  // BLACKLIGHT_AMD_BL_CTX_H_

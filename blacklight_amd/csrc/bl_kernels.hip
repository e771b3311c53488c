// bl_kernels.hip - hand-written gfx950 (MI355X, CDNA4) kernels of the blacklight hot path.
//
// Pipeline per chunk of rays (fp64 throughout, grid primitives fp32 in HBM; no MFMA: the work is
// per-ray ODE integration and gathers, not a contraction):
//
//   bl_geodesic_kernel   one ray per lane, wave64, persistent waves (1 per SIMD). Camera pixel ->
//                        (x^mu, k_mu), Dormand-Prince 5(4) / RK4 / RK2 stepping in Kerr-Schild
//                        coordinates with the reference's controller, dense-output sampling, online
//                        truncation test. Lanes whose ray has terminated are refilled from a global work
//                        queue (ballot + popcount prefix, one atomic per wave); samples go to 64-byte
//                        records in per-wave blocks of slots.   (geodesics.cpp:39-396, camera.cpp:528-671)
//   bl_locate_kernel     one SAMPLE per lane, 4 waves per SIMD (simulation mode): cuts, CKS->SKS, cell
//                        search on LDS tables, trilinear fractions -> 48-byte located sample.
//                        (simulation_sampling.cpp:201-575)
//   bl_shade_kernel      one SAMPLE per lane, 2 waves per SIMD ("coefficient kernel"): the 8-variable
//                        trilinear read from the interleaved grid, per-sample momentum renormalisation,
//                        thermal-synchrotron j_nu / alpha_nu (or the formula model), and the per-sample
//                        transfer coefficients (a, b) of I <- a (I + b).   (simulation_sampling.cpp:666-1033,
//                        simulation_coefficients.cpp:253-524, formula_coefficients.cpp:62-180,
//                        unpolarized.cpp:74-110)
//   bl_transfer_kernel   one ray per lane: replays the (a, b) records far -> near in the reference's
//                        order and scales by nu^3.   (unpolarized.cpp:71-110, :200-208)
//   auxiliary images     bl_geodesic_kernel<., true> also emits sample times, bl_shade_kernel<., true>
//                        writes (j, alpha) and a BlAuxSample per sample, bl_transfer_aux_kernel integrates
//                        image_light and the nine auxiliary images.   (unpolarized.cpp:113-196)
//
// The reference integrates camera -> source but evaluates the transfer equation source -> camera
// (ReverseGeodesics, geodesics.cpp:808-849); recording (a, b) per sample in the forward pass keeps
// the recurrence bit-identical to the reference's while never materialising its per-sample arrays.
//
// Compile with -ffp-contract=off: bit-exact sample counts depend on it (see blmath.h).
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdlib>

#include "bl_device.h"
#include "bl_pol_frame.h"
#include "bl_bessel.h"

namespace {

constexpr double kPi = 3.141592653589793;    // reference src/blacklight.hpp:12
constexpr double kSqrt2 = 1.4142135623730951;
constexpr double kC = 2.99792458e10;
constexpr double kH = 6.62607015e-27;
constexpr double kMp = 1.67262192369e-24;
constexpr double kMe = 9.1093837015e-28;
constexpr double kE = 4.80320425e-10;
// std::pow(2.0, 11.0 / 12.0) of simulation_coefficients.cpp:480, which g++ folds at compile time to
// the correctly rounded value (the constant is in the reference binary; 11/12 is not)
constexpr double kPow2_11_12 = 0x1.e3437e7101343p+0;
constexpr double kDeltaTauMax = 100.0;       // radiation_integrator.hpp:191
constexpr long long kGateClosed = 1ll << 46;   // added to BL_CNT_COMMITTED by the first refused reservation of a chunk

// Dormand-Prince RK5(4)7M tableau exactly as written in geodesics.cpp:42-72
constexpr double kA[7][6] = {
    {0.0, 0.0, 0.0, 0.0, 0.0, 0.0},
    {1.0 / 5.0, 0.0, 0.0, 0.0, 0.0, 0.0},
    {3.0 / 40.0, 9.0 / 40.0, 0.0, 0.0, 0.0, 0.0},
    {44.0 / 45.0, -56.0 / 15.0, 32.0 / 9.0, 0.0, 0.0, 0.0},
    {19372.0 / 6561.0, -25360.0 / 2187.0, 64448.0 / 6561.0, -212.0 / 729.0, 0.0, 0.0},
    {9017.0 / 3168.0, -355.0 / 33.0, 46732.0 / 5247.0, 49.0 / 176.0, -5103.0 / 18656.0, 0.0},
    {35.0 / 384.0, 0.0, 500.0 / 1113.0, 125.0 / 192.0, -2187.0 / 6784.0, 11.0 / 84.0}};
constexpr double kB5[7] = {35.0 / 384.0, 0.0, 500.0 / 1113.0, 125.0 / 192.0, -2187.0 / 6784.0, 11.0 / 84.0, 0.0};
constexpr double kB4[7] = {5179.0 / 57600.0, 0.0, 7571.0 / 16695.0, 393.0 / 640.0, -92097.0 / 339200.0,
                           187.0 / 2100.0, 1.0 / 40.0};
constexpr double kB4m[7] = {6025192743.0 / 30085553152.0, 0.0, 51252292925.0 / 65400821598.0,
                            -2691868925.0 / 45128329728.0, 187940372067.0 / 1594534317056.0,
                            -1776094331.0 / 19743644256.0, 11237099.0 / 235043384.0};
constexpr double kD[7] = {-12715105075.0 / 11282082432.0, 0.0, 87487479700.0 / 32700410799.0,
                          -10690763975.0 / 1880347072.0, 701980252875.0 / 199316789632.0,
                          -1453857185.0 / 822651844.0, 69997945.0 / 29380423.0};

// std::max / std::min semantics of the reference (first argument wins on NaN / equality)
__device__ __forceinline__ double std_max(double a, double b) { return (a < b) ? b : a; }
__device__ __forceinline__ double std_min(double a, double b) { return (b < a) ? b : a; }

__device__ __forceinline__ int wave_lane() { return threadIdx.x & 63; }

// Wave-level scan / reduction with DPP row shifts and row broadcasts (VALU only: a ds_bpermute
// shuffle costs an LDS round trip that nothing hides at one wave per SIMD).
//   row_shr:n = 0x110 + n, row_bcast:15 = 0x142 (rows 1 and 3 take lane 15 of the row below),
//   row_bcast:31 = 0x143 (rows 2 and 3 take lane 31). Lanes without a source keep `old` = identity.
#define BL_DPP(old, src, ctrl, row_mask) __builtin_amdgcn_update_dpp((old), (src), (ctrl), (row_mask), 0xf, false)

// Inclusive wave scan of ints (64 lanes)
__device__ __forceinline__ int wave_inclusive_scan(int v) {
  v += BL_DPP(0, v, 0x111, 0xf);
  v += BL_DPP(0, v, 0x112, 0xf);
  v += BL_DPP(0, v, 0x114, 0xf);
  v += BL_DPP(0, v, 0x118, 0xf);
  v += BL_DPP(0, v, 0x142, 0xa);
  v += BL_DPP(0, v, 0x143, 0xc);
  return v;
}

// Maximum of non-negative ints over the wave (uniform result)
__device__ __forceinline__ int wave_max_nonneg(int v) {
  int t;
  t = BL_DPP(0, v, 0x111, 0xf); v = t > v ? t : v;
  t = BL_DPP(0, v, 0x112, 0xf); v = t > v ? t : v;
  t = BL_DPP(0, v, 0x114, 0xf); v = t > v ? t : v;
  t = BL_DPP(0, v, 0x118, 0xf); v = t > v ? t : v;
  t = BL_DPP(0, v, 0x142, 0xa); v = t > v ? t : v;
  t = BL_DPP(0, v, 0x143, 0xc); v = t > v ? t : v;
  return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ long long std_max_ll(long long a, long long b) { return a > b ? a : b; }

// Mark the record slots [first, last) as dead (only the id word is written)
__device__ __forceinline__ void retire_record_slots(BlSampleHot *records, int stride, long long first, long long last, int lane) {
  for (long long at = first + lane; at < last; at += 64) {
    records[at * stride].ray = BL_DEAD_RAY;
    records[at * stride].n = 0u;
  }
}

// State vector component order used in the geodesic kernel:
//   0 t, 1 x, 2 y, 3 z, 4 k_x, 5 k_y, 6 k_z, 7 s      (k_t is constant along the ray: d k_t = 0)
// which is the reference's y_vals[0..8] without y_vals[4].
struct RayState {
  double y[8];
  double kt;
};

template <bool kWithDistance, bool kSpinZero>
__device__ __forceinline__ void rhs(const BlSpacetime &st, const double y[8], double kt, double k[8], double *r) {
  double pos[3] = {y[1], y[2], y[3]};
  double kcov[4] = {kt, y[4], y[5], y[6]};
  double dpos[4], dk[3], ds = 0.0;
  bl_geodesic_rhs<kWithDistance, kSpinZero>(st, pos, kcov, dpos, dk, &ds, r);
  k[0] = dpos[0];
  k[1] = dpos[1];
  k[2] = dpos[2];
  k[3] = dpos[3];
  k[4] = dk[0];
  k[5] = dk[1];
  k[6] = dk[2];
  k[7] = ds;
}

// Map a traversal index to the output (ray) index: walk 8x8 pixel tiles so that the 64 lanes of a
// wave start as a compact patch of the image (similar path lengths, shared grid cells).
__device__ __forceinline__ long long traversal_to_ray(long long q, int res, const int *tile_order) {
  if (res <= 0) return q;
  int tiles_per_row = res >> 3;
  long long tile = tile_order != nullptr ? (long long)tile_order[q >> 6] : (q >> 6);
  int within = (int)(q & 63);
  long long tile_row = tile / tiles_per_row;
  int tile_col = (int)(tile % tiles_per_row);
  long long m2 = tile_row * 8 + (within >> 3);
  int m1 = tile_col * 8 + (within & 7);
  return m2 * res + m1;
}

}  // namespace

// =================================================================================================
// Ray start kernel
// =================================================================================================
// One ray of the chunk per lane: pixel -> position, momentum, momentum factor (camera.cpp:393-396, :465-479, :528-671), the
// radial coordinate of the start point and, for the Dormand-Prince stepper, the first stage of the first step
// (geodesics.cpp:113-133, :155-156). This used to be the refill branch of the persistent geodesic kernel, where it ran with
// one or two active lanes almost every time a ray ended (~2 000 instructions per refill, a seventh of that kernel's
// instruction stream) and kept the camera frame - 28 doubles - in scalar registers through every step of every ray. Here
// every lane is busy, and the stepping kernel fetches 17 doubles per new ray instead. Same functions of the same inputs:
// same bits.
template <bool kDormandPrince, bool kSpinZero>
__global__ void __launch_bounds__(256) bl_ray_init_kernel(BlTraceArgs P) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= (long long)P.chunk_rays) return;
  const BlSpacetime st = P.st;
  const long long ray = traversal_to_ray(P.chunk_begin + q, P.swizzle_tiles, P.tile_order);
  const long long pixel = P.pixel_map != nullptr ? (long long)P.pixel_map[ray] : ray;
  double u_ind, v_ind, position[4], direction[4], factor;
  bl_pixel_indices(P.cam, pixel, P.block_locs, &u_ind, &v_ind);
  bl_pixel_ray(st, P.cam, u_ind, v_ind, position, direction, &factor);
  P.ray_kt[q] = direction[0];
  P.ray_factor[q] = factor;
  P.ray_out_index[q] = ray;
  if (P.camera_pos != nullptr)
    for (int mu = 0; mu < 4; mu++) P.camera_pos[4 * ray + mu] = position[mu];
  if (P.camera_dir != nullptr)
    for (int mu = 0; mu < 4; mu++) P.camera_dir[4 * ray + mu] = direction[mu];
  double *start = P.ray_start + q;
  const long long stride = P.ray_start_stride;
  RayState s;
  s.y[0] = position[0];
  s.y[1] = position[1];
  s.y[2] = position[2];
  s.y[3] = position[3];
  s.kt = direction[0];
  s.y[4] = direction[1];
  s.y[5] = direction[2];
  s.y[6] = direction[3];
  s.y[7] = 0.0;
  for (int p = 0; p < 7; p++) start[p * stride] = s.y[p];
  start[7 * stride] = s.kt;
  start[8 * stride] = bl_radial_coordinate<kSpinZero>(st, s.y[1], s.y[2], s.y[3]);
  if (kDormandPrince) {
    double k0[8], r_unused;
    rhs<true, kSpinZero>(st, s.y, s.kt, k0, &r_unused);
    for (int p = 0; p < 8; p++) start[(9 + p) * stride] = k0[p];
  }
}

// =================================================================================================
// Geodesic kernel
// =================================================================================================
// kTime: also emit the coordinate time of every sample (P.sample_t, for image_time). Without it the time
// component is only advanced, never sampled, which keeps its six stage derivatives out of the registers.
// kSpinZero: bh_a == 0.0 known at compile time (bl_geometry.h, "zero spin"): same bits, no hypot.
#ifndef BL_GEO_WAVES
#define BL_GEO_WAVES 2
#endif
// Instantiations held at one wave per SIMD: the Dormand-Prince stepper with spin or sample times. At two waves it needs 36-88
// bytes of scratch per lane, and although a frame with many rays per lane still gains (benchmark frame at a = 0.94: 33.8 ->
// 31.2 ms), the 512^2 formula frame (BASELINE configuration 2: four rays per lane, time set by its longest rays) loses more
// (72 -> 87 ms): a wave that shares its SIMD steps a long ray more slowly.
#ifndef BL_GEO_ONE_WAVE
#define BL_GEO_ONE_WAVE(integrator, with_time, spin_zero) ((integrator) == BL_INTEGRATOR_DP && ((with_time) || !(spin_zero)))
#endif
// A wave-uniform value the optimiser cannot see through (an empty instruction that claims to rewrite its scalar register)
__device__ __forceinline__ int opaque_uniform(int v) {
  asm volatile("" : "+s"(v));
  return v;
}
__device__ __forceinline__ double opaque_uniform(double v) {
  asm volatile("" : "+s"(v));
  return v;
}
template <int kIntegrator, bool kTime, bool kSpinZero, bool kShell = false>
// kShell: the instantiation that leaves no records of steps in the empty shell around the grid (BlTraceArgs::skip_low).
// Two waves per SIMD: the benchmark's instantiation (Dormand-Prince, no sample times, zero spin) and the Runge-Kutta steppers
// fit 256 registers; see BL_GEO_ONE_WAVE for the others.
__global__ void __launch_bounds__(64, BL_GEO_ONE_WAVE(kIntegrator, kTime, kSpinZero) ? 1 : BL_GEO_WAVES) bl_geodesic_kernel(BlTraceArgs P) {
  const int lane = wave_lane();
  const BlSpacetime st = P.st;

  bool have_ray = false;
  bool exhausted = false;
  // per-ray persistent state
  RayState s;
  double k0[8];
  double h_new = 0.0, r_cur = 0.0, r_prev_sample = 0.0;
  int num_retry = 0, n = 0, sample_num = 0, trunc_at = -1;
  int skipped = 0;   // samples of the ray without a record (BlTraceArgs::skip_low)
  unsigned int slot = 0;
  bool previous_fail = false, flag = false;
  for (int p = 0; p < 8; p++) {
    s.y[p] = 0.0;
    k0[p] = 0.0;
  }
  s.kt = 0.0;
  long long block_next = 0, block_end = 0;   // this wave's block of record slots (wave-uniform)
#ifdef BL_GEO_STATS
  // per lane: step attempts, accepted steps, samples emitted; per wave (lane 0): loop iterations, emission iterations, refills,
  // lane-iterations with a ray
  unsigned long long st_attempt = 0, st_accept = 0, st_emit = 0, st_iter = 0, st_emit_iter = 0, st_refill = 0, st_busy = 0;
#endif

  while (true) {
    // ------------------------------------------------------------------ refill idle lanes
    // A ray is handed out only while the record buffers can take its worst case: the leader reserves ray_max_steps slots per
    // idle lane in BL_CNT_COMMITTED and gives back what does not fit under the gate; a finished ray gives back what it did not
    // emit. So the buffers never overflow, and a chunk is as many rays as fit them as the rays turn out (704 of 2 000 steps
    // on the benchmark frame: one chunk where the worst case needed two). The first refusal closes the gate for the whole
    // chunk: every wave finishes the rays it has and ends, so the chunk drains within one ray's time. (Handing out the slots
    // that come back - two thirds of a reservation per finished ray - admits ever fewer rays per generation of rays: 30 ms
    // per chunk boundary. The persistent grid is sized so that its first fill always fits, bl_render.hip.) The rays nobody
    // took (BL_CNT_NEXT_RAY < chunk_rays) are the next chunk's.
    bool need = !have_ray && !exhausted;
    unsigned long long need_mask = __ballot(need);
    if (need_mask != 0ull) {
#ifdef BL_GEO_STATS
      st_refill += 1;
#endif
      const int count = __popcll(need_mask);
      const int leader = __ffsll((long long)need_mask) - 1;
      unsigned long long base = 0ull;
      int admitted = 0;
      if (lane == leader) {
        const unsigned long long per_ray = (unsigned long long)P.ray_max_steps;
        const unsigned long long want = (unsigned long long)count * per_ray;
        const long long over = (long long)(atomicAdd(&P.counters[BL_CNT_COMMITTED], want) + want) - P.record_gate;
        long long refused = over > 0 ? (over + (long long)per_ray - 1) / (long long)per_ray : 0;
        refused = refused < (long long)count ? refused : (long long)count;
        admitted = count - (int)refused;
        // the first refusal closes the gate for every wave (a large constant on the counter: every later reservation is over)
        long long give_back = -(refused * (long long)per_ray) + ((refused > 0 && over < kGateClosed / 2) ? kGateClosed : 0);
        if (admitted > 0) {
          base = atomicAdd(&P.counters[BL_CNT_NEXT_RAY], (unsigned long long)admitted);
          long long beyond = (long long)(base + (unsigned long long)admitted) - (long long)P.chunk_rays;   // past the end of the queue
          beyond = beyond < 0 ? 0 : (beyond < (long long)admitted ? beyond : (long long)admitted);
          give_back -= beyond * (long long)per_ray;
        }
        if (give_back != 0) atomicAdd(&P.counters[BL_CNT_COMMITTED], (unsigned long long)give_back);
      }
      base = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(base >> 32), leader) << 32)
          | (unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)base, leader);
      admitted = __builtin_amdgcn_readlane(admitted, leader);
      if (need) {
        const int rank = __popcll(need_mask & ((1ull << lane) - 1ull));
        const unsigned long long q = base + (unsigned long long)rank;
        if (rank >= admitted || q >= (unsigned long long)P.chunk_rays) {
          exhausted = true;   // no slots for this lane's ray, or no ray left in the queue
        } else {
          have_ray = true;
          slot = (unsigned int)q;
          // the ray's start state as bl_ray_init_kernel left it (geodesics.cpp:113-133, :155-156)
          const double *start = P.ray_start + q;
          const long long stride = P.ray_start_stride;
#pragma unroll
          for (int p = 0; p < 7; p++) s.y[p] = start[p * stride];
          s.y[7] = 0.0;
          s.kt = start[7 * stride];
          r_cur = start[8 * stride];
          if (kIntegrator == BL_INTEGRATOR_DP) {
#pragma unroll
            for (int p = 0; p < 8; p++) k0[p] = start[(9 + p) * stride];
          }
          h_new = -P.ray_step * r_cur;
          num_retry = 0;
          previous_fail = false;
          flag = false;
          n = 0;
          sample_num = 0;
          skipped = 0;
          trunc_at = -1;
          r_prev_sample = 0.0;
        }
      }
    }
    if (__ballot(have_ray) == 0ull) {
      retire_record_slots(P.records_hot, P.record_stride, block_next, block_end, lane);   // unused rest of the last block
      break;
    }
#ifdef BL_GEO_STATS
    st_iter += 1;
    st_busy += have_ray ? 1 : 0;
    st_attempt += (have_ray && (kIntegrator != BL_INTEGRATOR_DP || num_retry <= P.ray_max_retries)) ? 1 : 0;
#endif

    // ------------------------------------------------------------------ one step attempt
    int emit = 0;               // samples this lane writes in this iteration
    int num_steps = 0;          // samples the step contributes to sample_num
    int num_steps_ideal = 1;
    bool accepted = false;
    bool finish = false;
    double h = 0.0;
    double y5[8], k6[8], y4m[8];
    double rv0[8], rv1[8], rv2[8], rv3[8];   // dense-output coefficients (geodesics.cpp:264-273)
    double r_new = 0.0;
    // (dense-output coefficients of -0: a step that stores its one midpoint sample runs the interpolation formula of the
    // emission loop on them and gets that sample back bit for bit - every product and partial sum is -0, and x + (-0) = x
    // for every x, a zero of either sign included)
    for (int p = 0; p < 8; p++) {
      y5[p] = 0.0; k6[p] = 0.0; y4m[p] = 0.0; rv0[p] = -0.0; rv1[p] = -0.0; rv2[p] = -0.0; rv3[p] = -0.0;
    }

    if (have_ray) {
      if (kIntegrator == BL_INTEGRATOR_DP) {
        if (num_retry > P.ray_max_retries) {   // :139-143
          flag = true;
          finish = true;
        } else {
          h = h_new;
          double k1[8], k2[8], k3[8], k4[8], k5[8];
          double yt[8], r_stage;
          // Coordinate time (component 0) and proper distance (component 7) never enter a right-hand
          // side: their stage derivatives are only ever used in the b-weighted sums of the 5th / 4th
          // order solutions, whose terms are added in stage order. Those sums are advanced as soon as
          // each stage is known, so 2 x 6 stage values need not stay live through the later stages.
          double t5 = s.y[0], t4 = s.y[0], s5 = s.y[7];
#define BL_FOLD(Q, K)                     \
          t5 += kB5[Q] * h * K[0];        \
          t4 += kB4[Q] * h * K[0];        \
          s5 += kB5[Q] * h * K[7];
          BL_FOLD(0, k0)
          // stages 1..6 (:162-170): y_temp = y + sum_{q<s} a[s][q] * h * k[q], terms added in q order
#define BL_STAGE(S, KOUT, ...)                                                  \
          {                                                                      \
            const double *kq[6] = {__VA_ARGS__};                                 \
            _Pragma("unroll") for (int p = 1; p < 7; p++) {                      \
              double acc = s.y[p];                                               \
              _Pragma("unroll") for (int q = 0; q < S; q++) acc += kA[S][q] * h * kq[q][p]; \
              yt[p] = acc;                                                       \
            }                                                                    \
            yt[0] = 0.0; yt[7] = 0.0;                                            \
            rhs<true, kSpinZero>(st, yt, s.kt, KOUT, &r_stage);                             \
            BL_FOLD(S, KOUT)                                                     \
          }
          BL_STAGE(1, k1, k0, k0, k0, k0, k0, k0)
          BL_STAGE(2, k2, k0, k1, k0, k0, k0, k0)
          BL_STAGE(3, k3, k0, k1, k2, k0, k0, k0)
          BL_STAGE(4, k4, k0, k1, k2, k3, k0, k0)
          BL_STAGE(5, k5, k0, k1, k2, k3, k4, k0)
          BL_STAGE(6, k6, k0, k1, k2, k3, k4, k5)
#undef BL_STAGE
#undef BL_FOLD
          // 5th / 4th order solutions and error (:173-194). y_vals_5 equals the stage-6 argument
          // bit for bit (same coefficients, same order, the extra b5[6] = 0 term adds +-0), so
          // r_new = RadialGeodesicCoordinate(y_vals_5) is the r of stage 6.
          const double *kk[7] = {k0, k1, k2, k3, k4, k5, k6};
          double error = 0.0;
#pragma unroll
          for (int p = 0; p < 8; p++) {
            double a5 = s.y[p], a4 = s.y[p];
            if (p == 0 && !kTime) {
              a5 = t5;
              a4 = t4;
            } else if (p == 7) {
              a5 = s5;
            } else {
#pragma unroll
              for (int q = 0; q < 7; q++) {
                a5 += kB5[q] * h * kk[q][p];
                a4 += kB4[q] * h * kk[q][p];
              }
            }
            y5[p] = a5;
            if (p < 7) {   // reference p < 8 covers t, x, y, z, k_t (zero difference), k_x, k_y, k_z
              double y_abs = std_max(blm_abs(s.y[p]), blm_abs(a5));
              double error_scale = P.ray_tol_abs + P.ray_tol_rel * y_abs;
              double delta_y = blm_abs(a5 - a4);
              error = std_max(error, delta_y / error_scale);
            }
          }
          r_new = r_stage;

          if (!(error <= 1.0)) {   // :197-209
            double h_factor = 0.2;
            if (error - error == 0.0) {   // std::isfinite
              double h_factor_ideal = 0.9 * bl_pow(error, -0.2);
              h_factor = std_max(h_factor_ideal, 0.2);
            }
            h_new = h * h_factor;
            num_retry += 1;
            previous_fail = true;
          } else {                 // :210-224
            double h_factor = 10.0;
            if (error > 0.0) {
              h_factor = 0.9 * bl_pow(error, -0.2);
              h_factor = std_max(h_factor, 0.2);
              h_factor = std_min(h_factor, 10.0);
            }
            if (previous_fail) h_factor = std_min(h_factor, 1.0);
            h_new = h * h_factor;
            num_retry = 0;
            previous_fail = false;
            accepted = true;

            // midpoint (:227-231), subdivision (:234-245); component 0 (t) of the samples only with kTime
#pragma unroll
            for (int p = kTime ? 0 : 1; p < 7; p++) {
              double acc = s.y[p];
#pragma unroll
              for (int q = 0; q < 7; q++) acc += kB4m[q] * h * kk[q][p];
              y4m[p] = acc;
            }
            double r_mid = bl_radial_coordinate<kSpinZero>(st, y4m[1], y4m[2], y4m[3]);
            double delta_s_step = P.ray_step * r_mid;
            double delta_s_full = y5[7] - s.y[7];
            num_steps_ideal = (int)ceil(delta_s_full / delta_s_step);
            int num_steps_max = P.ray_max_steps - n;
            num_steps = num_steps_ideal;
            if (num_steps > num_steps_max) {
              num_steps = num_steps_max;
              flag = true;
            }
            emit = num_steps;
            if (num_steps_ideal > 1) {   // :262-274
#pragma unroll
              for (int p = kTime ? 0 : 1; p < 7; p++) {
                rv0[p] = y5[p] - s.y[p];
                rv1[p] = s.y[p] - y5[p] + h * k0[p];
                rv2[p] = 2.0 * (y5[p] - s.y[p]) - h * (k0[p] + k6[p]);
                double acc = 0.0;
#pragma unroll
                for (int q = 0; q < 7; q++) acc += kD[q] * h * kk[q][p];
                rv3[p] = acc;
              }
            }
          }
        }
      } else {
        // RK4 (:463-533) / RK2 (:670-722): one sample per step, fixed step rule
        double r = r_cur;
        h = -P.ray_step * (r - P.r_horizon);
        double kv[8], ysub[8], yacc[8], r_unused;
        if (kIntegrator == BL_INTEGRATOR_RK4) {
          rhs<false, kSpinZero>(st, s.y, s.kt, kv, &r_unused);
          for (int p = 0; p < 7; p++) yacc[p] = s.y[p] + 1.0 / 6.0 * h * kv[p];
          for (int p = 0; p < 7; p++) ysub[p] = s.y[p] + 0.5 * h * kv[p];
          ysub[7] = 0.0;
          rhs<false, kSpinZero>(st, ysub, s.kt, kv, &r_unused);
          for (int p = 0; p < 7; p++) yacc[p] += 1.0 / 3.0 * h * kv[p];
          for (int p = 0; p < 7; p++) ysub[p] = s.y[p] + 0.5 * h * kv[p];
          rhs<false, kSpinZero>(st, ysub, s.kt, kv, &r_unused);
          for (int p = 0; p < 7; p++) yacc[p] += 1.0 / 3.0 * h * kv[p];
          for (int p = 0; p < 7; p++) ysub[p] = s.y[p] + h * kv[p];
          rhs<false, kSpinZero>(st, ysub, s.kt, kv, &r_unused);
          for (int p = 0; p < 7; p++) yacc[p] += 1.0 / 6.0 * h * kv[p];
          for (int p = 0; p < 7; p++) y4m[p] = 0.5 * (s.y[p] + yacc[p]);   // stored midpoint (:496-500)
          for (int p = 0; p < 7; p++) y5[p] = yacc[p];
        } else {
          rhs<false, kSpinZero>(st, s.y, s.kt, kv, &r_unused);
          for (int p = 0; p < 7; p++) ysub[p] = s.y[p] + h * kv[p];
          ysub[7] = 0.0;
          for (int p = 0; p < 7; p++) yacc[p] = s.y[p] + 1.0 / 2.0 * h * kv[p];
          for (int p = 0; p < 7; p++) y4m[p] = yacc[p];                    // stored half-step state (:684-688)
          rhs<false, kSpinZero>(st, ysub, s.kt, kv, &r_unused);
          for (int p = 0; p < 7; p++) y5[p] = yacc[p] + 1.0 / 2.0 * h * kv[p];
        }
        y5[7] = 0.0;
        accepted = true;
        num_steps_ideal = 1;
        num_steps = 1;
        emit = 1;
      }
    }

    // ------------------------------------------------------------------ steps that leave no records (BlTraceArgs::skip_low)
    // Every sample of the step lies within d = sum over x, y, z of |r0| + |r1| + |r2| + |r3| of the base point (the weights of the
    // dense output are products of numbers in [0, 1]; a midpoint step has coefficients of -0). Such a step cannot hold the
    // sample that ends the ray either (r <= camera_r, r > r_terminate), and whatever sample follows it compares with a
    // predecessor inside the camera's sphere: any r_prev_sample <= camera_r gives the same answer.
    if (kShell && emit > 0) {
      const bool dense = num_steps_ideal > 1;
      double d = 0.0, rr = 0.0;
#pragma unroll
      for (int p = 1; p < 4; p++) {
        const double b = dense ? s.y[p] : y4m[p];
        rr += b * b;
        d += (blm_abs(rv0[p]) + blm_abs(rv1[p])) + (blm_abs(rv2[p]) + blm_abs(rv3[p]));
      }
      const double low = P.skip_low + d, high = P.skip_high - d;
      if (rr > low * low && high > 0.0 && rr < high * high && trunc_at < 0) {
        skipped += emit;
        emit = 0;
        r_prev_sample = 0.0;
      }
    }

    // ------------------------------------------------------------------ allocate sample slots
    // Each wave owns a block of BL_RECORD_BLOCK consecutive record slots and hands them out with one
    // wave scan per step; the global atomic (whose return has to be waited for, with nothing else to
    // run at one wave per SIMD) is only needed when a block runs out, about once in a dozen steps.
    // The samples of the step are laid end to end in lane order; when they do not fit in what is left of the
    // block, they fill it to its last slot and continue at the start of the next block (a lane's run of samples
    // may straddle the two), so no slot is lost at a switch and chunk_rays * ray_max_steps slots plus one block
    // per wave always suffice. Only the unused tail of a wave's last block is marked dead.
    const int scan = wave_inclusive_scan(emit);
    const int total = __builtin_amdgcn_readlane(scan, 63);
    const int excl = scan - emit;               // this lane's first sample among the step's
    long long old_base = block_next, new_base = 0;
    int old_room = 0x7fffffff;                   // samples of this step that go to the current block
    if (total > 0) {
      const long long remaining = block_end - block_next;
      if ((long long)total <= remaining) {
        block_next += total;
      } else {
        old_room = (int)remaining;
        const unsigned long long grab = (unsigned long long)std_max_ll((long long)total - remaining, BL_RECORD_BLOCK);
        unsigned long long fetched = 0ull;
        if (lane == 63) fetched = atomicAdd(&P.counters[BL_CNT_RECORDS], grab);
        fetched = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(fetched >> 32), 63) << 32)
            | (unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)fetched, 63);
        new_base = (long long)fetched;
        block_next = new_base + ((long long)total - remaining);
        block_end = (long long)(fetched + grab);
        if (block_end > P.record_capacity) {
          // cannot happen with capacity = chunk_rays * ray_max_steps + one block per wave; flagged for the host
          atomicExch(&P.counters[BL_CNT_OVERFLOW], 1ull);
          block_end = block_next = 0;
          emit = 0;
        }
      }
    }

    // ------------------------------------------------------------------ emit samples
    const int max_emit = wave_max_nonneg(emit);
#ifdef BL_GEO_STATS
    st_emit_iter += max_emit;
    st_emit += emit;
    st_accept += accepted ? 1 : 0;
#endif
    // The samples of the step. One midpoint sample (:248-259, and the stored state of RK4 / RK2) or num_steps_ideal samples of
    // the dense output (:277-293): smp = y + frac (r0 + (1 - frac) (r1 + frac (r2 + (1 - frac) r3))), frac = (nn + 0.5) /
    // num_steps_ideal, each of length h / num_steps_ideal. Both run the same formula: a midpoint step has base = the stored
    // state and coefficients of -0 (above), frac = 0.5 / 1 and h / 1 = h. The two quotients share one reciprocal per step
    // (small integers and step lengths: the IEEE quotients, bl_geometry.h).
    const bool dense_output = num_steps_ideal > 1;
    double base[7];
#pragma unroll
    for (int p = kTime ? 0 : 1; p < 7; p++) base[p] = dense_output ? s.y[p] : y4m[p];
    const BlRecip rc_steps = bl_recip((double)num_steps_ideal);
    const double len = bl_div_r(h, rc_steps);
    double position = 0.5;   // nn + 0.5, exact
    for (int nn = 0; nn < max_emit; nn++, position += 1.0) {
      if (nn < emit) {
        double smp[7];
        const double frac = bl_div_r(position, rc_steps);
#pragma unroll
        for (int p = kTime ? 0 : 1; p < 7; p++)
          smp[p] = base[p] + frac * (rv0[p] + (1.0 - frac) * (rv1[p] + frac * (rv2[p] + (1.0 - frac) * rv3[p])));
        // online form of the truncation pass (:327-349): the first sample (index >= 1) that moves
        // outward beyond the camera radius or falls inside r_terminate ends the kept part of the ray
        // (evaluated for every sample, decided only while the ray is still whole: after the first hit nothing reads
        // r_prev_sample again, and the ray ends with this step - its end test is the same comparison at the step's end)
        const int index = n + nn;
        const double r_s = bl_radial_coordinate<kSpinZero>(st, smp[1], smp[2], smp[3]);
        const bool hit = index >= 1 && ((r_s > P.camera_r && r_s > r_prev_sample) || r_s < P.r_terminate);
        const bool dead = trunc_at >= 0 || hit;
        trunc_at = (trunc_at < 0 && hit) ? index : trunc_at;
        r_prev_sample = r_s;
        BlSampleHot hot;
        hot.x = smp[1];
        hot.y = smp[2];
        hot.z = smp[3];
        hot.ray = dead ? BL_DEAD_RAY : slot;
        hot.n = (unsigned int)(kShell ? index - skipped : index);   // its row among the ray's records
        // (a lane's samples side by side, the lanes' runs end to end: consecutive records are consecutive samples of a ray, which
        // read the same grid cells - laid out row by row instead, the coefficient kernels take 4 ms longer per frame)
        const int place = excl + nn;
        const long long at = place < old_room ? old_base + place : new_base + (place - old_room);
        BlSampleCold cold;
        cold.kx = smp[4];
        cold.ky = smp[5];
        cold.kz = smp[6];
        cold.len = len;
        P.records_hot[at * P.record_stride] = hot;
        P.records_cold[at * P.record_stride] = cold;
        if (kTime) P.sample_t[at] = smp[0];
      }
    }

    // ------------------------------------------------------------------ finish the step
    if (have_ray && accepted) {
      // renormalise the spatial momentum at the new point (:296-309 / :507-521 / :696-710)
      double factor = bl_renormalization_factor<kSpinZero>(st, y5[1], y5[2], y5[3], s.kt, y5[4], y5[5], y5[6]);
      y5[4] *= factor;
      y5[5] *= factor;
      y5[6] *= factor;
      double r_before = r_cur;
      if (kIntegrator != BL_INTEGRATOR_DP) r_new = bl_radial_coordinate<kSpinZero>(st, y5[1], y5[2], y5[3]);
      sample_num += num_steps;
      bool terminate_outer = r_new > P.camera_r && r_new > r_before;
      bool terminate_inner = r_new < P.r_terminate;
      if (terminate_outer || terminate_inner) {
        finish = true;
      } else {
        bool last_step = n + num_steps >= P.ray_max_steps;
        if (last_step) flag = true;
        n += num_steps;
        if (n >= P.ray_max_steps) finish = true;
      }
      // FSAL: next step starts from y_vals_5 with k_vals[0] = k_vals[6], the latter evaluated
      // BEFORE the renormalisation above (:149-154)
#pragma unroll
      for (int p = 0; p < 8; p++) {
        s.y[p] = y5[p];
        k0[p] = k6[p];
      }
      r_cur = r_new;
    }
    if (have_ray && finish) {
      const int final_num = ((trunc_at >= 0) ? trunc_at : sample_num) - (kShell ? skipped : 0);   // kept samples with a record
      P.ray_sample_num[slot] = final_num;
      if (kShell) P.ray_skipped[slot] = skipped;
      P.ray_flags[slot] = flag ? 1 : 0;
      // rows of the kept samples in the per-sample arrays, in the order in which rays finish; slots not emitted go back
      P.ray_offset[slot] = (long long)atomicAdd(&P.counters[BL_CNT_SAMPLES], (unsigned long long)final_num);
      atomicAdd(&P.counters[BL_CNT_COMMITTED], (unsigned long long)(-(long long)(P.ray_max_steps - (sample_num - (kShell ? skipped : 0)))));
      have_ray = false;
    }
  }
#ifdef BL_GEO_STATS
  atomicAdd(&P.counters[BL_CNT_DEBUG + 0], st_attempt);
  atomicAdd(&P.counters[BL_CNT_DEBUG + 1], st_accept);
  atomicAdd(&P.counters[BL_CNT_DEBUG + 2], st_emit);
  atomicAdd(&P.counters[BL_CNT_DEBUG + 6], st_busy);
  if (lane == 0) {
    atomicAdd(&P.counters[BL_CNT_DEBUG + 3], st_iter);
    atomicAdd(&P.counters[BL_CNT_DEBUG + 4], st_emit_iter);
    atomicAdd(&P.counters[BL_CNT_DEBUG + 5], st_refill);
  }
#endif
}

// =================================================================================================
// Shading kernel
// =================================================================================================
namespace {

// Coordinate tables of the (single-block) grid staged in LDS: faces and centres of the three axes
// plus one bucket table per axis. The reference finds a cell by a linear scan over faces
// (simulation_sampling.cpp:458-466: first c with xf[c+1] >= x); the bucket table gives a start
// index that is never beyond that result, so a short forward scan lands on the same cell.
struct GridTables {
  const double *xf[3];
  const double *xv[3];
  const unsigned short *bucket[3];
};

__device__ __forceinline__ int find_cell(const BlGridDevice &g, const GridTables &t, int axis, double x) {
  const int n = g.n[axis];
  double u = (x - g.bucket_x0[axis]) * g.bucket_inv_w[axis];
  int b = (int)u;
  b = b < 0 ? 0 : (b >= g.n_bucket[axis] ? g.n_bucket[axis] - 1 : b);
  int c = t.bucket[axis][b];
  const double *xf = t.xf[axis];
  // straight-line common case (the answer is c or c + 1), then the general forward scan
  double f1 = xf[c + 1];
  double f2 = xf[(c + 2 <= n) ? c + 2 : n];
  if (!(f1 >= x)) {
    c += 1;
    if (!(f2 >= x)) {
      c += 1;
      while (c < n - 1 && !(xf[c + 1] >= x)) c++;
    }
  }
  if (c > n - 1) c = n - 1;
  return c;
}

__device__ __forceinline__ void load_cell(const BlGridDevice &g, int k, int j, int i, float v[8]) {
  size_t idx = (((size_t)k * g.n[1] + j) * g.n[0] + i) * 8;
  const float4 *p = reinterpret_cast<const float4 *>(g.cells + idx);
  float4 a = p[0], b = p[1];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// radiation_geometry.cpp:597-658
__device__ __forceinline__ void tetrad_build(const double ucon[4], const double ucov[4], const double kcon[4],
                                             const double kcov[4], const double up_con[4],
                                             const double gcov[4][4], const double gcon[4][4],
                                             double tetrad[4][4]) {
  double omega = 0.0;
  for (int mu = 0; mu < 4; mu++) omega -= kcov[mu] * ucon[mu];
  double k_up_over_omega = 0.0;
  for (int mu = 0; mu < 4; mu++) k_up_over_omega += kcov[mu] * up_con[mu];
  const BlRecip rc_omega = bl_recip(omega);   // six quotients over omega
  k_up_over_omega = bl_div_r(k_up_over_omega, rc_omega);
  double u_up_over_omega = 0.0;
  for (int mu = 0; mu < 4; mu++) u_up_over_omega += ucov[mu] * up_con[mu];
  u_up_over_omega = bl_div_r(u_up_over_omega, rc_omega);
  for (int mu = 0; mu < 4; mu++) tetrad[0][mu] = ucon[mu];
  for (int mu = 0; mu < 4; mu++) tetrad[3][mu] = bl_div_r(kcon[mu], rc_omega) - ucon[mu];
  for (int mu = 0; mu < 4; mu++)
    tetrad[2][mu] = up_con[mu] - k_up_over_omega * tetrad[3][mu] + u_up_over_omega * kcon[mu];
  double norm = 0.0;
  for (int mu = 0; mu < 4; mu++)
    for (int nu = 0; nu < 4; nu++) norm += gcov[mu][nu] * tetrad[2][mu] * tetrad[2][nu];
  norm = bl_sqrt_g(norm);
  const BlRecip rc_norm = bl_recip(norm);
  for (int mu = 0; mu < 4; mu++) tetrad[2][mu] = bl_div_r(tetrad[2][mu], rc_norm);
  double t1[4];
  t1[0] = tetrad[0][1] * (tetrad[2][3] * tetrad[3][2] - tetrad[2][2] * tetrad[3][3])
      + tetrad[0][2] * (tetrad[2][1] * tetrad[3][3] - tetrad[2][3] * tetrad[3][1])
      + tetrad[0][3] * (tetrad[2][2] * tetrad[3][1] - tetrad[2][1] * tetrad[3][2]);
  t1[1] = tetrad[0][0] * (tetrad[2][2] * tetrad[3][3] - tetrad[2][3] * tetrad[3][2])
      + tetrad[0][2] * (tetrad[2][3] * tetrad[3][0] - tetrad[2][0] * tetrad[3][3])
      + tetrad[0][3] * (tetrad[2][0] * tetrad[3][2] - tetrad[2][2] * tetrad[3][0]);
  t1[2] = tetrad[0][0] * (tetrad[2][3] * tetrad[3][1] - tetrad[2][1] * tetrad[3][3])
      + tetrad[0][1] * (tetrad[2][0] * tetrad[3][3] - tetrad[2][3] * tetrad[3][0])
      + tetrad[0][3] * (tetrad[2][1] * tetrad[3][0] - tetrad[2][0] * tetrad[3][1]);
  t1[3] = tetrad[0][0] * (tetrad[2][1] * tetrad[3][2] - tetrad[2][2] * tetrad[3][1])
      + tetrad[0][1] * (tetrad[2][2] * tetrad[3][0] - tetrad[2][0] * tetrad[3][2])
      + tetrad[0][2] * (tetrad[2][0] * tetrad[3][1] - tetrad[2][1] * tetrad[3][0]);
  for (int mu = 0; mu < 4; mu++) {
    double acc = 0.0;
    for (int nu = 0; nu < 4; nu++) acc += gcon[mu][nu] * t1[nu];
    tetrad[1][mu] = acc;
  }
}

// Optional geometric cuts (simulation_sampling.cpp:246-292, formula_coefficients.cpp:78-116); the
// unconditional r > camera_r cut (:238-243) is applied by the caller.
__device__ __forceinline__ bool optional_cuts(const BlShadeCold &c, double x1, double x2, double x3, double r) {
  if (c.omit_near || c.omit_far) {
    double dot_product = x1 * c.cam_x[1] + x2 * c.cam_x[2] + x3 * c.cam_x[3];
    if ((c.omit_near && dot_product > 0.0) || (c.omit_far && dot_product < 0.0)) return true;
  }
  if ((c.omit_in >= 0.0 && r < c.omit_in) || (c.omit_out >= 0.0 && r > c.omit_out)) return true;
  if (c.midplane_theta > 0.0 || c.midplane_theta < 0.0) {
    double th = bl_acos(x3 / r);
    if ((c.midplane_theta > 0.0 && blm_abs(th - kPi / 2.0) > c.midplane_theta)
        || (c.midplane_theta < 0.0 && blm_abs(th - kPi / 2.0) < -c.midplane_theta))
      return true;
  }
  if ((c.midplane_z > 0.0 && blm_abs(x3) > c.midplane_z) || (c.midplane_z < 0.0 && blm_abs(x3) < -c.midplane_z))
    return true;
  if (c.plane) {
    double dot_product = (x1 - c.plane_origin[0]) * c.plane_normal[0] + (x2 - c.plane_origin[1]) * c.plane_normal[1]
        + (x3 - c.plane_origin[2]) * c.plane_normal[2];
    if (dot_product < 0.0) return true;
  }
  return false;
}

// (a, b) of the affine update for one frequency (unpolarized.cpp:92-110); delta_lambda_cgs given
__device__ __forceinline__ double2 transfer_record(double j, double alpha, double delta_lambda_cgs) {
  double2 rec;
  if (alpha > 0.0) {
    double ss = j / alpha;
    double delta_tau = alpha * delta_lambda_cgs;
    if (delta_tau <= kDeltaTauMax) {
      rec.x = bl_exp(-delta_tau);
      rec.y = ss * bl_expm1(delta_tau);
    } else {
      rec.x = BL_THICK_MARK;
      rec.y = ss;
    }
  } else {
    rec.x = 1.0;
    rec.y = j * delta_lambda_cgs;
  }
  return rec;
}

// Everything the per-frequency loop needs from the per-sample (frequency independent) work
struct SampleShade {
  bool have_coefficients;      // false: j = alpha = 0 at every frequency
  double nu_fluid_over_nu;     // -k_mu u^mu (fluid-frame frequency per unit camera frequency*factor)
  double n_e_cgs, nu_c_cgs, theta_e, sin_theta_b, kb_tt_e_cgs;   // simulation
  double cos_theta_b, sin2_theta_b, cos2_theta_b, cos_sign;       // polarized coefficients only
  double theta_e_096, kk_0, kk_1, kk_2;   // polarized, thermal: theta_e^0.96 and K_0,1,2(1 / theta_e) - the same at every frequency
  double n_n0_fluid, fu[4];                                       // formula
  bool have_cell;              // cell_values recorded (simulation_coefficients.cpp:377-387)
  double cell[BL_NUM_CELL_VALUES];
};

// Status of a located sample (BlLocated::status)
enum SampleStatus { kSampleNone = 0, kSampleCut = 1, kSampleOffGrid = 2, kSampleNearest = 3, kSampleInterp = 4,
                    kSampleFormula = 5, kSampleAdvanced = 6 };

__device__ __forceinline__ void unpack_cell(const float4 &lo, const float4 &hi, float v[8]) {
  v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w;
  v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
}

__device__ __forceinline__ const float4 *cell_ptr(const BlGridDevice &g, int k, int j, int i) {
  size_t idx = (((size_t)k * g.n[1] + j) * g.n[0] + i) * 8;
  return reinterpret_cast<const float4 *>(g.cells + idx);
}

// First index i in [0, n) with table[i + 1] >= x, for x in [table[0], table[n]] (the rule of the reference's
// linear scans, simulation_sampling.cpp:458-466), by bisection.
__device__ __forceinline__ int first_upper_face(const double *table, int n, double x) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (table[mid + 1] >= x) hi = mid; else lo = mid + 1;
  }
  return lo;
}

// ---- inter-block interpolation (simulation_block_interp = true)

// Block of a refinement level at a logical location, -1: none. The reference finds it by scanning all blocks
// (simulation_sampling.cpp:1246-1256 and alike); blocks do not overlap, so the key is unique.
__device__ __forceinline__ unsigned long long block_key(int level, int li, int lj, int lk) {
  return ((unsigned long long)(unsigned)level << 57) | ((unsigned long long)(unsigned)li << 38) | ((unsigned long long)(unsigned)lj << 19)
      | (unsigned long long)(unsigned)lk;
}
__device__ __forceinline__ int find_block(const BlGridDevice &g, int level, int li, int lj, int lk) {
  if (level < 0 || level > g.max_level || li < 0 || lj < 0 || lk < 0 || li >= (1 << 19) || lj >= (1 << 19) || lk >= (1 << 19)) return -1;
  const unsigned long long key = block_key(level, li, lj, lk);
  unsigned int slot = (unsigned int)((key * 0x9e3779b97f4a7c15ull) >> 32) & g.hash_mask;
  while (true) {
    const unsigned long long found = g.hash_keys[slot];
    if (found == key) return g.hash_blocks[slot];
    if (found == ~0ull) return -1;
    slot = (slot + 1) & g.hash_mask;
  }
}

// FindNearbyInds (simulation_sampling.cpp:1068-1321): the cell that stands for cell (k, j, i) of block b when an
// index is one beyond the block. c[] = cell closest to the sample, s[] = the sample. Returns the cell's position
// in the [block][k][j][i] array, or -1 where the reference throws "Grid interpolation failed."
__device__ long long find_nearby(const BlGridDevice &g, bool sks, int b, int k, int j, int i, const int c[3], const double s[3]) {
  const int n_i = g.nb[0], n_j = g.nb[1], n_k = g.nb[2];
  const size_t block_cells = (size_t)g.nb[2] * g.stride_plane;
  const int i_safe = max(min(i, n_i - 1), 0), j_safe = max(min(j, n_j - 1), 0), k_safe = max(min(k, n_k - 1), 0);
  if (i == i_safe && j == j_safe && k == k_safe)
    return (long long)(b * block_cells + (size_t)k * g.stride_plane + (size_t)j * g.stride_row + i);
  const int level = g.levels[b];
  const int li = g.locations[3 * b], lj = g.locations[3 * b + 1], lk = g.locations[3 * b + 2];
  const bool upper_i = i > n_i / 2, upper_j = j > n_j / 2, upper_k = k > n_k / 2;
  const int n3 = g.n_3_level0 << level;
  const int fi = upper_i ? li * 2 + 1 : li * 2, fj = upper_j ? lj * 2 + 1 : lj * 2, fk = upper_k ? lk * 2 + 1 : lk * 2;
  // does the mesh continue beyond the block in each direction (:1098-1221)?
  bool x1_off_grid = i != i_safe, x2_off_grid = j != j_safe, x3_off_grid = k != k_safe;
  if (x1_off_grid) {
    const int d = i == -1 ? -1 : 1;
    if (find_block(g, level, li + d, lj, lk) >= 0
        || find_block(g, level - 1, i == -1 ? (li - 1) / 2 : (li + 1) / 2, lj / 2, lk / 2) >= 0
        || find_block(g, level + 1, i == -1 ? li * 2 - 1 : li * 2 + 2, fj, fk) >= 0)
      x1_off_grid = false;
  }
  if (x2_off_grid) {
    const int d = j == -1 ? -1 : 1;
    if (find_block(g, level, li, lj + d, lk) >= 0
        || find_block(g, level - 1, li / 2, j == -1 ? (lj - 1) / 2 : (lj + 1) / 2, lk / 2) >= 0
        || find_block(g, level + 1, fi, j == -1 ? lj * 2 - 1 : lj * 2 + 2, fk) >= 0)
      x2_off_grid = false;
  }
  if (x3_off_grid) {
    const int d = k == -1 ? -1 : 1;
    if (find_block(g, level, li, lj, lk + d) >= 0
        || find_block(g, level - 1, li / 2, lj / 2, k == -1 ? (lk - 1) / 2 : (lk + 1) / 2) >= 0
        || find_block(g, level + 1, fi, fj, k == -1 ? lk * 2 - 1 : lk * 2 + 2) >= 0)
      x3_off_grid = false;
    // across the periodic boundary in x^3 (:1181-1219)
    if (x3_off_grid && sks && k == -1 && lk == 0
        && (find_block(g, level, li, lj, n3 - 1) >= 0 || find_block(g, level - 1, li / 2, lj / 2, (g.n_3_level0 << (level - 1)) - 1) >= 0
            || find_block(g, level + 1, fi, fj, (g.n_3_level0 << (level + 1)) - 1) >= 0))
      x3_off_grid = false;
    if (x3_off_grid && sks && k == n_k && lk == n3 - 1
        && (find_block(g, level, li, lj, 0) >= 0 || find_block(g, level - 1, li / 2, lj / 2, 0) >= 0 || find_block(g, level + 1, fi, fj, 0) >= 0))
      x3_off_grid = false;
  }
  if (x1_off_grid) i = i_safe;
  if (x2_off_grid) j = j_safe;
  if (x3_off_grid) k = k_safe;
  const bool wrap_low = sks && k == -1 && lk == 0, wrap_high = sks && k == n_k && lk == n3 - 1;
  // same level (:1239-1261)
  {
    int lks = k == k_safe ? lk : k == -1 ? lk - 1 : lk + 1;
    if (wrap_low) lks = n3 - 1;
    if (wrap_high) lks = 0;
    const int b_alt = find_block(g, level, i == i_safe ? li : i == -1 ? li - 1 : li + 1, j == j_safe ? lj : j == -1 ? lj - 1 : lj + 1, lks);
    if (b_alt >= 0) {
      const int is = i == i_safe ? i : i == -1 ? n_i - 1 : 0, js = j == j_safe ? j : j == -1 ? n_j - 1 : 0, ks = k == k_safe ? k : k == -1 ? n_k - 1 : 0;
      return (long long)(b_alt * block_cells + (size_t)ks * g.stride_plane + (size_t)js * g.stride_row + is);
    }
  }
  // coarser level (:1264-1291)
  if (level - 1 >= 0) {
    int lks = k == k_safe ? lk / 2 : k == -1 ? (lk - 1) / 2 : (lk + 1) / 2;
    if (wrap_low) lks = (g.n_3_level0 << (level - 1)) - 1;
    if (wrap_high) lks = 0;
    const int b_alt = find_block(g, level - 1, i == i_safe ? li / 2 : i == -1 ? (li - 1) / 2 : (li + 1) / 2,
                                 j == j_safe ? lj / 2 : j == -1 ? (lj - 1) / 2 : (lj + 1) / 2, lks);
    if (b_alt >= 0) {
      const int is = i == i_safe ? (li % 2 * n_i + i) / 2 : i == -1 ? n_i - 1 : 0;
      const int js = j == j_safe ? (lj % 2 * n_j + j) / 2 : j == -1 ? n_j - 1 : 0;
      const int ks = k == k_safe ? (lk % 2 * n_k + k) / 2 : k == -1 ? n_k - 1 : 0;
      return (long long)(b_alt * block_cells + (size_t)ks * g.stride_plane + (size_t)js * g.stride_row + is);
    }
  }
  // finer level (:1294-1316)
  {
    int lks = lk * 2 + (k == k_safe ? 0 : k == -1 ? -1 : 1) + (upper_k ? 1 : 0);
    if (wrap_low && level + 1 <= g.max_level) lks = (g.n_3_level0 << (level + 1)) - 1;
    if (wrap_high) lks = 0;
    const int b_alt = find_block(g, level + 1, li * 2 + (i == i_safe ? 0 : i == -1 ? -1 : 1) + (upper_i ? 1 : 0),
                                 lj * 2 + (j == j_safe ? 0 : j == -1 ? -1 : 1) + (upper_j ? 1 : 0), lks);
    if (b_alt >= 0) {
      int is = i == i_safe ? (upper_i ? (i - n_i / 2) * 2 : i * 2) : i == -1 ? n_i - 2 : 0;
      int js = j == j_safe ? (upper_j ? (j - n_j / 2) * 2 : j * 2) : j == -1 ? n_j - 2 : 0;
      int ks = k == k_safe ? (upper_k ? (k - n_k / 2) * 2 : k * 2) : k == -1 ? n_k - 2 : 0;
      const double *x1v = g.bxv[0] + (size_t)b * n_i, *x2v = g.bxv[1] + (size_t)b * n_j, *x3v = g.bxv[2] + (size_t)b * n_k;
      ks += (k < c[2] || (k == c[2] && s[2] > x3v[c[2]])) ? 1 : 0;
      js += (j < c[1] || (j == c[1] && s[1] > x2v[c[1]])) ? 1 : 0;
      is += (i < c[0] || (i == c[0] && s[0] > x1v[c[0]])) ? 1 : 0;
      return (long long)(b_alt * block_cells + (size_t)ks * g.stride_plane + (size_t)js * g.stride_row + is);
    }
  }
  return -1;
}

// The same on a mesh with refinement: the block from the lattice of block boundaries, then the cell inside
// it from the block's own coordinate rows (global memory; this path is not the benchmark's).
// What the locate kernel finds out about one sample (stored as BlLocated + tag)
struct LocatedSample {
  double f_i, f_j, f_k, ph;
  uint32_t cell, status;
};

__device__ __forceinline__ void locate_sample_refined(const BlShadeArgs &P, unsigned int *anchors, double s1, double s2, double s3,
                                                      LocatedSample *out, unsigned long long *gathers) {
  const BlPlasmaDevice &pl = P.plasma;
  const BlGridDevice &g = P.grid;
  const double s[3] = {s1, s2, s3};
  int box[3];
  for (int a = 0; a < 3; a++) {
    if (s[a] < g.edge[a][0] || s[a] > g.edge[a][g.n_edge[a]]) {
      out->status = kSampleOffGrid;
      return;
    }
    box[a] = first_upper_face(g.edge[a], g.n_edge[a], s[a]);
  }
  const int b = g.lattice[((size_t)box[2] * g.n_edge[1] + box[1]) * g.n_edge[0] + box[0]];
  if (b < 0) {
    out->status = kSampleOffGrid;
    return;
  }
  int c[3];
  const double *xv[3];
  for (int a = 0; a < 3; a++) {
    const int n = g.nb[a];
    const double *xf = g.bxf[a] + (size_t)b * (n + 1);
    xv[a] = g.bxv[a] + (size_t)b * n;
    // start from the position in a uniform block, then walk to the first cell whose upper face is >= s
    int i = (int)((s[a] - xf[0]) / (xf[n] - xf[0]) * (double)n);
    i = i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
    while (i < n - 1 && xf[i + 1] < s[a]) i++;
    while (i > 0 && xf[i] >= s[a]) i--;
    c[a] = i;
  }
  *gathers += 1ull;
  const size_t block_base = (size_t)b * g.nb[2] * g.stride_plane;
  if (!pl.simulation_interp) {
    out->status = kSampleNearest;
    out->cell = (uint32_t)(block_base + (size_t)c[2] * g.stride_plane + (size_t)c[1] * g.stride_row + c[0]);
    return;
  }
  if (g.block_interp) {   // inter-block interpolation (:505-546)
    int m[3], pp[3];
    double f[3];
    bool undefined = false;
    for (int a = 0; a < 3; a++) {
      const int n = g.nb[a], i = c[a];
      const double *xf = g.bxf[a] + (size_t)b * (n + 1);
      m[a] = s[a] >= xv[a][i] ? i : i - 1;
      pp[a] = m[a] + 1;
      // :520-522 read x1v(b, i + 1) at a block's upper edge: the next block's first centre in the reference's
      // Array (xv[a][i + 1] here as well: rows are contiguous); past the array for the last block - undefined
      const bool past_the_array = pp[a] == n && b == g.n_blocks - 1;
      if (past_the_array) undefined = true;
      const double x_m = m[a] == -1 ? 2.0 * xf[i] - xv[a][i] : xv[a][m[a]];
      // BL_UNDEFINED_EDGE: the centre mirrored about the block's upper face, the rule the lower edge has (x_m above)
      const double x_p = (pp[a] == n) ? (past_the_array ? 2.0 * xf[i + 1] - xv[a][i] : 2.0 * xv[a][i + 1] - xv[a][i]) : xv[a][pp[a]];
      f[a] = (s[a] - x_m) / (x_p - x_m);
    }
    if (undefined) {
      atomicAdd(&P.counters[BL_CNT_UNDEFINED], 1ull);
      if (!P.undefined_edge) {
        out->status = kSampleCut;
        return;
      }
    }
    const bool sks = pl.simulation_coord == BL_COORD_SKS;
    bool failed = false;
    for (int corner = 0; corner < 8; corner++) {
      const long long cell = find_nearby(g, sks, b, (corner & 4) ? pp[2] : m[2], (corner & 2) ? pp[1] : m[1], (corner & 1) ? pp[0] : m[0], c, s);
      failed = failed || cell < 0;
      anchors[corner] = (unsigned int)cell;
    }
    if (failed) {
      atomicAdd(&P.counters[BL_CNT_INTERP_FAILED], 1ull);
      out->status = kSampleCut;
      return;
    }
    out->f_i = f[0];
    out->f_j = f[1];
    out->f_k = f[2];
    out->status = kSampleAdvanced;
    out->cell = anchors[0];
    return;
  }
  int m[3];
  double f[3];
  for (int a = 0; a < 3; a++) {   // :485-490 with the block's own centres
    const int i = c[a];
    m[a] = (i == 0 || (i != g.nb[a] - 1 && s[a] >= xv[a][i])) ? i : i - 1;
    f[a] = (s[a] - xv[a][m[a]]) / (xv[a][m[a] + 1] - xv[a][m[a]]);
  }
  out->f_i = f[0];
  out->f_j = f[1];
  out->f_k = f[2];
  out->status = kSampleInterp;
  out->cell = (uint32_t)(block_base + (size_t)m[2] * g.stride_plane + (size_t)m[1] * g.stride_row + m[0]);
}

// Locate one sample on the simulation grid: ConvertFromCKS (radiation_geometry.cpp:37-57), block test
// and cell search (simulation_sampling.cpp:352-394, :458-490), trilinear fractions (:736-760).
// Slow light: which time slice(s) a sample at coordinate time x0 reads (simulation_sampling.cpp:296-349).
// Returns t_ind; *t_frac for slow_interp. Extrapolation beyond the window is recorded per ray and as maxima.
__device__ __forceinline__ int locate_time(const BlSlowDevice &sl, double x0, uint32_t ray, double *t_frac) {
  const double *time = sl.times;
  const int chunk = sl.n;
  const double tolerance = 1.0;   // simulation_reader.hpp:99
  int t_ind = 0;
  *t_frac = 0.0;
  int kind = -1;
  double by = 0.0;
  if (x0 >= time[0]) {
    by = x0 - time[0];
    if (x0 > time[0] + tolerance) kind = 1;
    else if (x0 > time[0]) kind = 0;
  } else if (x0 <= time[chunk - 1]) {
    by = time[chunk - 1] - x0;
    if (x0 < time[chunk - 1] - tolerance) kind = 3;
    else if (x0 < time[chunk - 1]) kind = 2;
    if (sl.interp) {
      t_ind = chunk - 2;
      *t_frac = 1.0;
    } else {
      t_ind = chunk - 1;
    }
  } else {
    while (time[t_ind] > x0) t_ind++;   // first slice not later than the sample (exists: x0 > time[chunk - 1])
    if (sl.interp) {
      t_ind--;
      *t_frac = (x0 - time[t_ind]) / (time[t_ind + 1] - time[t_ind]);
    } else if (time[t_ind - 1] - x0 <= x0 - time[t_ind]) {
      t_ind--;
    }
  }
  if (kind >= 0) {
    atomicOr(&sl.ray_extrap[ray], 1u << kind);
    atomicMax(&sl.extrap_max[kind], (unsigned long long)__double_as_longlong(by));
  }
  return t_ind;
}

template <bool kRefined, bool kSpinZero>
__device__ __forceinline__ void locate_sample(const BlShadeArgs &P, const GridTables &tab, const BlSpacetime &st,
                                              double x1, double x2, double x3, double r, LocatedSample *out,
                                              unsigned long long *gathers, unsigned int *anchors) {
  const BlPlasmaDevice &pl = P.plasma;
  const BlGridDevice &g = P.grid;
  const bool sks = pl.simulation_coord == BL_COORD_SKS;
  double s1 = x1, s2 = x2, s3 = x3;
  out->ph = 0.0;
  out->f_i = out->f_j = out->f_k = 0.0;
  out->cell = 0u;
  if (sks) {
    // z / r is cos(theta) in ConvertFromCKS, in the SKS metric and in the Jacobian (same expression)
    double th = bl_acos(blm_div(x3, r));   // (|z| <= r, both of the order of the coordinates: ordinary operands)
    // zero spin: atan(0 / r) = +0 and atan2(y, x) - 0 = atan2(y, x)
    double ph = kSpinZero ? bl_atan2(x2, x1) : bl_atan2(x2, x1) - bl_atan(blm_div(st.bh_a, r));
    out->ph = ph;
    ph += ph < 0.0 ? 2.0 * kPi : 0.0;
    ph -= ph >= 2.0 * kPi ? 2.0 * kPi : 0.0;
    s1 = r;
    s2 = th;
    s3 = ph;
  }
  if (kRefined) {
    locate_sample_refined(P, anchors, s1, s2, s3, out, gathers);
    return;
  }
  const int n_i = g.n[0], n_j = g.n[1], n_k = g.n[2];
  if (g.fmks) {
    // FMKS grid (simulation_sampling.cpp:190-198, :396-456): bounds in (r, theta, phi); (r, theta) -> (x^1, x^2) from the
    // reader's table by plain scaling (its x^2 read at row j + 1 in both terms, as written there); zone and fraction from
    // the lower FACE in x^1 and x^2 (no half-cell shift), the usual centre rule in x^3. The reference reads cells
    // (k_m .. k_m + 1, j_m .. j_m + 1, i_m .. i_m + 1) of an array without bounds: one beyond a row is the next row's
    // cell - reproduced (the gather steps through the same linear order) - but beyond the block it is another variable's
    // data: no value to reproduce, counted as undefined.
    if (!(s1 >= g.fmks_bounds[0] && s1 <= g.fmks_bounds[1] && s2 >= g.fmks_bounds[2] && s2 <= g.fmks_bounds[3]
          && s3 >= g.fmks_bounds[4] && s3 <= g.fmks_bounds[5])) {
      out->status = kSampleOffGrid;
      return;
    }
    const size_t m1 = (size_t)g.sks_map_n1, m2 = (size_t)g.sks_map_n2;
    double i_ind, j_ind;
    double f_i = bl_modf((s1 - g.sks_map_r_in) / g.sks_map_dr, &i_ind);
    double f_j = bl_modf(s2 / g.sks_map_dtheta, &j_ind);
    size_t mi = (size_t)(int)i_ind, mj = (size_t)(int)j_ind;
    if (mi + 1 >= m1 || mj + 1 >= m2) {   // r = r_out or theta = pi exactly: the reference reads past the table
      atomicAdd(&P.counters[BL_CNT_UNDEFINED], 1ull);
      if (!P.undefined_edge) {
        out->status = kSampleCut;
        return;
      }
      if (mi + 1 >= m1) { mi = m1 - 2; f_i = 1.0; }   // BL_UNDEFINED_EDGE: the table's last entry
      if (mj + 1 >= m2) { mj = m2 - 2; f_j = 1.0; }
    }
    const double fmks_x1 = (1.0 - f_i) * g.sks_map[mj * m1 + mi] + f_i * g.sks_map[mj * m1 + mi + 1];
    const double map_x2 = g.sks_map[(m2 + mj + 1) * m1 + mi];
    const double fmks_x2 = (1.0 - f_j) * map_x2 + f_j * map_x2;
    f_i = bl_modf((fmks_x1 - g.fmks_x1_0) / g.fmks_dx1, &i_ind);
    f_j = bl_modf(fmks_x2 / g.fmks_dx2, &j_ind);
    int i_m = (int)i_ind, j_m = (int)j_ind;
    const int k = find_cell(g, tab, 2, s3);
    const int k_m = (k == 0 || (k != n_k - 1 && s3 >= tab.xv[2][k])) ? k : k - 1;
    const double f_k = (s3 - tab.xv[2][k_m]) / (tab.xv[2][k_m + 1] - tab.xv[2][k_m]);
    const long long n_cells = (long long)n_k * n_j * n_i;
    long long first, last;
    if (!pl.simulation_interp) {
      first = last = ((long long)k * n_j + (f_j >= 0.5 ? j_m + 1 : j_m)) * n_i + (f_i >= 0.5 ? i_m + 1 : i_m);
    } else {
      first = ((long long)k_m * n_j + j_m) * n_i + i_m;
      last = ((long long)(k_m + 1) * n_j + (j_m + 1)) * n_i + (i_m + 1);
    }
    if (first < 0 || last >= n_cells) {
      atomicAdd(&P.counters[BL_CNT_UNDEFINED], 1ull);
      if (!P.undefined_edge) {
        out->status = kSampleCut;
        return;
      }
      // BL_UNDEFINED_EDGE: the zone's own row / column stands for the missing one
      if (pl.simulation_interp) {
        if (j_m + 1 >= n_j) { j_m = n_j - 2; f_j = 1.0; }
        if (i_m + 1 >= n_i) { i_m = n_i - 2; f_i = 1.0; }
        first = ((long long)k_m * n_j + j_m) * n_i + i_m;
      } else {
        const int jn = min(f_j >= 0.5 ? j_m + 1 : j_m, n_j - 1), in = min(f_i >= 0.5 ? i_m + 1 : i_m, n_i - 1);
        first = ((long long)k * n_j + jn) * n_i + in;
      }
    }
    *gathers += 1ull;
    out->cell = (uint32_t)first;
    if (!pl.simulation_interp) {
      out->status = kSampleNearest;
    } else {
      out->f_i = f_i;
      out->f_j = f_j;
      out->f_k = f_k;
      out->status = kSampleInterp;
    }
    return;
  }
  if (s1 < tab.xf[0][0] || s1 > tab.xf[0][n_i] || s2 < tab.xf[1][0] || s2 > tab.xf[1][n_j]
      || s3 < tab.xf[2][0] || s3 > tab.xf[2][n_k]) {
    out->status = kSampleOffGrid;
    return;
  }
  int i = find_cell(g, tab, 0, s1);
  int j = find_cell(g, tab, 1, s2);
  int k = find_cell(g, tab, 2, s3);
  *gathers += 1ull;
  if (!pl.simulation_interp) {   // :710-734
    out->status = kSampleNearest;
    out->cell = (uint32_t)((k * n_j + j) * n_i + i);
    return;
  }
  // :485-490
  // the anchor rule is per block (the indices of :485-487 are block-local): several equal blocks live in one
  // merged array here, g.nb is the block size
  // (one block - the usual case, known to the whole launch - needs no remainders: three integer divisions per sample)
  const bool one_block = g.nb[0] == n_i && g.nb[1] == n_j && g.nb[2] == n_k;
  const int i_b = one_block ? i : i % g.nb[0], j_b = one_block ? j : j % g.nb[1], k_b = one_block ? k : k % g.nb[2];
  int i_m = (i_b == 0 || (i_b != g.nb[0] - 1 && s1 >= tab.xv[0][i])) ? i : i - 1;
  int j_m = (j_b == 0 || (j_b != g.nb[1] - 1 && s2 >= tab.xv[1][j])) ? j : j - 1;
  int k_m = (k_b == 0 || (k_b != g.nb[2] - 1 && s3 >= tab.xv[2][k])) ? k : k - 1;
  // (fractions of cell widths: ordinary operands for the short division; a sample exactly on a centre gives 0 / width = 0)
  out->f_i = blm_div(s1 - tab.xv[0][i_m], tab.xv[0][i_m + 1] - tab.xv[0][i_m]);
  out->f_j = blm_div(s2 - tab.xv[1][j_m], tab.xv[1][j_m + 1] - tab.xv[1][j_m]);
  out->f_k = blm_div(s3 - tab.xv[2][k_m], tab.xv[2][k_m + 1] - tab.xv[2][k_m]);
  out->status = kSampleInterp;
  out->cell = (uint32_t)((k_m * n_j + j_m) * n_i + i_m);
}

// SampleSimulation's nearest / trilinear read (simulation_sampling.cpp:666-1033, InterpolateSimple
// :1334-1351) for a located sample: pr = rho, pgas, uu1..3, bb1..3 as float.
__device__ __forceinline__ void sample_primitives(const BlShadeArgs &P, int status, uint32_t cell, double f_i,
                                                  double f_j, double f_k, float pr[8]) {
  const BlPlasmaDevice &pl = P.plasma;
  const BlGridDevice &g = P.grid;
  const float4 *base = reinterpret_cast<const float4 *>(g.cells) + (size_t)cell * 2;
  if (status == kSampleInterp) {
    // the 8 corner cells: 16 independent 16-byte loads in flight per lane; the two cells of an
    // i-pair are one contiguous 64-byte segment
    const size_t row = (size_t)g.stride_row * 2, plane = (size_t)g.stride_plane * 2;
    float4 lo[8], hi[8];
#pragma unroll
    for (int corner = 0; corner < 8; corner++) {
      const int dk = corner >> 2, dj = (corner >> 1) & 1, di = corner & 1;
      const float4 *p = base + dk * plane + dj * row + di * 2;
      lo[corner] = p[0];
      hi[corner] = p[1];
    }
    // InterpolateSimple sums w_c * v_c over the corners in the order mmm, mmp, mpm, mpp, pmm, pmp,
    // ppm, ppp (:1345-1349); accumulating corner by corner keeps that order.
    const double w_k[2] = {1.0 - f_k, f_k}, w_j[2] = {1.0 - f_j, f_j}, w_i[2] = {1.0 - f_i, f_i};
    double val[8];
    float first[8];
#pragma unroll
    for (int corner = 0; corner < 8; corner++) {
      const int dk = corner >> 2, dj = (corner >> 1) & 1, di = corner & 1;
      float c[8];
      unpack_cell(lo[corner], hi[corner], c);
      double w = w_k[dk] * w_j[dj] * w_i[di];
#pragma unroll
      for (int v = 0; v < 8; v++) {
        if (corner == 0) {
          val[v] = w * (double)c[v];
          first[v] = c[v];
        } else {
          val[v] += w * (double)c[v];
        }
      }
    }
    if (val[0] <= 0.0) val[0] = (double)first[0];   // :822-825
    if (val[1] <= 0.0) val[1] = (double)first[1];
#pragma unroll
    for (int v = 0; v < 8; v++) pr[v] = (float)val[v];   // :830-839
  } else if (status == kSampleNearest) {
    unpack_cell(base[0], base[1], pr);
  } else if (status == kSampleOffGrid) {
    const float fnan = __int_as_float(0x7fc00000);
    pr[0] = pl.fallback_nan ? fnan : P.cold->fallback_rho;    // :377-384, :678-706
    pr[1] = pl.fallback_nan ? fnan : P.cold->fallback_pgas;
    for (int v = 2; v < 8; v++) pr[v] = pl.fallback_nan ? fnan : 0.0f;
  } else {
    for (int v = 0; v < 8; v++) pr[v] = 0.0f;
  }
}

// Inter-block interpolation: the nine values of a sample from its eight anchor cells (InterpolateAdvanced,
// simulation_sampling.cpp:1365-1386: the weights and order of InterpolateSimple; "<= 0 -> first anchor", :936-945)
__device__ __forceinline__ void sample_primitives_advanced(const BlShadeArgs &P, const unsigned int *anchors, double f_i, double f_j,
                                                           double f_k, float pr[8], float *kappa_out) {
  const BlGridDevice &g = P.grid;
  const double w_k[2] = {1.0 - f_k, f_k}, w_j[2] = {1.0 - f_j, f_j}, w_i[2] = {1.0 - f_i, f_i};
  double val[9];
  float first[9];
  for (int corner = 0; corner < 8; corner++) {
    const unsigned int cell = anchors[corner];
    const float4 *p = reinterpret_cast<const float4 *>(g.cells) + (size_t)cell * 2;
    float c[9];
    unpack_cell(p[0], p[1], c);
    c[8] = g.kappa != nullptr ? g.kappa[cell] : 0.0f;
    const double w = w_k[corner >> 2] * w_j[(corner >> 1) & 1] * w_i[corner & 1];
    for (int v = 0; v < 9; v++) {
      if (corner == 0) {
        val[v] = w * (double)c[v];
        first[v] = c[v];
      } else {
        val[v] += w * (double)c[v];
      }
    }
  }
  if (val[0] <= 0.0) val[0] = (double)first[0];
  if (val[1] <= 0.0) val[1] = (double)first[1];
  if (val[8] <= 0.0) val[8] = (double)first[8];
  for (int v = 0; v < 8; v++) pr[v] = (float)val[v];
  *kappa_out = (float)val[8];
}

// The electron entropy of a located sample (plasma_model = code_kappa; simulation_sampling.cpp:726-727,
// :812-833): same nearest / trilinear rule as the other primitives, from its own array.
__device__ __forceinline__ float sample_kappa(const BlShadeArgs &P, int status, uint32_t cell, double f_i, double f_j,
                                              double f_k) {
  const BlGridDevice &g = P.grid;
  const float *base = g.kappa + cell;
  if (status == kSampleInterp) {
    const size_t row = (size_t)g.stride_row, plane = (size_t)g.stride_plane;
    float c[8];
#pragma unroll
    for (int corner = 0; corner < 8; corner++)
      c[corner] = base[(corner >> 2) * plane + ((corner >> 1) & 1) * row + (corner & 1)];
    const double w_k[2] = {1.0 - f_k, f_k}, w_j[2] = {1.0 - f_j, f_j}, w_i[2] = {1.0 - f_i, f_i};
    double val = 0.0;
#pragma unroll
    for (int corner = 0; corner < 8; corner++) {
      const double w = w_k[corner >> 2] * w_j[(corner >> 1) & 1] * w_i[corner & 1];
      val = corner == 0 ? w * (double)c[0] : val + w * (double)c[corner];
    }
    if (val <= 0.0) val = (double)c[0];   // :826-827
    return (float)val;
  }
  if (status == kSampleNearest) return base[0];
  if (status == kSampleOffGrid) return P.plasma.fallback_nan ? __int_as_float(0x7fc00000) : P.cold->fallback_kappa;
  return 0.0f;
}

// Slow light: the nine values of a located sample from time slice t_ind, or blended linearly in time with
// slice t_ind + 1 (simulation_sampling.cpp:710-786 nearest, :809-912 trilinear). Spatial interpolation and the
// "<= 0 -> anchor cell" rule of rho, pgas, kappa apply per slice, before the blend; values stay double until the
// final conversion to float. Not on the benchmark path: plain loops, the extended instantiation only.
__device__ __forceinline__ void sample_slice_values(const BlShadeArgs &P, const float *cells, const float *kappa,
                                                    int status, uint32_t cell, double f_i, double f_j, double f_k,
                                                    double val[9]) {
  const BlGridDevice &g = P.grid;
  const size_t row = (size_t)g.stride_row, plane = (size_t)g.stride_plane;
  if (status == kSampleNearest) {
    const float *c = cells + (size_t)cell * 8;
    for (int v = 0; v < 8; v++) val[v] = (double)c[v];
    val[8] = kappa != nullptr ? (double)kappa[cell] : 0.0;
    return;
  }
  const double w_k[2] = {1.0 - f_k, f_k}, w_j[2] = {1.0 - f_j, f_j}, w_i[2] = {1.0 - f_i, f_i};
  double first[9];
  for (int corner = 0; corner < 8; corner++) {
    const int dk = corner >> 2, dj = (corner >> 1) & 1, di = corner & 1;
    const size_t at = (size_t)cell + dk * plane + dj * row + di;
    const double w = w_k[dk] * w_j[dj] * w_i[di];
    const float *c = cells + at * 8;
    for (int v = 0; v < 9; v++) {
      const double x = v < 8 ? (double)c[v] : (kappa != nullptr ? (double)kappa[at] : 0.0);
      if (corner == 0) {
        val[v] = w * x;
        first[v] = x;
      } else {
        val[v] += w * x;
      }
    }
  }
  if (val[0] <= 0.0) val[0] = first[0];
  if (val[1] <= 0.0) val[1] = first[1];
  if (kappa != nullptr && val[8] <= 0.0) val[8] = first[8];
}

// Slow light with inter-block interpolation: the nine values of one time slice from the eight anchor cells, double,
// with the "<= 0 -> first anchor" rule per slice (simulation_sampling.cpp:960-1033)
__device__ __forceinline__ void sample_slice_values_advanced(const BlShadeArgs &P, const float *cells, const float *kappa,
                                                             const unsigned int *anchors, double f_i, double f_j, double f_k,
                                                             double val[9]) {
  const double w_k[2] = {1.0 - f_k, f_k}, w_j[2] = {1.0 - f_j, f_j}, w_i[2] = {1.0 - f_i, f_i};
  double first[9];
  for (int corner = 0; corner < 8; corner++) {
    const unsigned int cell = anchors[corner];
    const float *c = cells + (size_t)cell * 8;
    const double w = w_k[corner >> 2] * w_j[(corner >> 1) & 1] * w_i[corner & 1];
    for (int v = 0; v < 9; v++) {
      const double x = v < 8 ? (double)c[v] : (kappa != nullptr ? (double)kappa[cell] : 0.0);
      if (corner == 0) {
        val[v] = w * x;
        first[v] = x;
      } else {
        val[v] += w * x;
      }
    }
  }
  if (val[0] <= 0.0) val[0] = first[0];
  if (val[1] <= 0.0) val[1] = first[1];
  if (val[8] <= 0.0) val[8] = first[8];
}

__device__ __forceinline__ void sample_primitives_slow(const BlShadeArgs &P, int status, uint32_t cell, const unsigned int *anchors,
                                                       int t_ind, double t_frac, double f_i, double f_j, double f_k, float pr[8],
                                                       float *kappa_out) {
  const BlPlasmaDevice &pl = P.plasma;
  const BlSlowDevice &sl = P.slow;
  if (status == kSampleAdvanced) {
    const bool entropy = pl.code_kappa != 0;
    double val[9];
    sample_slice_values_advanced(P, sl.cells[t_ind], entropy ? sl.kappa[t_ind] : nullptr, anchors, f_i, f_j, f_k, val);
    if (sl.interp) {
      double next[9];
      sample_slice_values_advanced(P, sl.cells[t_ind + 1], entropy ? sl.kappa[t_ind + 1] : nullptr, anchors, f_i, f_j, f_k, next);
      for (int v = 0; v < 9; v++) val[v] = (1.0 - t_frac) * val[v] + t_frac * next[v];
    }
    for (int v = 0; v < 8; v++) pr[v] = (float)val[v];
    *kappa_out = entropy ? (float)val[8] : 0.0f;
  } else if (status == kSampleInterp || status == kSampleNearest) {
    const bool entropy = pl.code_kappa != 0;
    double val[9];
    sample_slice_values(P, sl.cells[t_ind], entropy ? sl.kappa[t_ind] : nullptr, status, cell, f_i, f_j, f_k, val);
    if (sl.interp) {
      double next[9];
      sample_slice_values(P, sl.cells[t_ind + 1], entropy ? sl.kappa[t_ind + 1] : nullptr, status, cell, f_i, f_j, f_k, next);
      for (int v = 0; v < 9; v++) val[v] = (1.0 - t_frac) * val[v] + t_frac * next[v];
    }
    for (int v = 0; v < 8; v++) pr[v] = (float)val[v];
    *kappa_out = entropy ? (float)val[8] : 0.0f;
  } else if (status == kSampleOffGrid) {
    const float fnan = __int_as_float(0x7fc00000);
    pr[0] = pl.fallback_nan ? fnan : P.cold->fallback_rho;
    pr[1] = pl.fallback_nan ? fnan : P.cold->fallback_pgas;
    for (int v = 2; v < 8; v++) pr[v] = pl.fallback_nan ? fnan : 0.0f;
    *kappa_out = pl.fallback_nan ? fnan : P.cold->fallback_kappa;
  } else {
    for (int v = 0; v < 8; v++) pr[v] = 0.0f;
    *kappa_out = 0.0f;
  }
}

// Simulation mode: the frequency-independent part of CalculateSimulationCoefficients
// (simulation_coefficients.cpp:253-455). kExtended: the instantiation that also knows plasma_model = code_kappa.
template <bool kExtended, bool kSksCurved>
__device__ __forceinline__ void sample_finish_simulation(const BlShadeArgs &P, const BlSpacetime &st,
                                                         const BlKerrSchild &ks, double cth, double ph_unwrapped,
                                                         const float pr[8], float kappa_f, const double kcov[4],
                                                         int need_coefficients, SampleShade *out, BlPolSample *pol_out) {
  const BlPlasmaDevice &pl = P.plasma;
  const double bh_a = st.bh_a, bh_m = st.bh_m;
  const double r = ks.r, r2 = ks.r2, a2 = ks.a2;
  // kSksCurved: spherical Kerr-Schild simulation in a curved spacetime known at compile time (the common case)
  const bool sks = kSksCurved || pl.simulation_coord == BL_COORD_SKS;
  const bool ray_flat = !kSksCurved && st.ray_flat;

  // ---------------- coefficients (simulation_coefficients.cpp:274-455)
  const double rho = pr[0], pgas = pr[1];
  const double uu1 = pr[2], uu2 = pr[3], uu3 = pr[4];
  const double bb1 = pr[5], bb2 = pr[6], bb3 = pr[7];
  const double rho_cgs = rho * pl.d_unit;
  const double pgas_cgs = pgas * pl.e_unit;
  // The quotients below have operands that are products of single-precision grid values (zero, or
  // 1e-45 .. 3e38 in magnitude) and unit constants: far inside the range where bl_div_g() is the IEEE
  // quotient (bl_geometry.h). Quotients that involve results of exp() keep the plain division.
  const double n_cgs = bl_div_g(rho_cgs, pl.plasma_mu * kMp);
  const double n_e_cgs = bl_div_g(n_cgs, 1.0 + 1.0 / pl.plasma_ne_ni);

  // Velocity and field in simulation coordinates (:292-330). In SKS the metric is sparse
  // (radiation_geometry.cpp:462-489, :543-571); sums that the reference runs over all 16 entries
  // are evaluated over the non-zero ones, which is exact (the dropped terms are 0 * finite = +-0).
  double ucon_sim[4], bcon_sim[4], b_sq;
  if (sks) {
    const double cth2 = cth * cth;
    const double sth2 = 1.0 - cth2;
    const double delta = r2 - 2.0 * bh_m * r + a2;
    const double sigma = r2 + a2 * cth2;
    // three quotients over sigma (2 m r / sigma is one expression in the reference, used six times)
    const BlRecip rc_sigma = bl_recip(sigma);
    const double two_mr_sigma = bl_div_r(2.0 * bh_m * r, rc_sigma);
    const double g00 = -(1.0 - two_mr_sigma);
    const double g01 = two_mr_sigma;
    const double g03 = bl_div_r(-2.0 * bh_m * bh_a * r * sth2, rc_sigma);
    const double g11 = 1.0 + two_mr_sigma;
    const double g13 = -(1.0 + two_mr_sigma) * bh_a * sth2;
    const double g22 = sigma;
    const double g33 = (r2 + a2 + bl_div_r(2.0 * bh_m * a2 * r * sth2, rc_sigma)) * sth2;
    const double gc00 = -(1.0 + two_mr_sigma);
    const double gc01 = two_mr_sigma;
    (void)delta;
    // uu0 (:297-300): gcov_sim[1][2] = gcov_sim[2][3] = 0
    const double uu0 = bl_sqrt_g(1.0 + g11 * uu1 * uu1 + 2.0 * 0.0 * uu1 * uu2 + 2.0 * g13 * uu1 * uu3
        + g22 * uu2 * uu2 + 2.0 * 0.0 * uu2 * uu3 + g33 * uu3 * uu3);
    const double lapse = bl_div_g(1.0, bl_sqrt_g(-gc00));
    const double shift1 = bl_div_g(-gc01, gc00);
    // shift2 = shift3 = -gcon_sim[0][2,3] / gcon_sim[0][0] = -0 / gc00 = +0 (gc00 <= -1, or NaN and then
    // uu0 is NaN too), so the reference's  uu_a - shift_a * uu0 / lapse  subtracts (+0 * uu0) / lapse:
    // +0 when uu0 is finite (lapse is in (0, 1]), NaN when uu0 is not. 0.0 * uu0 is exactly that.
    const BlRecip rc_lapse = bl_recip(lapse);
    ucon_sim[0] = bl_div_r(uu0, rc_lapse);
    ucon_sim[1] = uu1 - bl_div_r(shift1 * uu0, rc_lapse);
    ucon_sim[2] = uu2 - 0.0 * uu0;
    ucon_sim[3] = uu3 - 0.0 * uu0;
    // ucov_sim[mu] = sum_nu gcov_sim[mu][nu] ucon_sim[nu] (:310-313), non-zero entries only
    const double ucov1 = (g01 * ucon_sim[0] + g11 * ucon_sim[1]) + g13 * ucon_sim[3];
    const double ucov2 = g22 * ucon_sim[2];
    const double ucov3 = (g03 * ucon_sim[0] + g13 * ucon_sim[1]) + g33 * ucon_sim[3];
    bcon_sim[0] = ucov1 * bb1 + ucov2 * bb2 + ucov3 * bb3;
    const BlRecip rc_u0 = bl_recip(ucon_sim[0]);
    bcon_sim[1] = bl_div_r(bb1 + bcon_sim[0] * ucon_sim[1], rc_u0);
    bcon_sim[2] = bl_div_r(bb2 + bcon_sim[0] * ucon_sim[2], rc_u0);
    bcon_sim[3] = bl_div_r(bb3 + bcon_sim[0] * ucon_sim[3], rc_u0);
    const double bcov0 = (g00 * bcon_sim[0] + g01 * bcon_sim[1]) + g03 * bcon_sim[3];
    const double bcov1 = (g01 * bcon_sim[0] + g11 * bcon_sim[1]) + g13 * bcon_sim[3];
    const double bcov2 = g22 * bcon_sim[2];
    const double bcov3 = (g03 * bcon_sim[0] + g13 * bcon_sim[1]) + g33 * bcon_sim[3];
    b_sq = bcov0 * bcon_sim[0] + bcov1 * bcon_sim[1] + bcov2 * bcon_sim[2] + bcov3 * bcon_sim[3];
  } else {
    // Cartesian Kerr-Schild simulation: same metric as the geodesic one, never flat
    double gs_cov[4][4], gs_con[4][4];
    bl_gcov_ks(ks, gs_cov);
    bl_gcon_ks(ks, gs_con);
    const double uu0 = bl_sqrt_g(1.0 + gs_cov[1][1] * uu1 * uu1 + 2.0 * gs_cov[1][2] * uu1 * uu2
        + 2.0 * gs_cov[1][3] * uu1 * uu3 + gs_cov[2][2] * uu2 * uu2 + 2.0 * gs_cov[2][3] * uu2 * uu3
        + gs_cov[3][3] * uu3 * uu3);
    const double lapse = 1.0 / bl_sqrt_g(-gs_con[0][0]);
    const double shift1 = -gs_con[0][1] / gs_con[0][0];
    const double shift2 = -gs_con[0][2] / gs_con[0][0];
    const double shift3 = -gs_con[0][3] / gs_con[0][0];
    ucon_sim[0] = uu0 / lapse;
    ucon_sim[1] = uu1 - shift1 * uu0 / lapse;
    ucon_sim[2] = uu2 - shift2 * uu0 / lapse;
    ucon_sim[3] = uu3 - shift3 * uu0 / lapse;
    double ucov_sim[4];
    for (int mu = 0; mu < 4; mu++) {
      double acc = 0.0;
      for (int nu = 0; nu < 4; nu++) acc += gs_cov[mu][nu] * ucon_sim[nu];
      ucov_sim[mu] = acc;
    }
    bcon_sim[0] = ucov_sim[1] * bb1 + ucov_sim[2] * bb2 + ucov_sim[3] * bb3;
    bcon_sim[1] = (bb1 + bcon_sim[0] * ucon_sim[1]) / ucon_sim[0];
    bcon_sim[2] = (bb2 + bcon_sim[0] * ucon_sim[2]) / ucon_sim[0];
    bcon_sim[3] = (bb3 + bcon_sim[0] * ucon_sim[3]) / ucon_sim[0];
    b_sq = 0.0;
    for (int mu = 0; mu < 4; mu++) {
      double acc = 0.0;
      for (int nu = 0; nu < 4; nu++) acc += gs_cov[mu][nu] * bcon_sim[nu];
      b_sq += acc * bcon_sim[mu];
    }
  }
  const double bb_cgs = bl_sqrt_g(b_sq) * pl.b_unit;
  const double sigma_cut = b_sq / rho;
  const double beta_inv = bl_div_g(b_sq, 2.0 * pgas);

  // electron temperature, T_i/T_e(beta) model (:333-348)
  double theta_e = __longlong_as_double(0x7ff8000000000000ll);
  double kb_tt_e_cgs = theta_e;
  if (kExtended && pl.code_kappa) {
    // electron entropy model (:351-358)
    if (pl.plasma_thermal_frac != 0.0) {
      const double kappa = kappa_f;
      const double mu_e = pl.plasma_mu * (1.0 + 1.0 / pl.plasma_ne_ni);
      const double rho_e = rho * kMe / (mu_e * kMp);
      const double rho_kappa_e_cbrt = bl_cbrt(rho_e * kappa);
      theta_e = 1.0 / 5.0 * (blm_sqrt(1.0 + 25.0 * rho_kappa_e_cbrt * rho_kappa_e_cbrt) - 1.0);
      kb_tt_e_cgs = theta_e * kMe * kC * kC;
    }
  } else if (pl.plasma_thermal_frac != 0.0) {
    double tti_tte = bl_div_g(pl.plasma_rat_high + pl.plasma_rat_low * beta_inv * beta_inv, 1.0 + beta_inv * beta_inv);
    double kb_tt_tot_cgs = bl_div_g(pl.plasma_mu * kMp * pgas_cgs, rho_cgs);
    if (pl.plasma_use_p) {
      kb_tt_e_cgs = bl_div_g(1.0 + pl.plasma_ne_ni, tti_tte + pl.plasma_ne_ni) * kb_tt_tot_cgs;
    } else {
      kb_tt_e_cgs = (1.0 + pl.plasma_ne_ni) * kb_tt_tot_cgs / (P.cold->plasma_gamma - 1.0);
      kb_tt_e_cgs /= tti_tte / (P.cold->plasma_gamma_i - 1.0) + pl.plasma_ne_ni / (P.cold->plasma_gamma_e - 1.0);
    }
    theta_e = bl_div_g(kb_tt_e_cgs, kMe * kC * kC);
  }

  // cell cuts (:361-375); all thresholds negative = disabled is the common case
  bool cell_cut = false;
  if (pl.any_cell_cut) {
    const BlShadeCold &cc = *P.cold;
    // disabled thresholds are -inf / +inf (bl_api.hip): one compare each, same decisions as "cut >= 0 and ..."
    cell_cut = rho_cgs < cc.cut_rho_min || rho_cgs > cc.cut_rho_max || n_e_cgs < cc.cut_n_e_min || n_e_cgs > cc.cut_n_e_max
        || pgas_cgs < cc.cut_p_gas_min || pgas_cgs > cc.cut_p_gas_max || theta_e < cc.cut_theta_e_min || theta_e > cc.cut_theta_e_max
        || bb_cgs < cc.cut_b_min || bb_cgs > cc.cut_b_max || sigma_cut < cc.cut_sigma_min || sigma_cut > cc.cut_sigma_max
        || beta_inv < cc.cut_beta_inverse_min || beta_inv > cc.cut_beta_inverse_max;
  }
  const bool no_field = bb1 == 0.0 && bb2 == 0.0 && bb3 == 0.0;   // :394
  out->have_coefficients = false;
  out->have_cell = false;
  if (cell_cut) return;
  out->have_cell = true;   // :377-387 (used in auxiliary-image mode only)
  out->cell[0] = rho_cgs;
  out->cell[1] = n_e_cgs;
  out->cell[2] = pgas_cgs;
  out->cell[3] = theta_e;
  out->cell[4] = bb_cgs;
  out->cell[5] = sigma_cut;
  out->cell[6] = beta_inv;
  if (need_coefficients == 0 || no_field) return;   // :389-395

  // Transform u and b to geodesic (CKS) coordinates (:398-408). The Jacobian of
  // radiation_geometry.cpp:69-126 has row/column 0 = identity and jacobian[3][3] = 0; the products
  // with those 0 / 1 entries are dropped (exact).
  double ucon[4], bcon[4];
  if (sks) {
    const double sth = bl_sqrt_g(1.0 - cth * cth);
    double sph, cph;
    bl_sincos(ph_unwrapped, &sph, &cph);
    const double j11 = sth * cph;
    const double j12 = cth * (r * cph - bh_a * sph);
    const double j13 = sth * (-r * sph - bh_a * cph);
    const double j21 = sth * sph;
    const double j22 = cth * (r * sph + bh_a * cph);
    const double j23 = sth * (r * cph - bh_a * sph);
    const double j31 = cth;
    const double j32 = -r * sth;
    ucon[0] = ucon_sim[0];
    ucon[1] = (j11 * ucon_sim[1] + j12 * ucon_sim[2]) + j13 * ucon_sim[3];
    ucon[2] = (j21 * ucon_sim[1] + j22 * ucon_sim[2]) + j23 * ucon_sim[3];
    ucon[3] = j31 * ucon_sim[1] + j32 * ucon_sim[2];
    bcon[0] = bcon_sim[0];
    bcon[1] = (j11 * bcon_sim[1] + j12 * bcon_sim[2]) + j13 * bcon_sim[3];
    bcon[2] = (j21 * bcon_sim[1] + j22 * bcon_sim[2]) + j23 * bcon_sim[3];
    bcon[3] = j31 * bcon_sim[1] + j32 * bcon_sim[2];
  } else {
    for (int mu = 0; mu < 4; mu++) {
      ucon[mu] = ucon_sim[mu];
      bcon[mu] = bcon_sim[mu];
    }
  }
  // kcon, ucov, bcov in the geodesic metric (:411-428); the metric is rebuilt from the Kerr-Schild
  // scalars here (a dozen multiplies) rather than kept live across the gather
  double gcov[4][4], gcon[4][4];
  if (ray_flat) {
    bl_minkowski(gcov);
    bl_minkowski(gcon);
  } else {
    bl_gcov_ks(ks, gcov);
    bl_gcon_ks(ks, gcon);
  }
  double kcon[4], ucov[4], bcov[4];
  for (int mu = 0; mu < 4; mu++) {
    double ak = 0.0, au = 0.0, ab = 0.0;
    for (int nu = 0; nu < 4; nu++) {
      ak += gcon[mu][nu] * kcov[nu];
      au += gcov[mu][nu] * ucon[nu];
      ab += gcov[mu][nu] * bcon[nu];
    }
    kcon[mu] = ak;
    ucov[mu] = au;
    bcov[mu] = ab;
  }
  double tetrad[4][4];
  tetrad_build(ucon, ucov, kcon, kcov, bcon, gcov, gcon, tetrad);
  double k_tet[3] = {0.0, 0.0, 0.0}, b_tet[3] = {0.0, 0.0, 0.0};   // :434-455
  for (int mu = 0; mu < 4; mu++)
    for (int a = 0; a < 3; a++) {
      k_tet[a] += tetrad[a + 1][mu] * kcov[mu];
      b_tet[a] += tetrad[a + 1][mu] * bcov[mu];
    }
  const double k_sq_tet = k_tet[0] * k_tet[0] + k_tet[1] * k_tet[1] + k_tet[2] * k_tet[2];
  const double b_sq_tet = b_tet[0] * b_tet[0] + b_tet[1] * b_tet[1] + b_tet[2] * b_tet[2];
  const double k_b_tet = k_tet[0] * b_tet[0] + k_tet[1] * b_tet[1] + k_tet[2] * b_tet[2];
  const double cos2_theta_b = std_min(bl_div_g(k_b_tet * k_b_tet, k_sq_tet * b_sq_tet), 1.0);
  const double sin2_theta_b = 1.0 - cos2_theta_b;
  double nu_sum = 0.0;   // :461-463
  for (int mu = 0; mu < 4; mu++) nu_sum -= kcov[mu] * ucon[mu];
  out->have_coefficients = true;
  out->nu_fluid_over_nu = nu_sum;
  out->n_e_cgs = n_e_cgs;
  out->nu_c_cgs = bl_div_g(kE * bb_cgs, 2.0 * kPi * kMe * kC);
  out->theta_e = theta_e;
  out->sin_theta_b = bl_sqrt_g(sin2_theta_b);
  out->kb_tt_e_cgs = kb_tt_e_cgs;
  out->sin2_theta_b = sin2_theta_b;
  out->cos2_theta_b = cos2_theta_b;
  out->cos_sign = k_b_tet >= 0.0 ? 1.0 : -1.0;
  out->cos_theta_b = bl_sqrt_g(cos2_theta_b) * (k_b_tet >= 0.0 ? 1.0 : -1.0);   // :455
  if (kExtended && pol_out != nullptr) {
    // polarized.cpp:163-265 rebuilds k^mu and this tetrad from the same sampled values: hand them over
    for (int mu = 0; mu < 4; mu++) {
      pol_out->kcon[mu] = kcon[mu];
      pol_out->e1[mu] = tetrad[1][mu];
      pol_out->e2[mu] = tetrad[2][mu];
    }
  }
}

// exact arithmetic tier of the coefficient formulas
#define BLC_NAME(f) f
#define BLC_SQRT bl_sqrt_g
#define BLC_SQRT_M blm_sqrt
#define BLC_CBRT bl_cbrt
#define BLC_EXP bl_exp
#define BLC_EXPM1 bl_expm1
#define BLC_LOG bl_log
#define BLC_POW bl_pow
#define BLC_POWBASE_T blm_powbase
#define BLC_POW_BASE bl_pow_base
#define BLC_POW_OF bl_pow_of
#define BLC_DIV_G bl_div_g
#define BLC_SIN bl_sin
#define BLC_COS bl_cos
#define BLC_TANH bl_tanh
#include "bl_coefficients.inc"
#undef BLC_NAME
#undef BLC_SQRT
#undef BLC_SQRT_M
#undef BLC_CBRT
#undef BLC_EXP
#undef BLC_EXPM1
#undef BLC_LOG
#undef BLC_POW
#undef BLC_POWBASE_T
#undef BLC_POW_BASE
#undef BLC_POW_OF
#undef BLC_DIV_G
#undef BLC_SIN
#undef BLC_COS
#undef BLC_TANH

// Formula mode, one sample (formula_coefficients.cpp:118-161)
__device__ __forceinline__ void shade_formula(const BlShadeArgs &P, const BlSpacetime &st, double r, double x1,
                                              double x2, double x3, SampleShade *out) {
  const BlFormulaDevice &fm = P.formula;
  const double bh_a = st.bh_a, bh_m = st.bh_m;
  double rr = blm_sqrt(r * r - x3 * x3);
  double cth = x3 / r;
  double sth = blm_sqrt(1.0 - cth * cth);
  double ph = bl_atan2(x2, x1) - bl_atan(bh_a / r);
  double sph, cph;
  bl_sincos(ph, &sph, &cph);
  double delta = r * r - 2.0 * bh_m * r + bh_a * bh_a;
  double sigma = r * r + bh_a * bh_a * cth * cth;
  double gtt_bl = -(1.0 + 2.0 * bh_m * r * (r * r + bh_a * bh_a) / (delta * sigma));
  double gtph_bl = -2.0 * bh_m * bh_a * r / (delta * sigma);
  double grr_bl = delta / sigma;
  double gthth_bl = 1.0 / sigma;
  double gphph_bl = (sigma - 2.0 * bh_m * r) / (delta * sigma * sth * sth);
  double ll = fm.l0 / (1.0 + rr) * bl_pow(rr, 1.0 + fm.q);
  double u_norm = 1.0 / blm_sqrt(-gtt_bl + 2.0 * gtph_bl * ll - gphph_bl * ll * ll);
  double u_t_bl = -u_norm;
  double u_r_bl = 0.0;
  double u_th_bl = 0.0;
  double u_ph_bl = u_norm * ll;
  double ut_bl = gtt_bl * u_t_bl + gtph_bl * u_ph_bl;
  double ur_bl = grr_bl * u_r_bl;
  double uth_bl = gthth_bl * u_th_bl;
  double uph_bl = gtph_bl * u_t_bl + gphph_bl * u_ph_bl;
  double ut = ut_bl + 2.0 * bh_m * r / delta * ur_bl;
  double ur = ur_bl;
  double uth = uth_bl;
  double uph = uph_bl + bh_a / delta * ur_bl;
  out->fu[0] = ut;
  out->fu[1] = sth * cph * ur + cth * (r * cph - bh_a * sph) * uth + sth * (-r * sph - bh_a * cph) * uph;
  out->fu[2] = sth * sph * ur + cth * (r * sph + bh_a * cph) * uth + sth * (r * cph - bh_a * sph) * uph;
  out->fu[3] = cth * ur - r * sth * uth;
  out->n_n0_fluid = bl_exp(-0.5 * (r * r / (fm.r0 * fm.r0) + fm.h * fm.h * cth * cth));
  out->have_coefficients = true;
}

}  // namespace

// ---- locate kernel (simulation mode): one sample record per lane. Coordinate conversion and the
// LDS table walks of the cell search; no grid reads (the coefficient kernel issues those, where they
// overlap its arithmetic instead of saturating the texture addresser here).
// kRefined: mesh with refinement; block and cell come from tables in global memory, no LDS staging.
// kSlow: slow light; the time slice of every sample that passed the cuts is found first (:296-349).
// kTablesInHbm: the coordinate tables of a merged grid are too large for LDS and are searched where they lie (a
// compile-time choice: table pointers that may be either LDS or global become flat loads, each of which waits on both
// memory counters).
template <bool kRefined, bool kSlow, bool kSpinZero, bool kTablesInHbm = false>
__global__ void __launch_bounds__(256, 4) bl_locate_kernel(const BlShadeArgs P) {
  const BlSpacetime st = P.st;
  extern __shared__ double lds_tables[];
  GridTables tab;
  if (kRefined) {
    for (int a = 0; a < 3; a++) {
      tab.xf[a] = tab.xv[a] = nullptr;
      tab.bucket[a] = nullptr;
    }
  } else if (kTablesInHbm) {
    const BlGridDevice &g = P.grid;
    for (int a = 0; a < 3; a++) {
      tab.xf[a] = g.xf[a];
      tab.xv[a] = g.xv[a];
      tab.bucket[a] = g.bucket[a];
    }
  } else {
    const BlGridDevice &g = P.grid;
    double *dst = lds_tables;
    for (int a = 0; a < 3; a++) {
      tab.xf[a] = dst;
      for (int i = threadIdx.x; i <= g.n[a]; i += blockDim.x) dst[i] = g.xf[a][i];
      dst += g.n[a] + 1;
      tab.xv[a] = dst;
      for (int i = threadIdx.x; i < g.n[a]; i += blockDim.x) dst[i] = g.xv[a][i];
      dst += g.n[a];
    }
    unsigned short *bdst = reinterpret_cast<unsigned short *>(dst);
    for (int a = 0; a < 3; a++) {
      tab.bucket[a] = bdst;
      for (int i = threadIdx.x; i < g.n_bucket[a]; i += blockDim.x) bdst[i] = g.bucket[a][i];
      bdst += g.n_bucket[a];
    }
    __syncthreads();
  }
  const unsigned long long n_records = P.counters_in[BL_CNT_RECORDS];
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  unsigned long long gathers_local = 0ull;
  // Position and id of the next record are requested one iteration ahead
  unsigned long long idx = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool more = idx < n_records;
  double2 nq0 = make_double2(0.0, 0.0), nq1 = nq0;
  if (more) {
    const double2 *src = reinterpret_cast<const double2 *>(P.records_hot + (idx) * P.record_stride);
    nq0 = src[0];
    nq1 = src[1];
  }
  while (more) {
    const unsigned long long at = idx;
    const double x1 = nq0.x, x2 = nq0.y, x3 = nq1.x;
    const uint32_t ray = (uint32_t)__double_as_longlong(nq1.y);
    idx += stride;
    more = idx < n_records;
    if (more) {
      const double2 *src = reinterpret_cast<const double2 *>(P.records_hot + (idx) * P.record_stride);
      nq0 = src[0];
      nq1 = src[1];
    }
    if (ray == BL_DEAD_RAY) {
      // kSampleNone: the tolerant coefficient kernel requests corner cells from the tag alone
      if (P.tag_in_record) reinterpret_cast<double2 *>(P.located + at)[1] = make_double2(0.0, 0.0);
      else P.located_tag[at] = 0ull;
      continue;
    }
    double r2;
    const double r = bl_radial_coordinate2<kSpinZero>(st, x1, x2, x3, &r2);
    bool skip = r > P.cuts.camera_r;                                 // simulation_sampling.cpp:238-243
    if (!skip && P.cuts.any_optional) skip = optional_cuts(*P.cold, x1, x2, x3, r);
    LocatedSample loc;
    loc.f_i = loc.f_j = loc.f_k = loc.ph = 0.0;
    loc.cell = 0u;
    loc.status = kSampleCut;
    unsigned long long t_ind = 0ull;
    if (kSlow && !skip && !(P.plasma.fallback_nan && P.ray_flags[ray] != 0)) {   // NaN rays are not sampled (:211-216)
      double t_frac;
      t_ind = (unsigned long long)locate_time(P.slow, P.sample_t[at] + P.slow.snapshot_time, ray, &t_frac);
      P.slow.frac[at] = t_frac;
    }
    if (!skip) locate_sample<kRefined, kSpinZero>(P, tab, st, x1, x2, x3, r, &loc, &gathers_local, P.anchors != nullptr ? P.anchors + at * 8 : nullptr);
    double2 *dst = reinterpret_cast<double2 *>(P.located + at);
    const unsigned long long tag = (t_ind << 40) | ((unsigned long long)loc.status << 32) | loc.cell;
    dst[0] = make_double2(loc.f_i, loc.f_j);
    if (P.tag_in_record) {   // tolerant tier: the tag rides in the azimuth's slot (the few samples the exact kernel re-does
      dst[1] = make_double2(loc.f_k, __longlong_as_double((long long)tag));   // recompute the azimuth): 32 bytes, one stream
    } else {
      dst[1] = make_double2(loc.f_k, loc.ph);
      P.located_tag[at] = tag;
    }
  }
  // S_in accounting: one atomic per wave
  for (int offset = 32; offset > 0; offset >>= 1) gathers_local += __shfl_xor(gathers_local, offset, 64);
  if ((threadIdx.x & 63) == 0 && gathers_local != 0ull) atomicAdd(&P.counters[BL_CNT_GATHERS], gathers_local);
}

// ---- locating a sample in the common case: one grid (or equal blocks merged into one) in spherical Kerr-Schild coordinates
// with its coordinate tables in LDS, trilinear sampling, no optional geometric cut, no slow light. The same functions of the
// same values as locate_sample() - bit-identical results - as one straight line: every lane runs the whole search (a dead
// slot, a cut or an off-grid sample on clamped inputs) and the status is selected at the end, where the general kernel nests
// a dozen divergent branches whose masks, merges and live scalars cost it more instructions than the arithmetic they skip
// (555 vector instructions per sample there, 371 here).
struct PlainGrid {   // what the search needs of the grid, fetched once per workgroup
  GridTables tab;
  int n_i, n_j, n_k, nb_i, nb_j, nb_k;
  bool one_block;
  const double *inv_w[3];   // tolerant tier: 1 / (xv[c + 1] - xv[c]) per axis (LDS), or null: the fractions are IEEE quotients
};
__device__ __forceinline__ void stage_grid_tables(const BlGridDevice &g, double *lds, PlainGrid *pg) {
  double *dst = lds;
  for (int a = 0; a < 3; a++) {
    pg->tab.xf[a] = dst;
    for (int i = threadIdx.x; i <= g.n[a]; i += blockDim.x) dst[i] = g.xf[a][i];
    dst += g.n[a] + 1;
    pg->tab.xv[a] = dst;
    for (int i = threadIdx.x; i < g.n[a]; i += blockDim.x) dst[i] = g.xv[a][i];
    dst += g.n[a];
  }
  unsigned short *bdst = reinterpret_cast<unsigned short *>(dst);
  for (int a = 0; a < 3; a++) {
    pg->tab.bucket[a] = bdst;
    for (int i = threadIdx.x; i < g.n_bucket[a]; i += blockDim.x) bdst[i] = g.bucket[a][i];
    bdst += g.n_bucket[a];
  }
  pg->n_i = g.n[0]; pg->n_j = g.n[1]; pg->n_k = g.n[2];
  pg->nb_i = g.nb[0]; pg->nb_j = g.nb[1]; pg->nb_k = g.nb[2];
  pg->one_block = g.nb[0] == g.n[0] && g.nb[1] == g.n[1] && g.nb[2] == g.n[2];
  pg->inv_w[0] = pg->inv_w[1] = pg->inv_w[2] = nullptr;
}
// ... and the reciprocal widths between cell centres behind them (call after a barrier: reads the staged centres)
__device__ __forceinline__ void stage_reciprocal_widths(const BlGridDevice &g, double *lds, PlainGrid *pg) {
  double *dst = lds;
  for (int a = 0; a < 3; a++) {
    pg->inv_w[a] = dst;
    for (int i = threadIdx.x; i + 1 < g.n[a]; i += blockDim.x) dst[i] = 1.0 / (pg->tab.xv[a][i + 1] - pg->tab.xv[a][i]);
    dst += g.n[a];
  }
}
struct PlainLocated {
  double f_i, f_j, f_k, ph_unwrapped;
  uint32_t status, cell;   // status: kSample...; | kPlainUndecided from locate_plain_sample_tolerant()
};
constexpr uint32_t kPlainUndecided = 0x100u;
// From (r, theta, unwrapped phi) to status, cell and fractions. margin (if asked for): how far theta and phi are from the nearest
// value they are compared with on the way - the faces and the centre of their cells, the ends of the azimuth's range.
// guess_mask (tolerant tier): axes whose cell is guessed as floor((x - x0) / width) instead of searched (evenly spaced faces); a guess
// the faces do not confirm comes back as margin = 0.
__device__ __forceinline__ PlainLocated locate_plain_from_angles(const BlGridDevice &g, const PlainGrid &pg, bool live, bool cut, double r, double th,
                                                                 double ph_unwrapped, double *margin, int guess_mask = 0) {
  const GridTables &tab = pg.tab;
  double ph = ph_unwrapped;
  ph += ph < 0.0 ? 2.0 * kPi : 0.0;
  const double ph_once = ph;
  ph -= ph >= 2.0 * kPi ? 2.0 * kPi : 0.0;
  const double s1 = r, s2 = th, s3 = ph;
  const bool off_grid = s1 < tab.xf[0][0] || s1 > tab.xf[0][pg.n_i] || s2 < tab.xf[1][0] || s2 > tab.xf[1][pg.n_j]
      || s3 < tab.xf[2][0] || s3 > tab.xf[2][pg.n_k];             // :352-394
  const int i = find_cell(g, tab, 0, s1);
  int j, k;
  if (guess_mask & 2) {
    j = (int)((s2 - g.cell_x0[1]) * g.cell_inv_w[1]);
    j = j < 0 ? 0 : (j > pg.n_j - 1 ? pg.n_j - 1 : j);
  } else {
    j = find_cell(g, tab, 1, s2);
  }
  if (guess_mask & 4) {
    k = (int)((s3 - g.cell_x0[2]) * g.cell_inv_w[2]);
    k = k < 0 ? 0 : (k > pg.n_k - 1 ? pg.n_k - 1 : k);
  } else {
    k = find_cell(g, tab, 2, s3);
  }
  // :485-490, per block of a merged grid (one block - the usual case - needs no remainders)
  const int i_b = pg.one_block ? i : i % pg.nb_i, j_b = pg.one_block ? j : j % pg.nb_j, k_b = pg.one_block ? k : k % pg.nb_k;
  const double xv_at_j = tab.xv[1][j], xv_at_k = tab.xv[2][k];
  const int i_m = (i_b == 0 || (i_b != pg.nb_i - 1 && s1 >= tab.xv[0][i])) ? i : i - 1;
  const int j_m = (j_b == 0 || (j_b != pg.nb_j - 1 && s2 >= xv_at_j)) ? j : j - 1;
  const int k_m = (k_b == 0 || (k_b != pg.nb_k - 1 && s3 >= xv_at_k)) ? k : k - 1;
  const double xv_i = tab.xv[0][i_m], xv_j = tab.xv[1][j_m], xv_k = tab.xv[2][k_m];
  const bool sampled = live && !cut && !off_grid;
  PlainLocated out;
  // (fractions of cell widths: ordinary operands for the short division)
  double f_i, f_j, f_k;
  if (pg.inv_w[0] != nullptr) {   // tolerant tier: one multiplication by the width's reciprocal (2e-16 of the fraction)
    f_i = (s1 - xv_i) * pg.inv_w[0][i_m];
    f_j = (s2 - xv_j) * pg.inv_w[1][j_m];
    f_k = (s3 - xv_k) * pg.inv_w[2][k_m];
  } else {
    f_i = blm_div(s1 - xv_i, tab.xv[0][i_m + 1] - xv_i);
    f_j = blm_div(s2 - xv_j, tab.xv[1][j_m + 1] - xv_j);
    f_k = blm_div(s3 - xv_k, tab.xv[2][k_m + 1] - xv_k);
  }
  out.f_i = sampled ? f_i : 0.0;
  out.f_j = sampled ? f_j : 0.0;
  out.f_k = sampled ? f_k : 0.0;
  out.ph_unwrapped = (!live || cut) ? 0.0 : ph_unwrapped;
  out.status = !live ? (uint32_t)kSampleNone : (cut ? (uint32_t)kSampleCut : (off_grid ? (uint32_t)kSampleOffGrid : (uint32_t)kSampleInterp));
  out.cell = sampled ? (uint32_t)((k_m * pg.n_j + j_m) * pg.n_i + i_m) : 0u;
  if (margin != nullptr) {
    const double fj0 = tab.xf[1][j], fj1 = tab.xf[1][j + 1], fk0 = tab.xf[2][k], fk1 = tab.xf[2][k + 1];
    double m = std_min(blm_abs(s2 - fj0), blm_abs(s2 - fj1));
    m = std_min(m, blm_abs(s2 - xv_at_j));
    m = std_min(m, std_min(blm_abs(s3 - fk0), blm_abs(s3 - fk1)));
    m = std_min(m, blm_abs(s3 - xv_at_k));
    m = std_min(m, std_min(blm_abs(ph_unwrapped), blm_abs(ph_once - 2.0 * kPi)));
    // a guessed cell has to hold the coordinate (the search's "first upper face >= x" then names the same cell: the margin above
    // keeps x off the faces); off the grid the guess is the clamped end cell, as the search's
    if ((guess_mask & 2) && !off_grid && !(s2 >= fj0 && s2 <= fj1)) m = 0.0;
    if ((guess_mask & 4) && !off_grid && !(s3 >= fk0 && s3 <= fk1)) m = 0.0;
    *margin = m;
  }
  return out;
}
template <bool kSpinZero>
__device__ __forceinline__ PlainLocated locate_plain_sample(const BlSpacetime &st, const BlGridDevice &g, const PlainGrid &pg, double camera_r,
                                                            bool live, double x1, double x2, double x3) {
  // a dead slot may hold anything: the search runs on a harmless point instead
  x1 = live ? x1 : 1.0;
  x2 = live ? x2 : 1.0;
  x3 = live ? x3 : 1.0;
  double r2;
  const double r = bl_radial_coordinate2<kSpinZero>(st, x1, x2, x3, &r2);
  const bool cut = r > camera_r;                                   // simulation_sampling.cpp:238-243
  // ConvertFromCKS (radiation_geometry.cpp:37-57), as in locate_sample()
  const double th = bl_acos(blm_div(x3, r));
  const double ph_unwrapped = kSpinZero ? bl_atan2(x2, x1) : bl_atan2(x2, x1) - bl_atan(blm_div(st.bh_a, r));
  return locate_plain_from_angles(g, pg, live, cut, r, th, ph_unwrapped, nullptr);
}

// The locate kernel of that case
template <bool kSpinZero>
__global__ void __launch_bounds__(256, 4) bl_locate_plain_kernel(const BlShadeArgs P) {
  const BlSpacetime st = P.st;
  extern __shared__ double lds_tables[];
  PlainGrid pg;
  stage_grid_tables(P.grid, lds_tables, &pg);
  __syncthreads();
  const double camera_r = P.cuts.camera_r;
  const bool tag_in_record = P.tag_in_record != 0;
  const unsigned long long n_records = P.counters_in[BL_CNT_RECORDS];
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  unsigned long long gathers_local = 0ull;
  unsigned long long idx = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool more = idx < n_records;
  double2 nq0 = make_double2(1.0, 1.0), nq1 = make_double2(1.0, __longlong_as_double((long long)BL_DEAD_RAY));
  if (more) {
    const double2 *src = reinterpret_cast<const double2 *>(P.records_hot + (idx) * P.record_stride);
    nq0 = src[0];
    nq1 = src[1];
  }
  while (more) {
    const unsigned long long at = idx;
    const bool live = (uint32_t)__double_as_longlong(nq1.y) != BL_DEAD_RAY;
    const double x1 = nq0.x, x2 = nq0.y, x3 = nq1.x;
    idx += stride;
    more = idx < n_records;
    if (more) {
      const double2 *src = reinterpret_cast<const double2 *>(P.records_hot + (idx) * P.record_stride);
      nq0 = src[0];
      nq1 = src[1];
    }
    const PlainLocated loc = locate_plain_sample<kSpinZero>(st, P.grid, pg, camera_r, live, x1, x2, x3);
    gathers_local += loc.status == kSampleInterp ? 1ull : 0ull;
    const unsigned long long tag = ((unsigned long long)loc.status << 32) | loc.cell;
    double2 *dst = reinterpret_cast<double2 *>(P.located + at);
    if (tag_in_record) {
      // (a dead slot gets its tag alone in the general kernel; the fractions nobody reads are written here as zeros)
      dst[0] = make_double2(loc.f_i, loc.f_j);
      dst[1] = make_double2(loc.f_k, __longlong_as_double((long long)tag));
    } else if (live) {
      dst[0] = make_double2(loc.f_i, loc.f_j);
      dst[1] = make_double2(loc.f_k, loc.ph_unwrapped);
      P.located_tag[at] = tag;
    } else {
      P.located_tag[at] = 0ull;
    }
  }
  for (int offset = 32; offset > 0; offset >>= 1) gathers_local += __shfl_xor(gathers_local, offset, 64);
  if ((threadIdx.x & 63) == 0 && gathers_local != 0ull) atomicAdd(&P.counters[BL_CNT_GATHERS], gathers_local);
}

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
__global__ void __launch_bounds__(256, 2) bl_shade_kernel(const BlShadeArgs P) {
  const BlSpacetime st = P.st;
  const unsigned long long n_all = P.counters_in[BL_CNT_RECORDS];
  const unsigned long long n_listed = kRedo ? P.counters_in[BL_CNT_REDO] : 0ull;
  const bool listed = kRedo && n_listed <= P.redo_capacity;
  const unsigned long long n_records = listed ? n_listed : n_all;   // work items: list entries or records
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  // The record and the located sample of the next iteration are requested at the top of this one, behind
  // this sample's grid reads, so they arrive while the arithmetic runs.
  unsigned long long pos = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= n_records) return;
  unsigned long long idx = listed ? P.redo_list[pos] : pos;
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
      // second pass behind bl_shade_fused_kernel, which leaves no located samples: the few samples it deferred are located
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
      if (!(r_here > P.cuts.camera_r)) locate_sample<false, kSpinZero>(P, tab, st, x1, x2, x3, r_here, &loc, &counted_already, nullptr);
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
        sample_primitives_advanced(P, P.anchors + idx_cur * 8, l0.x, l0.y, l1.x, pr, &kappa_f);
      } else {
        sample_primitives(P, status, (uint32_t)tag, l0.x, l0.y, l1.x, pr);
        if (kExtended && P.plasma.code_kappa) kappa_f = sample_kappa(P, status, (uint32_t)tag, l0.x, l0.y, l1.x);
      }
    }
    pos += stride;
    more = pos < n_records;
    if (more) {
      idx = listed ? P.redo_list[pos] : pos;
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
    if (kAux && !(kPolarized && P.aux_record_unused)) {
      BlAuxSample aux;
      aux.delta_lambda = delta_lambda;
      aux.t = P.sample_t != nullptr ? P.sample_t[idx_cur] : 0.0;
      aux.plane = P.cam_x[1] * x1 + P.cam_x[2] * x2 + P.cam_x[3] * x3;
      aux.length_term = 0.0;
      aux.pad = 0.0;
      if (P.aux_need_length) {
        // unpolarized.cpp:115-129 with the renormalised sample momentum
        double gcov[4][4], gcon[4][4];
        if (st.ray_flat) {
          bl_minkowski(gcov);
          bl_minkowski(gcon);
        } else {
          bl_gcov_ks(ks, gcov);
          bl_gcon_ks(ks, gcon);
        }
        double temp_a[4] = {0.0, 0.0, 0.0, 0.0};
        for (int a = 1; a < 4; a++)
          for (int mu = 0; mu < 4; mu++) temp_a[a] += (gcon[a][mu] - gcon[0][a] * gcon[0][mu] / gcon[0][0]) * kcov[mu];
        double dl_dlambda_sq = 0.0;
        for (int a = 1; a < 4; a++)
          for (int b = 1; b < 4; b++) dl_dlambda_sq += gcov[a][b] * temp_a[a] * temp_a[b];
        aux.length_term = blm_sqrt(dl_dlambda_sq) * delta_lambda * P.x_unit;
      }
      const double nan = __longlong_as_double(0x7ff8000000000000ll);
      for (int a = 0; a < BL_NUM_CELL_VALUES; a++) aux.cell[a] = sh.have_cell ? sh.cell[a] : nan;
      P.aux[row] = aux;
    }
    if (kPolarized) {
      BlPolSample *ps = P.pol_samples + row;
      ps->x[0] = x1; ps->x[1] = x2; ps->x[2] = x3;
      ps->delta_lambda = delta_lambda;
    }
    if (kPolarized) {
      // polarized run (an auxiliary-image, extended, simulation-mode instantiation): the per-frequency formulas (Bessel functions, a dozen powers and exponentials) need few
      // registers and many waves - bl_polarized_coefficients_kernel evaluates them from these scalars
      BlCoefInputs ci;
      if (sh.have_coefficients) {
        ci.nu_fluid_over_nu = sh.nu_fluid_over_nu;
        ci.n_e_cgs = sh.n_e_cgs;
        ci.nu_c_cgs = sh.nu_c_cgs;
        ci.theta_e = sh.theta_e;
        ci.kb_tt_e_cgs = sh.kb_tt_e_cgs;
        ci.cos2_theta_b = sh.cos2_theta_b;
        ci.cos_sign = sh.cos_sign;
        ci.have_coefficients = 1.0;
      } else {
        // cut samples, cells cut or without field: the coefficient code never reached its tetrad, but the polarized transfer
        // needs the frame (with zero velocity / field where the sample was cut). Rare, and a frame's worth of registers:
        // bl_polarized_frame_kernel builds it from what is parked here in the fields nobody reads for such a sample -
        // the renormalised k_mu and the sampled velocity and field.
        ci.nu_fluid_over_nu = kcov[0];
        ci.n_e_cgs = kcov[1];
        ci.nu_c_cgs = kcov[2];
        ci.theta_e = kcov[3];
        ci.kb_tt_e_cgs = __hiloint2double(__float_as_int(pr[3]), __float_as_int(pr[2]));
        ci.cos2_theta_b = __hiloint2double(__float_as_int(pr[5]), __float_as_int(pr[4]));
        ci.cos_sign = __hiloint2double(__float_as_int(pr[7]), __float_as_int(pr[6]));
        ci.have_coefficients = 0.0;
        // ... and listed for that kernel (the list of the tolerant tier's deferred records, unused in polarized runs): one
        // atomic per wave for the lanes that are here; a full list makes the frame kernel scan every record instead
        if (P.redo_list != nullptr) {
          const unsigned long long here = __ballot(1);
          const unsigned int rank = __builtin_amdgcn_mbcnt_hi((unsigned int)(here >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)here, 0u));
          unsigned long long first = 0ull;
          if (rank == 0u) first = atomicAdd(&P.counters[BL_CNT_REDO], (unsigned long long)__popcll(here));
          const unsigned long long at = (((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(first >> 32)) << 32)
                                         | (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)first)) + rank;
          if (at < P.redo_capacity) P.redo_list[at] = idx_cur;
        }
      }
      P.coef_inputs[idx_cur] = ci;
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
        if (kRedo) rec = rec.x == BL_THICK_MARK ? make_double2(0.0, rec.y) : make_double2(rec.x, rec.x * rec.y);   // (a, b) -> (a, c)
        out[l] = rec;
        if (kRedo && P.tau_inc != nullptr) P.tau_inc[(out - P.transfer) + l] = alpha_val * delta_lambda_cgs;   // unpolarized.cpp:150-151
      }
    }
  }
}

// =================================================================================================
// Tolerant arithmetic tier (bl_set_arithmetic(ctx, BL_ARITH_TOLERANT)): the coefficient kernel of plain
// unpolarized images of a spherical Kerr-Schild simulation with thermal electrons - the benchmark's path and the
// many-frequency renders - rewritten inside the tolerance BASELINE.json's north_star grants ("pixel intensities
// match the reference within a stated fp64 tolerance (ray-step counts and termination masks bit-exact)", per-pixel
// L-infinity < 1e-6). The geodesic, locate and transfer kernels, the trilinear read of the primitives, every
// integer / index result and every cut DECISION are those of the exact tier; what changes is the fp64 arithmetic
// between the primitives and the transfer record of a sample:
//   * fused multiply-adds (this section is compiled with fp contract(fast)), reciprocals by v_rcp + Newton steps;
//   * exp / expm1 / cbrt without the bit-reproducibility apparatus of blmath.h (same polynomials; hardware ldexp,
//     frexp, a single-precision seed for the cube root), accurate to a few 1e-16;
//   * the frame algebra of simulation_coefficients.cpp:398-455 in closed form. The reference transforms u^mu and
//     b^mu to Cartesian Kerr-Schild coordinates, builds an orthonormal tetrad and projects k and b on it to get
//     nu_fluid = -k.u and cos^2(theta_B) = (k_a b^a)^2 / (k_a k^a  b_a b^a). Those are invariants: for a null k,
//     sum_a (e_a.k)^2 = (k.u)^2, and b.u = 0 gives sum_a (e_a.b)^2 = b.b, sum_a (e_a.k)(e_a.b) = k.b. So the kernel
//     transforms k_i to the simulation's coordinates instead (three components, Jacobian of radiation_geometry.cpp:
//     69-126, whose entries are x, y, l_i and cot(theta)) and contracts there; b.b = (B.B + (u.B)^2) / (u^t)^2.
//   * the transfer record is (a, c) of I <- a I + c with a = 1 + expm1(-dtau), c = -(j / alpha) expm1(-dtau)
//     (one exponential instead of two; thick: a = 0, c = j / alpha).
// Cut decisions: a value within 1e-9 (relative) of an active cut threshold, or a sample on the polar axis, is not
// decided here - the record goes on a list and bl_shade_kernel<..., kRedo> (exact arithmetic) shades it afterwards.
// The differences to the exact tier are rounding-level (measured ~1e-13 of the image maximum, tests/test_gpu_tolerant.py).
// =================================================================================================
#ifndef BL_FAST_WAVES
#define BL_FAST_WAVES 2
#endif

#pragma clang fp contract(fast)
#include "bl_fastmath.h"

// locate_plain_sample() with the tolerant tier's inverse trigonometric functions (bl_fastmath.h: below 1e-15) where the exact tier has
// the pinned ones. Radius, cut at the camera's sphere, cell search and fractions are the same code on the same tables; a sample
// whose theta or phi comes within `band` (1e-12) of anything it is compared with - a face or centre of its cell, the ends of the azimuth's
// range - is marked kPlainUndecided and left to the exact kernel's second pass, so status and cell are the exact tier's everywhere
// else (and the fractions within 1e-13 of a cell width).
template <bool kSpinZero>
__device__ __forceinline__ PlainLocated locate_plain_sample_tolerant(const BlSpacetime &st, const BlGridDevice &g, const PlainGrid &pg, double camera_r,
                                                                     double band, const double (&acos_c)[14], bool live, double x1, double x2,
                                                                     double x3) {
  x1 = live ? x1 : 1.0;
  x2 = live ? x2 : 1.0;
  x3 = live ? x3 : 1.0;
  double r2;
  const double r = bl_radial_coordinate2<kSpinZero>(st, x1, x2, x3, &r2);
  const bool cut = r > camera_r;
  const double th = fastmath::acos(blm_div(x3, r), acos_c);
  const double ph_unwrapped = kSpinZero ? fastmath::atan2(x2, x1) : fastmath::atan2(x2, x1) - fastmath::atan2(st.bh_a, r);
  double margin;
  PlainLocated out = locate_plain_from_angles(g, pg, live, cut, r, th, ph_unwrapped, &margin, pg.one_block ? (g.uniform_mask & 6) : 0);
  if (live && !cut && !(margin > band)) out.status |= kPlainUndecided;
  return out;
}

// tolerant arithmetic tier of the coefficient formulas (sin, cos, tanh keep the pinned versions: few calls, and their
// arguments need a real range reduction)
#define BLC_NAME(f) f##_fast
#define BLC_SQRT bl_sqrt_g
#define BLC_SQRT_M bl_sqrt_g
#define BLC_CBRT fastmath::cbrt
#define BLC_EXP fastmath::exp
#define BLC_EXPM1 fastmath::expm1
#define BLC_LOG fastmath::log
#define BLC_POW fastmath::pow
#define BLC_POWBASE_T fastmath::PowBase
#define BLC_POW_BASE fastmath::pow_base
#define BLC_POW_OF fastmath::pow_of
#define BLC_DIV_G fastmath::div
#define BLC_SIN bl_sin
#define BLC_COS bl_cos
#define BLC_TANH bl_tanh
#include "bl_coefficients.inc"
#undef BLC_NAME
#undef BLC_SQRT
#undef BLC_SQRT_M
#undef BLC_CBRT
#undef BLC_EXP
#undef BLC_EXPM1
#undef BLC_LOG
#undef BLC_POW
#undef BLC_POWBASE_T
#undef BLC_POW_BASE
#undef BLC_POW_OF
#undef BLC_DIV_G
#undef BLC_SIN
#undef BLC_COS
#undef BLC_TANH

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
          rec = make_double2(0.0, j_total * fastmath::rcp(alpha_val));
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

// The loads of one sample: its located sample (what the gather needs) and its record halves (what the arithmetic needs)
struct FastLocated {
  double2 l0, l1;
  unsigned long long tag;
};
struct FastRay {
  double2 q0, q1, q2, q3;
};
__device__ __forceinline__ void fast_load_located(const BlShadeArgs &P, unsigned long long idx, FastLocated &r) {
  // f_i, f_j | f_k, tag: the locate kernel writes the tag where the exact tier keeps the azimuth (BlShadeArgs::tag_in_record)
  const double2 *loc = reinterpret_cast<const double2 *>(P.located + idx);
  r.l0 = loc[0];
  r.l1 = loc[1];
  r.tag = (unsigned long long)__double_as_longlong(r.l1.y);
}
__device__ __forceinline__ void fast_load_ray(const BlShadeArgs &P, unsigned long long idx, FastRay &r) {
  const double2 *hot = reinterpret_cast<const double2 *>(P.records_hot + (idx) * P.record_stride);
  const double2 *cold = reinterpret_cast<const double2 *>(P.records_cold + (idx) * P.record_stride);
  r.q0 = hot[0]; r.q1 = hot[1]; r.q2 = cold[0]; r.q3 = cold[1];
}

// The corner cells of a located sample, requested (gather_issue) one sample ahead of their use (gather_finish): the two
// halves of sample_primitives(), same operations in the same order. Every load is unconditional - a sample without
// cells to read (cut, off the grid, dead slot) reads cell 0, a nearest-cell sample reads its cell sixteen times - so
// that the number of loads in flight is the same on every path and the compiler's s_waitcnt for an OLDER load (the
// per-ray constants) does not have to wait for these (it counts conservatively across branches).
__device__ __forceinline__ void gather_issue(const BlShadeArgs &P, int status, uint32_t cell, float4 (&lo)[8], float4 (&hi)[8]) {
  const BlGridDevice &g = P.grid;
  const bool interp = status == kSampleInterp;
  const size_t first = (interp || status == kSampleNearest) ? (size_t)cell : 0;
  const float4 *base = reinterpret_cast<const float4 *>(g.cells) + first * 2;
  const size_t row = interp ? (size_t)g.stride_row * 2 : 0, plane = interp ? (size_t)g.stride_plane * 2 : 0, next = interp ? 2 : 0;
#pragma unroll
  for (int corner = 0; corner < 8; corner++) {
    const float4 *p = base + (corner >> 2) * plane + ((corner >> 1) & 1) * row + (corner & 1) * next;
    lo[corner] = p[0];
    hi[corner] = p[1];
  }
}
__device__ __forceinline__ void gather_finish(const BlShadeArgs &P, float fallback_rho, float fallback_pgas, int status, const float4 (&lo)[8],
                                              const float4 (&hi)[8], double f_i, double f_j, double f_k, float pr[8]) {
#pragma clang fp contract(off)
  const BlPlasmaDevice &pl = P.plasma;
  if (status == kSampleInterp) {
    // InterpolateSimple (simulation_sampling.cpp:1334-1351): w_c * v_c summed in the order mmm, mmp, mpm, mpp, pmm, ...
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
          val[q] += w * (double)v[q];
        }
      }
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
}

// The tolerant tier's form of gather_finish(): the same weights, the eight products summed with fused multiply-adds (56 additions
// fewer per sample). The sums differ from the reference's by a few units in the last place of a double, which the conversion to
// float hides - unless a sum lies that close to the midpoint of two floats, where the two could round apart and move a primitive by
// 6e-8: then the function returns true and the sample is left to the exact kernel. The midpoint is where the 29 bits below a
// float's precision read 2^28; the window around it is 64 units for the positive sums of density and pressure, 4 096 for the
// components of velocity and field, whose terms may cancel (a sum that is a 500th of its terms or less is a component that small
// beside the others: a unit of its float precision is 1e-10 of the vector).
__device__ __forceinline__ bool gather_finish_tolerant(const BlShadeArgs &P, float fallback_rho, float fallback_pgas, int status, const float4 (&lo)[8],
                                                       const float4 (&hi)[8], double f_i, double f_j, double f_k, float pr[8]) {
  const BlPlasmaDevice &pl = P.plasma;
  bool near_midpoint = false;
  if (status == kSampleInterp) {
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
      const uint32_t window = q < 2 ? 64u : 4096u;
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
template <bool kSpinZero, bool kGeneral>
__global__ void __launch_bounds__(256, BL_FAST_WAVES) bl_shade_fast_kernel(const BlShadeArgs P) {
  const unsigned long long n_records = P.counters_in[BL_CNT_RECORDS];
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  unsigned long long idx = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;   // record of `next`
  // LDS: 3 x 14 cut thresholds / guard bands, the two fallback primitives, the frequencies and their roots / reciprocals
  extern __shared__ double fast_table[];
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
    const bool near_midpoint = gather_finish_tolerant(P, (float)fast_table[42], (float)fast_table[43], live ? status : (int)kSampleNone, lo, hi,
                                                      loc_prev.l0.x, loc_prev.l0.y, loc_prev.l1.x, pr);
    gather_issue(P, have_cur ? (int)(loc_cur.tag >> 32) & 0xff : (int)kSampleNone, (uint32_t)loc_cur.tag, lo, hi);
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
    if (live) {
      // ReverseGeodesics: sample_len = -geodesic_len (geodesics.cpp:840)
      if (near_midpoint || !fast_shade_sample<kSpinZero, kGeneral>(P, fast_table, pr, status, row, rec.q0.x, rec.q0.y, rec.q1.x, rec.q2.x, rec.q2.y,
                                                                    rec.q3.x, kt, momentum_factor, -rec.q3.y)) {
        fast_defer(P, idx_rec);
      }
    }
  }
}

// Tolerant tier, common case of the grid (locate_plain_sample): the locate step inside the coefficient kernel. The located
// samples - 32 bytes written and 32 read per sample, and the 32 bytes of record the locate kernel reads - never exist: the
// locate kernel alone, at 371 vector instructions per sample, was bound by those 64 bytes per sample (10.9 ms per frame at
// 4.4 TB/s). Three samples in flight per lane:
//   next: its position record is requested before the arithmetic of `prev` and located after it (coordinate tables in LDS);
//   cur:  located -> corner cells and momentum record requested at the top of the iteration;
//   prev: cells and records arrived -> trilinear read, arithmetic.
// One wait per iteration (before the search, on loads a whole sample's arithmetic old). Deferred cut decisions go to the
// exact kernel's second pass, which locates those samples itself (BlShadeArgs::located == nullptr).
template <bool kSpinZero>
__global__ void __launch_bounds__(256, BL_FAST_WAVES) bl_shade_fused_kernel(const BlShadeArgs P) {
  const BlSpacetime st = P.st;
  const unsigned long long n_records = P.counters_in[BL_CNT_RECORDS];
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  unsigned long long idx = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  extern __shared__ double fast_table[];   // bl_shade_fast_kernel's table, then the grid's coordinate tables
  const int table_doubles = 44 + 5 * P.n_nu;
  for (int i = threadIdx.x; i < table_doubles; i += blockDim.x) {
    const BlShadeCold &cc = *P.cold;
    double value;
    if (i < 44) {
      value = i < 14 ? cc.fast_cut[i] : (i < 28 ? cc.fast_cut_lo[i - 14] : (i < 42 ? cc.fast_cut_hi[i - 28]
          : (i == 42 ? (double)cc.fallback_rho : (double)cc.fallback_pgas)));
    } else {
      const int which = (i - 44) / P.n_nu;
      const double f = P.frequencies[(i - 44) - which * P.n_nu];
      const double f_1_3 = fastmath::cbrt(f);
      value = which == 0 ? f : (which == 1 ? bl_sqrt_g(f) : (which == 2 ? f_1_3 : (which == 3 ? bl_sqrt_g(f_1_3) : fastmath::rcp(f))));
    }
    fast_table[i] = value;
  }
  PlainGrid pg;
  stage_grid_tables(P.grid, fast_table + table_doubles, &pg);
  __syncthreads();
  stage_reciprocal_widths(P.grid, fast_table + table_doubles + P.lds_table_bytes / sizeof(double), &pg);
  __syncthreads();
  if (n_records == 0ull) return;
  const unsigned long long last = n_records - 1ull;
  const double camera_r = P.cuts.camera_r;
  const double angle_band = P.fast_angle_band;
  double acos_c[14];   // in registers for the whole loop (bl_fastmath.h) where there is room: the spinning instantiation keeps literals
#pragma unroll
  for (int t = 0; t < 14; t++) acos_c[t] = kSpinZero ? fastmath::opaque_register(fastmath::kAcosCoefficients[t]) : fastmath::kAcosCoefficients[t];
  unsigned long long gathers_local = 0ull;
  // (a position beyond the last record reads the last record and comes back marked dead: whether a slot of the pipeline holds a
  // sample is then a property of its record, not a flag carried beside it - four lane masks fewer across the loop)
  auto load_position = [&](unsigned long long at, double2 &q0, double2 &q1) {
    const bool have = at < n_records;
    const double2 *hot = reinterpret_cast<const double2 *>(P.records_hot + (have ? at : last) * P.record_stride);
    q0 = hot[0];
    q1 = hot[1];
    q1.y = have ? q1.y : __longlong_as_double((long long)BL_DEAD_RAY);
  };
  FastRay rec_prev;                        // q0, q1: position record; q2, q3: momentum record
  double2 hot_cur0, hot_cur1, hot_next0, hot_next1;
  PlainLocated loc_prev, loc_cur;
  float4 lo[8], hi[8];
  unsigned long long idx_prev = ~0ull, idx_cur = idx;
  rec_prev.q0 = rec_prev.q1 = rec_prev.q2 = rec_prev.q3 = make_double2(0.0, 0.0);
  rec_prev.q1.y = __longlong_as_double((long long)BL_DEAD_RAY);
  loc_prev.f_i = loc_prev.f_j = loc_prev.f_k = loc_prev.ph_unwrapped = 0.0;
  loc_prev.status = kSampleNone;
  loc_prev.cell = 0u;
#pragma unroll
  for (int c = 0; c < 8; c++) lo[c] = hi[c] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  load_position(idx_cur, hot_cur0, hot_cur1);
  idx += stride;
  unsigned long long idx_next = idx;
  // (the position record of `next` is requested a whole iteration before its search, so that the wait in front of the search
  // is for loads of the previous iteration, not for the cells and records requested in this one)
  load_position(idx_next, hot_next0, hot_next1);
  idx += stride;
  {
    const bool live = (uint32_t)__double_as_longlong(hot_cur1.y) != BL_DEAD_RAY;
    loc_cur = locate_plain_sample_tolerant<kSpinZero>(st, P.grid, pg, camera_r, angle_band, acos_c, live, hot_cur0.x, hot_cur0.y, hot_cur1.x);
  }
  while (idx_prev < n_records || idx_cur < n_records) {   // (idx_prev starts beyond every record)
    const uint32_t ray = (uint32_t)__double_as_longlong(rec_prev.q1.y);
    const bool live = ray != BL_DEAD_RAY;
    const uint32_t n = (uint32_t)(((unsigned long long)__double_as_longlong(rec_prev.q1.y)) >> 32);
    const int status = (int)(loc_prev.status & 0xffu);
    const bool undecided = (loc_prev.status & kPlainUndecided) != 0u;   // theta or phi too close to a decision: the exact kernel's sample
    // per-ray constants of `prev`: requested before the next sample's cells
    const double kt = P.ray_kt[live ? ray : 0u], momentum_factor = P.ray_factor[live ? ray : 0u];
    const size_t row = (size_t)P.ray_offset[live ? ray : 0u] + n;
    float pr[8];
    const bool near_midpoint = gather_finish_tolerant(P, (float)fast_table[42], (float)fast_table[43], status, lo, hi, loc_prev.f_i, loc_prev.f_j,
                                                      loc_prev.f_k, pr);
    gathers_local += (live && status == kSampleInterp) ? 1ull : 0ull;
    gather_issue(P, (int)(loc_cur.status & 0xffu), loc_cur.cell, lo, hi);
    double2 cold_cur0, cold_cur1;
    {
      const double2 *cold = reinterpret_cast<const double2 *>(P.records_cold + (idx_cur < n_records ? idx_cur : last) * P.record_stride);
      cold_cur0 = cold[0];
      cold_cur1 = cold[1];
    }
    double2 hot_after0, hot_after1;
    load_position(idx, hot_after0, hot_after1);
    if (live) {
      // ReverseGeodesics: sample_len = -geodesic_len (geodesics.cpp:840)
      if (undecided || near_midpoint || !fast_shade_sample<kSpinZero>(P, fast_table, pr, status, row, rec_prev.q0.x, rec_prev.q0.y, rec_prev.q1.x, rec_prev.q2.x,
                                                     rec_prev.q2.y, rec_prev.q3.x, kt, momentum_factor, -rec_prev.q3.y)) {
        fast_defer(P, idx_prev);
      }
    }
    // the search for `next`
    const bool live_next = (uint32_t)__double_as_longlong(hot_next1.y) != BL_DEAD_RAY;
    const PlainLocated loc_next = locate_plain_sample_tolerant<kSpinZero>(st, P.grid, pg, camera_r, angle_band, acos_c, live_next, hot_next0.x, hot_next0.y, hot_next1.x);
    rec_prev.q0 = hot_cur0;
    rec_prev.q1 = hot_cur1;
    rec_prev.q2 = cold_cur0;
    rec_prev.q3 = cold_cur1;
    loc_prev = loc_cur;
    idx_prev = idx_cur;
    hot_cur0 = hot_next0;
    hot_cur1 = hot_next1;
    loc_cur = loc_next;
    idx_cur = idx_next;
    hot_next0 = hot_after0;
    hot_next1 = hot_after1;
    idx_next = idx;
    idx += stride;
  }
  for (int offset = 32; offset > 0; offset >>= 1) gathers_local += __shfl_xor(gathers_local, offset, 64);
  if ((threadIdx.x & 63) == 0 && gathers_local != 0ull) atomicAdd(&P.counters[BL_CNT_GATHERS], gathers_local);
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
          rec_out = make_double2(0.0, j_val * fastmath::rcp(alpha_val));
        }
      } else {
        rec_out = make_double2(1.0, j_val * delta_lambda_cgs);
      }
      out[l] = rec_out;
    }
  }
}

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
      const double2 *in = reinterpret_cast<const double2 *>(P.freq_inputs + (size_t)P.ray_offset[slot]);
      for (int n = num - 1; n >= 0; n--) {   // reference sample order is reversed integration order (geodesics.cpp:832-840)
        const double2 q0 = in[4 * (size_t)n], q1 = in[4 * (size_t)n + 1], q2 = in[4 * (size_t)n + 2], q3 = in[4 * (size_t)n + 3];
        double a = 1.0, c = 0.0;
        if (q0.x == 2.0) {
          c = nan;
        } else if (q0.x == 1.0) {
          // bl_shade_fast_kernel's frequency loop (simulation_coefficients.cpp:464-523, unpolarized.cpp:74-110)
          const double xx_1_3 = q1.x * f_1_3;
          const double var_c = q0.y * f_1_2 + kPow2_11_12 * (q1.y * f_1_6);
          const double j_val = q2.y * f_inv2 * fastmath::exp(-xx_1_3) * var_c * var_c;
          // h nu / (k T_e) is tiny for the hot plasma that shines (Rayleigh-Jeans): the same four-term form as for the thin step
          const double xp = q2.x * f;
          const double planck = xp < 0x1p-10 ? xp * (1.0 + 0.5 * xp * (1.0 + (1.0 / 3.0) * xp * (1.0 + 0.25 * xp))) : fastmath::expm1(xp);
          const double inv_b_nu = planck * (kC * kC / (2.0 * kH));
          double alpha_val = j_val * inv_b_nu;
          if (alpha_val * alpha_val <= 0x1p-1024) alpha_val = 0.0;
          const double delta_lambda_cgs = q3.x * f_inv;
          if (alpha_val > 0.0) {
            const double delta_tau = alpha_val * delta_lambda_cgs;
            if (delta_tau < 0x1p-10) {
              // optically thin step (nearly every sample): expm1(-t) = -t p(t), p = 1 - t/2 (1 - t/3 (1 - t/4)) to 2^-53, so
              // a = 1 - t p and c = -(j / alpha) expm1(-t) = j dl p: no exponential and no division
              const double p = 1.0 - 0.5 * delta_tau * (1.0 - (1.0 / 3.0) * delta_tau * (1.0 - 0.25 * delta_tau));
              a = 1.0 - delta_tau * p;
              c = j_val * delta_lambda_cgs * p;
            } else {
              const double ss = j_val * fastmath::rcp(alpha_val);
              if (delta_tau <= kDeltaTauMax) {
                const double e1 = fastmath::expm1(-delta_tau);
                a = 1.0 + e1;
                c = -ss * e1;
              } else {
                a = 0.0;
                c = ss;
              }
            }
          } else {
            c = j_val * delta_lambda_cgs;
          }
        }
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
// Polarized coefficient kernel: the per-frequency part of CalculateSimulationCoefficients (simulation_coefficients.cpp:
// 458-698) for polarized runs, one sample record per lane from the scalars the coefficient kernel left
// (BlCoefInputs). Writes (j_I, alpha_I) and the three polarized pairs, [ray][n][frequency].
// =================================================================================================
// kTolerant: the tolerant arithmetic tier's functions and fused multiply-adds in the formulas (bl_coefficients.inc,
// second inclusion) - the kernel is almost entirely the double-double pow / log of the pinned library otherwise.
template <bool kTolerant>
__global__ void __launch_bounds__(256, 2) bl_polarized_coefficients_kernel(const BlShadeArgs P) {
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
        if (sh.have_coefficients) simulation_coefficients_fast<true>(P, sh, freq, momentum_factor, &j_val, &alpha_val);
        polarized_coefficients_fast(P, sh, freq, momentum_factor, j_val, alpha_val, pc);
      } else {
        if (sh.have_coefficients) simulation_coefficients<true>(P, sh, freq, momentum_factor, &j_val, &alpha_val);
        polarized_coefficients(P, sh, freq, momentum_factor, j_val, alpha_val, pc);
      }
      P.transfer[at + l] = make_double2(j_val, alpha_val);
      double2 *out = P.pol_coeffs + (at + l) * 3;
      out[0] = pc[0];
      out[1] = pc[1];
      out[2] = pc[2];
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
    if (c3.y != 0.0) continue;   // have_coefficients: the coefficient kernel wrote the frame
    const uint32_t n = (uint32_t)(tag >> 32);
    const double2 q0 = hot[0], c0 = in[0], c1 = in[1], c2 = in[2];
    const double kcov[4] = {c0.x, c0.y, c1.x, c1.y};
    const float uu[3] = {__int_as_float(__double2loint(c2.x)), __int_as_float(__double2hiint(c2.x)), __int_as_float(__double2loint(c2.y))};
    const float bb[3] = {__int_as_float(__double2hiint(c2.y)), __int_as_float(__double2loint(c3.x)), __int_as_float(__double2hiint(c3.x))};
    bl_pol::sample_frame(P.st, P.plasma.simulation_coord, q0.x, q0.y, q1.x, kcov, uu, bb, P.pol_samples + ((size_t)P.ray_offset[ray] + n));
  }
}

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
            if (kAffine) intensity = __builtin_fma(ab[u].x, intensity, ab[u].y);
            else intensity = (ab[u].x == BL_THICK_MARK) ? ab[u].y : ab[u].x * (intensity + ab[u].y);
          }
        }
        for (; n >= 0; n--) {
          double2 ab = rec[(size_t)n * P.n_nu];
          if (kAffine) intensity = __builtin_fma(ab.x, intensity, ab.y);
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
    const double2 *ja = P.transfer + (size_t)P.ray_offset[slot] * P.n_nu;
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
        const double2 c = ja[(size_t)n * P.n_nu + l];
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
    default: break;
  }
  out[i] = r;
}

// =================================================================================================
// Launch wrappers (called from bl_api.hip)
// =================================================================================================
// Start states of the rays [chunk_begin, chunk_begin + chunk_rays) (bl_ray_init_kernel)
extern "C" hipError_t bl_launch_ray_init(const BlTraceArgs *args, int integrator, hipStream_t stream) {
  const bool spin_zero = args->st.bh_a == 0.0;
  const int grid = (args->chunk_rays + 255) / 256;
  const bool dp = integrator == BL_INTEGRATOR_DP;
  if (dp && spin_zero) hipLaunchKernelGGL((bl_ray_init_kernel<true, true>), dim3(grid), dim3(256), 0, stream, *args);
  else if (dp) hipLaunchKernelGGL((bl_ray_init_kernel<true, false>), dim3(grid), dim3(256), 0, stream, *args);
  else if (spin_zero) hipLaunchKernelGGL((bl_ray_init_kernel<false, true>), dim3(grid), dim3(256), 0, stream, *args);
  else hipLaunchKernelGGL((bl_ray_init_kernel<false, false>), dim3(grid), dim3(256), 0, stream, *args);
  return hipGetLastError();
}

// One expression per instantiation of the geodesic kernel (integrator x sample times x zero spin x empty shell; the last two
// only without sample times)
#define BL_GEODESIC_CASES(I, DO)                                                      \
  do {                                                                                \
    if (with_time && spin_zero) DO((bl_geodesic_kernel<I, true, true, false>));       \
    else if (with_time) DO((bl_geodesic_kernel<I, true, false, false>));              \
    else if (spin_zero && shell) DO((bl_geodesic_kernel<I, false, true, true>));      \
    else if (spin_zero) DO((bl_geodesic_kernel<I, false, true, false>));              \
    else if (shell) DO((bl_geodesic_kernel<I, false, false, true>));                  \
    else DO((bl_geodesic_kernel<I, false, false, false>));                            \
  } while (0)

extern "C" hipError_t bl_launch_geodesic(const BlTraceArgs *args, int integrator, int grid, hipStream_t stream) {
  const bool with_time = args->sample_t != nullptr;
  const bool spin_zero = args->st.bh_a == 0.0;   // also true for -0.0: the instantiation never reads bh_a
  const bool shell = args->ray_skipped != nullptr;
  if (shell && with_time) return hipErrorInvalidValue;
#define BL_LAUNCH_G(K) hipLaunchKernelGGL(K, dim3(grid), dim3(64), 0, stream, *args)
  switch (integrator) {
    case BL_INTEGRATOR_DP: BL_GEODESIC_CASES(BL_INTEGRATOR_DP, BL_LAUNCH_G); break;
    case BL_INTEGRATOR_RK4: BL_GEODESIC_CASES(BL_INTEGRATOR_RK4, BL_LAUNCH_G); break;
    default: BL_GEODESIC_CASES(BL_INTEGRATOR_RK2, BL_LAUNCH_G); break;
  }
#undef BL_LAUNCH_G
  return hipGetLastError();
}

// Workgroups (= waves) of the geodesic kernel one CU holds: the persistent grid is this many per CU
extern "C" int bl_geodesic_occupancy(int integrator, int with_time_flag, int spin_zero_flag, int shell_flag) {
  int blocks = 0;
  hipError_t err = hipSuccess;
  const bool with_time = with_time_flag != 0, spin_zero = spin_zero_flag != 0, shell = shell_flag != 0;
#define BL_OCCUPANCY_G(K) err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, K, 64, 0)
  switch (integrator) {
    case BL_INTEGRATOR_DP: BL_GEODESIC_CASES(BL_INTEGRATOR_DP, BL_OCCUPANCY_G); break;
    case BL_INTEGRATOR_RK4: BL_GEODESIC_CASES(BL_INTEGRATOR_RK4, BL_OCCUPANCY_G); break;
    default: BL_GEODESIC_CASES(BL_INTEGRATOR_RK2, BL_OCCUPANCY_G); break;
  }
#undef BL_OCCUPANCY_G
#undef BL_GEODESIC_CASES
  if (err != hipSuccess || blocks < 1) blocks = 4;
  return blocks;
}

// Locate kernel (simulation mode only); lds_bytes = size of the coordinate tables it stages in LDS
extern "C" hipError_t bl_launch_locate(const BlShadeArgs *args, int grid, int lds_bytes, hipStream_t stream) {
  const bool refined = args->grid.n_blocks > 0, slow = args->slow.n > 0;
  const bool spin_zero = args->st.bh_a == 0.0;
  // the common case has a kernel of its own (bl_locate_plain_kernel): same located samples
  const bool plain = !refined && !slow && lds_bytes > 0 && !args->grid.fmks && args->plasma.simulation_interp && !args->cuts.any_optional
      && args->plasma.simulation_coord == BL_COORD_SKS && args->anchors == nullptr && std::getenv("BLACKLIGHT_AMD_GENERAL_LOCATE") == nullptr;
  if (plain) {
    if (spin_zero) hipLaunchKernelGGL((bl_locate_plain_kernel<true>), dim3(grid), dim3(256), lds_bytes, stream, *args);
    else hipLaunchKernelGGL((bl_locate_plain_kernel<false>), dim3(grid), dim3(256), lds_bytes, stream, *args);
    return hipGetLastError();
  }
#define BL_LAUNCH_L(R, S, LDS)                                                                                        \
  do {                                                                                                                \
    if (spin_zero) hipLaunchKernelGGL((bl_locate_kernel<R, S, true>), dim3(grid), dim3(256), LDS, stream, *args);     \
    else hipLaunchKernelGGL((bl_locate_kernel<R, S, false>), dim3(grid), dim3(256), LDS, stream, *args);              \
  } while (0)
  if (refined && slow) BL_LAUNCH_L(true, true, 0);
  else if (refined) BL_LAUNCH_L(true, false, 0);
  else if (lds_bytes == 0) {   // merged grid with tables beyond the LDS budget (not with slow light: its instantiation needs them in LDS)
    if (spin_zero) hipLaunchKernelGGL((bl_locate_kernel<false, false, true, true>), dim3(grid), dim3(256), 0, stream, *args);
    else hipLaunchKernelGGL((bl_locate_kernel<false, false, false, true>), dim3(grid), dim3(256), 0, stream, *args);
  }
  else if (slow) BL_LAUNCH_L(false, true, lds_bytes);
  else BL_LAUNCH_L(false, false, lds_bytes);
#undef BL_LAUNCH_L
  return hipGetLastError();
}

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
    if (aux && args->pol_samples != nullptr && sks_curved && spin_zero)   // polarized run: frame and coefficient inputs per sample, no frequency loop
      hipLaunchKernelGGL((bl_shade_kernel<BL_MODEL_SIMULATION, true, true, true, true, true>), dim3(grid), dim3(256), 0, stream, *args);
    else if (aux && args->pol_samples != nullptr && sks_curved)
      hipLaunchKernelGGL((bl_shade_kernel<BL_MODEL_SIMULATION, true, true, true, true, false>), dim3(grid), dim3(256), 0, stream, *args);
    else if (aux && args->pol_samples != nullptr)
      hipLaunchKernelGGL((bl_shade_kernel<BL_MODEL_SIMULATION, true, true, false, true, false>), dim3(grid), dim3(256), 0, stream, *args);
    else if (aux && power) BL_LAUNCH_S(BL_MODEL_SIMULATION, true, true);
    else if (aux) BL_LAUNCH_S(BL_MODEL_SIMULATION, true, false);
    else if (power) BL_LAUNCH_S(BL_MODEL_SIMULATION, false, true);
    else if (sks_curved && std::getenv("BLACKLIGHT_AMD_UNPIPELINED_SHADE") == nullptr) {   // the benchmark's case: the software-pipelined kernel
      if (spin_zero) hipLaunchKernelGGL((bl_shade_exact_kernel<true>), dim3(grid), dim3(256), 0, stream, *args);
      else hipLaunchKernelGGL((bl_shade_exact_kernel<false>), dim3(grid), dim3(256), 0, stream, *args);
    }
    else if (sks_curved && spin_zero) hipLaunchKernelGGL((bl_shade_kernel<BL_MODEL_SIMULATION, false, false, true, false, true>), dim3(grid), dim3(256), 0, stream, *args);
    else if (sks_curved) hipLaunchKernelGGL((bl_shade_kernel<BL_MODEL_SIMULATION, false, false, true, false, false>), dim3(grid), dim3(256), 0, stream, *args);
    else BL_LAUNCH_S(BL_MODEL_SIMULATION, false, false);
  } else {
    if (aux) BL_LAUNCH_S(BL_MODEL_FORMULA, true, false);
    else BL_LAUNCH_S(BL_MODEL_FORMULA, false, false);
  }
#undef BL_LAUNCH_S
  return hipGetLastError();
}

// Tolerant tier: the fast coefficient kernel, then the exact kernel over the records it deferred
// Tolerant tier in formula mode: the fast kernel, then the exact kernel over the records it deferred
extern "C" hipError_t bl_launch_shade_formula_fast(const BlShadeArgs *args, int grid, hipStream_t stream) {
  hipLaunchKernelGGL(bl_shade_formula_fast_kernel, dim3(grid), dim3(256), 0, stream, *args);
  hipLaunchKernelGGL((bl_shade_kernel<BL_MODEL_FORMULA, false, false, false, false, false, true>), dim3(grid), dim3(256), 0, stream, *args);
  return hipGetLastError();
}

extern "C" hipError_t bl_launch_shade_fast(const BlShadeArgs *args, int grid, hipStream_t stream) {
  if (args->located == nullptr) {   // no locate kernel ran: the fused kernel (coordinate tables in LDS behind its own table)
    const size_t lds = (44 + 5 * args->n_nu) * sizeof(double) + args->lds_table_bytes
        + (size_t)(args->grid.n[0] + args->grid.n[1] + args->grid.n[2]) * sizeof(double);   // (+ the reciprocal widths)
    if (args->st.bh_a == 0.0) {
      hipLaunchKernelGGL((bl_shade_fused_kernel<true>), dim3(grid), dim3(256), lds, stream, *args);
      hipLaunchKernelGGL((bl_shade_kernel<BL_MODEL_SIMULATION, false, false, true, false, true, true>), dim3(grid), dim3(256), 0, stream, *args);
    } else {
      hipLaunchKernelGGL((bl_shade_fused_kernel<false>), dim3(grid), dim3(256), lds, stream, *args);
      hipLaunchKernelGGL((bl_shade_kernel<BL_MODEL_SIMULATION, false, false, true, false, false, true>), dim3(grid), dim3(256), 0, stream, *args);
    }
    return hipGetLastError();
  }
  const size_t lds = (44 + 5 * args->n_nu) * sizeof(double);
  const bool spin_zero = args->st.bh_a == 0.0;
  const bool power_law = args->plasma.power_frac != 0.0 || args->tau_inc != nullptr, cartesian = args->plasma.simulation_coord == BL_COORD_CKS;
  // (an optical-depth image goes through the general instantiation as well)
  // (the records it defers go to the exact kernel: its extended instantiation knows power laws, its general one Cartesian grids)
#define BL_FAST_PAIR(SPIN, GENERAL, EXTENDED, SKS)                                                                               \
  do {                                                                                                                           \
    hipLaunchKernelGGL((bl_shade_fast_kernel<SPIN, GENERAL>), dim3(grid), dim3(256), lds, stream, *args);                        \
    hipLaunchKernelGGL((bl_shade_kernel<BL_MODEL_SIMULATION, false, EXTENDED, SKS, false, SPIN, true>), dim3(grid), dim3(256), 0, \
                       stream, *args);                                                                                           \
  } while (0)
  if (cartesian) {
    if (spin_zero) BL_FAST_PAIR(true, true, true, false); else BL_FAST_PAIR(false, true, true, false);
  } else if (power_law) {
    if (spin_zero) BL_FAST_PAIR(true, true, true, true); else BL_FAST_PAIR(false, true, true, true);
  } else {
    if (spin_zero) BL_FAST_PAIR(true, false, false, true); else BL_FAST_PAIR(false, false, false, true);
  }
#undef BL_FAST_PAIR
  return hipGetLastError();
}

extern "C" hipError_t bl_launch_polarized_coefficients(const BlShadeArgs *args, int grid, hipStream_t stream) {
  if (args->tolerant) hipLaunchKernelGGL(bl_polarized_coefficients_kernel<true>, dim3(grid), dim3(256), 0, stream, *args);
  else hipLaunchKernelGGL(bl_polarized_coefficients_kernel<false>, dim3(grid), dim3(256), 0, stream, *args);
  hipLaunchKernelGGL(bl_polarized_frame_kernel, dim3(grid), dim3(256), 0, stream, *args);
  return hipGetLastError();
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
  if (extended) hipLaunchKernelGGL(bl_coefficients_freq_kernel<true>, dim3(grid), dim3(256), 0, stream, *args);
  else hipLaunchKernelGGL(bl_coefficients_freq_kernel<false>, dim3(grid), dim3(256), 0, stream, *args);
  return hipGetLastError();
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
#pragma unroll
          for (int d = 1; d < kLanes; d <<= 1) {
            const double pa = __shfl_up(a, d, kLanes), pc = __shfl_up(c, d, kLanes);   // the map of the d records before this lane's segment
            if (q >= d) {
              c = __builtin_fma(a, pc, c);
              a *= pa;
            }
          }
          const double block_a = __shfl(a, kLanes - 1, kLanes), block_c = __shfl(c, kLanes - 1, kLanes);
          intensity = __builtin_fma(block_a, intensity, block_c);
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

extern "C" hipError_t bl_launch_transfer(const BlTransferArgs *args, hipStream_t stream) {
  int grid = (int)(((long long)args->chunk_rays * args->n_nu + 255) / 256);
  if (args->affine && args->n_nu == 1 && std::getenv("BLACKLIGHT_AMD_LANE_TRANSFER") == nullptr) {
    hipLaunchKernelGGL(bl_transfer_quad_kernel, dim3((int)(((long long)args->chunk_rays * 4 + 255) / 256)), dim3(256), 0, stream, *args);
    return hipGetLastError();
  }
  if (args->affine) hipLaunchKernelGGL(bl_transfer_kernel<true>, dim3(grid), dim3(256), 0, stream, *args);
  else hipLaunchKernelGGL(bl_transfer_kernel<false>, dim3(grid), dim3(256), 0, stream, *args);
  return hipGetLastError();
}

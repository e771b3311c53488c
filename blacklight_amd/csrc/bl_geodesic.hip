// bl_geodesic.hip - the geodesic stage of the hot path (gfx950).
//
//   bl_ray_init_kernel   one ray per lane: camera pixel -> (x^mu, k_mu), the start state of a ray.   (camera.cpp:528-671)
//   bl_geodesic_kernel   one ray per lane, wave64, persistent waves. Dormand-Prince 5(4) / RK4 / RK2 stepping in Kerr-Schild
//                        coordinates with the reference's controller, dense-output sampling, online truncation test. Lanes
//                        whose ray has terminated are refilled from a global work queue (ballot + popcount prefix, one atomic
//                        per wave); samples go to 64-byte records in per-wave blocks of slots.   (geodesics.cpp:39-396)
//
// The reference integrates camera -> source but evaluates the transfer equation source -> camera (ReverseGeodesics,
// geodesics.cpp:808-849); the coefficient kernels record the transfer coefficients per sample in the forward pass, which keeps
// the recurrence the reference's while never materialising its per-sample arrays.
#include "bl_kernel_util.h"
#include "bl_geodesic_common.h"

namespace {

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
// The lane's index recomputed on the spot (two instructions the optimiser cannot merge with another copy): where the index is
// needed once in a while - a refill, a new block of slots - this costs less than a register that holds it through every step
__device__ __forceinline__ int lane_here() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
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
  const BlSpacetime st = P.st;

  bool have_ray = false;
  bool exhausted = false;
  // per-ray persistent state
  RayState s;
  double k0[8];
  double h_new = 0.0, r_cur = 0.0, r_prev_sample = 0.0;
  // (sample_num doubles as the index of the step's first sample: a ray that goes on has had every accepted step's samples added)
  int num_retry = 0, sample_num = 0, trunc_at = -1;
  int skipped = 0;   // samples of the ray without a record (BlTraceArgs::skip_low)
  int seg = 0;       // BlTraceArgs::segment_rows: segments of kept samples so far (the rows of the ray's composed transfer maps)
  unsigned int slot = 0;
  bool previous_fail = false, flag = false;
  for (int p = 0; p < 8; p++) {
    s.y[p] = 0.0;
    k0[p] = 0.0;
  }
  s.kt = 0.0;
  long long block_next = 0, block_end = 0;   // this wave's block of record slots (wave-uniform)
  // BlTraceArgs::parked as the loop reads it - from LDS: a scalar register held through the steps is one more spilled, and the
  // benchmark's instantiation has no vector register left to spill it to. Words: [0] the buffer, [1] its capacity, [2] low half
  // park_below (-1: no ray is parked, 64: every ray at once), high half park_after, [3] the passes this wave has made since it
  // first found the queue dry (low half) and before (high half). Read through the LDS address space, volatile: a flat load would wait for the
  // sample stores in flight (s_waitcnt vmcnt(0)) in every pass, which a wave alone on its SIMD cannot afford (an eighth of the
  // benchmark frame: 4.1 -> 5.6 ms); a plain load would be hoisted into a register.
  constexpr bool kPark = kIntegrator == BL_INTEGRATOR_DP && !kTime && !kShell;
  __shared__ long long park_lds[5];
  typedef volatile __attribute__((address_space(3))) long long *ParkWord;
  const ParkWord park_word = (ParkWord)park_lds;
  if (kPark && lane_here() == 0) {
    park_lds[0] = reinterpret_cast<long long>(P.parked);
    park_lds[1] = (long long)P.park_capacity;
    // (with the rays predicted long parked beforehand - BlTraceArgs::split_b_hi - this kernel parks nothing: the other stepper has
    // read the list by now)
    const int below = (P.parked == nullptr || P.split_b_hi > 0.0) ? -1 : (P.park_always != 0 ? 64 : (P.park_below < 63 ? P.park_below : 63));
    park_lds[2] = (long long)(((unsigned long long)(unsigned int)P.park_after << 32) | (unsigned long long)(unsigned int)below);
    park_lds[3] = 0;
    park_lds[4] = 0;
  }
  if (kPark) __syncthreads();   // (lane 0's words before any lane reads them; one s_barrier on a one-wave workgroup, executed once)
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
      const int lane = lane_here();
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
          if (kPark && r_cur < 0.0) {
            // a ray bl_split_long_kernel parked before the first step (BlTraceArgs::split_b_hi): bl_geodesic_quad_kernel's; the
            // reservation the leader made for it above goes back (the parked ray holds one of its own)
            have_ray = false;
            atomicAdd(&P.counters[BL_CNT_COMMITTED], (unsigned long long)(-(long long)P.ray_max_steps));
          }
          if (kIntegrator == BL_INTEGRATOR_DP) {
#pragma unroll
            for (int p = 0; p < 8; p++) k0[p] = start[(9 + p) * stride];
          }
          h_new = -P.ray_step * r_cur;
          num_retry = 0;
          previous_fail = false;
          flag = false;
          sample_num = 0;
          skipped = 0;
          seg = 0;
          trunc_at = -1;
          r_prev_sample = 0.0;
        }
      }
    }
    // ------------------------------------------------------------------ park the last rays (BlTraceArgs::parked)
    // A wave that has nothing left to refill its idle lanes from, and few lanes still holding a ray, hands those rays to
    // bl_geodesic_quad_kernel as they stand between two steps and ends.
    if (kPark) {
      const long long word = park_word[2];
      const int park_below = __builtin_amdgcn_readfirstlane((int)word), park_after = __builtin_amdgcn_readfirstlane((int)(word >> 32));
      const unsigned long long holding = __ballot(have_ray);
      bool park = park_below >= 64;
      if (park_below >= 0 && !park) {
        // Whether the queue has run dry: a wave sees it when one of its lanes asks in vain - or, a wave whose lanes all hold
        // long rays (configuration 2: such waves went 48 ms without noticing), by looking at the queue's head every sixteenth
        // pass. That load is dear - the wave waits for it behind all its sample stores, s_waitcnt vmcnt(0): 4 ms of the
        // benchmark frame's 20 when parking is switched on, none when it is off. [3]: low half the passes since the queue ran
        // dry, high half the passes before.
        const long long seen = park_word[3];
        int passes = __builtin_amdgcn_readfirstlane((int)seen);
        const int before = __builtin_amdgcn_readfirstlane((int)(seen >> 32));
        bool dry = passes > 0 || __ballot(exhausted) != 0ull;
        if (!dry && (before & 15) == 15)
          dry = __hip_atomic_load(&P.counters[BL_CNT_NEXT_RAY], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned long long)P.chunk_rays;
        dry = __builtin_amdgcn_readfirstlane(dry ? 1 : 0) != 0;
        park_word[3] = dry ? (long long)(((unsigned long long)(unsigned int)before << 32) | (unsigned int)(passes + 1))
                           : (long long)((unsigned long long)(unsigned int)(before + 1) << 32);   // (every lane writes the same number)
        // [4]: the passes since a lane of this wave last finished a ray (a lane that asked for a new one at the top of this
        // pass has). A wave that goes park_quiet passes without finishing any holds long rays only - a frame's photon-ring rays
        // of thousands of steps, not the benchmark's, whose longest take ~350.
        const int quiet = need_mask != 0ull ? 0 : __builtin_amdgcn_readfirstlane((int)park_word[4]) + 1;
        park_word[4] = (long long)quiet;
        park = dry && (__popcll(holding) <= park_below || passes >= park_after || quiet >= P.park_quiet);
      }
      if (park && holding != 0ull) {
        double *const parked = reinterpret_cast<double *>(park_word[0]);
        const long long park_capacity = park_word[1];
        // (rays that are long so far from the front of the buffer, the others from its back: BlTraceArgs::park_age)
        const bool old_ray = have_ray && sample_num >= P.park_age;
        const unsigned long long old_mask = __ballot(old_ray), young_mask = holding & ~old_mask;
        const int leader = __ffsll((long long)holding) - 1;
        const int lane = lane_here();
        unsigned long long base_old = 0ull, base_young = 0ull;
        if (lane == leader) {
          if (old_mask != 0ull) base_old = atomicAdd(&P.counters[BL_CNT_PARKED], (unsigned long long)__popcll(old_mask));
          if (young_mask != 0ull) base_young = atomicAdd(&P.counters[BL_CNT_PARKED_YOUNG], (unsigned long long)__popcll(young_mask));
        }
        const int old_first = __builtin_amdgcn_readlane((int)base_old, leader), young_first = __builtin_amdgcn_readlane((int)base_young, leader);   // (fewer than 2^31 rays)
        const unsigned long long below = (1ull << lane) - 1ull;
        const long long at = old_ray ? (long long)old_first + __popcll(old_mask & below)
                                     : park_capacity - 1 - ((long long)young_first + __popcll(young_mask & below));
        if (have_ray && at >= 0 && at < park_capacity) {
          double *pk = parked + at * BL_PARK_DOUBLES;
#pragma unroll
          for (int p = 0; p < 8; p++) {
            pk[p] = s.y[p];
            pk[8 + p] = k0[p];
          }
          pk[16] = s.kt;
          pk[17] = h_new;
          pk[18] = r_cur;
          pk[19] = r_prev_sample;
          pk[20] = __longlong_as_double((long long)(((unsigned long long)(unsigned int)sample_num << 32) | (unsigned long long)slot));
          pk[21] = __longlong_as_double((long long)(((unsigned long long)(unsigned int)trunc_at << 32) | (unsigned long long)(unsigned int)num_retry));
          pk[22] = __longlong_as_double((long long)(((unsigned long long)((previous_fail ? 1u : 0u) | (flag ? 2u : 0u)) << 32) | (unsigned long long)(unsigned int)seg));
          // The rows of the ray's kept samples are set aside now - ray_max_steps of them, its reservation in BL_CNT_COMMITTED -
          // so that the coefficient kernel can place the records the ray has left so far before the ray has ended.
          P.ray_offset[slot] = (long long)atomicAdd(&P.counters[BL_CNT_SAMPLES], (unsigned long long)P.ray_max_steps);
          have_ray = false;
        }
      }
    }
    if (__ballot(have_ray) == 0ull) {
      retire_record_slots(P.records_hot, P.record_stride, block_next, block_end, lane_here());   // unused rest of the last block
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
    // (all of these are read only in lanes that have set them in this pass - the coefficients of a midpoint step are set where
    // the step is accepted; the zeros are what a lane without a step holds: constants the compiler rematerialises, where values
    // left over from the last pass would occupy their registers through the six stages)
    for (int p = 0; p < 8; p++) {
      y5[p] = 0.0; k6[p] = 0.0; y4m[p] = 0.0; rv0[p] = 0.0; rv1[p] = 0.0; rv2[p] = 0.0; rv3[p] = 0.0;
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
              // (the IEEE quotient without the range scaling of the compiler's sequence, bl_geometry.h: the scale is >= ray_tol_abs,
              // the difference of two solutions zero or an ordinary number)
              error = std_max(error, bl_div_g(delta_y, error_scale));
            }
          }
          r_new = r_stage;

          // (one evaluation of error^-0.2 for the two places that use it, :200 and :214: a rejected step needs it when the error
          // is finite - it is > 1 then -, an accepted one when the error is > 0 - it is finite then; the function is ~500
          // instructions with its fallback, and the loop's code is what the instruction cache has to hold)
          const double error_power = (error - error == 0.0 && error > 0.0) ? bl_pow_neg_fifth(error) : 0.0;
          if (!(error <= 1.0)) {   // :197-209
            double h_factor = 0.2;
            if (error - error == 0.0) {   // std::isfinite
              double h_factor_ideal = 0.9 * error_power;
              h_factor = std_max(h_factor_ideal, 0.2);
            }
            h_new = h * h_factor;
            num_retry += 1;
            previous_fail = true;
          } else {                 // :210-224
            double h_factor = 10.0;
            if (error > 0.0) {
              h_factor = 0.9 * error_power;
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
            int num_steps_max = P.ray_max_steps - sample_num;
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
            } else {
#pragma unroll
              for (int p = kTime ? 0 : 1; p < 7; p++) {
                rv0[p] = -0.0; rv1[p] = -0.0; rv2[p] = -0.0; rv3[p] = -0.0;
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
        for (int p = 0; p < 7; p++) {
          rv0[p] = -0.0; rv1[p] = -0.0; rv2[p] = -0.0; rv3[p] = -0.0;
        }
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
        // (whole multiples of 64: the number of records allocated so far - where a second pass of a coefficient kernel starts,
        // BlShadeArgs::record_begin_counter - stays aligned with the groups of 16 that the composed maps are cut at)
        const unsigned long long grab = ((unsigned long long)std_max_ll((long long)total - remaining, BL_RECORD_BLOCK) + 63ull) & ~63ull;
        unsigned long long fetched = 0ull;
        if (lane_here() == 63) fetched = atomicAdd(&P.counters[BL_CNT_RECORDS], grab);
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
        const int index = sample_num + nn;
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
        if (!kShell && P.segment_rows) {   // (the instantiation that skips the empty shell has no register left for it: bl_render.hip does not ask)
          // Composed transfer maps (bl_shade_fused2_kernel): the kept samples of a ray that lie side by side within one aligned
          // group of 16 records - a lane's run of a step, cut where it crosses such a boundary - are one SEGMENT, numbered along the
          // ray; the record carries the segment's number instead of the sample's, and the coefficient kernel, whose lanes of a
          // DPP row hold exactly such a group, composes a segment's affine maps and stores one map per segment in row
          // ray_offset + number.
          // (the lane's previous sample of this step sits one slot below, or at the end of the old block: a new group of 16
          // either way when this slot's index is a multiple of 16 or the first of the new block)
          const bool first_of_group = nn == 0 || ((unsigned int)at & 15u) == 0u || place == old_room;
          seg += (!dead && first_of_group) ? 1 : 0;
          hot.n = (unsigned int)(seg - 1);
        }
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
      } else if (sample_num >= P.ray_max_steps) {   // :311-321: the step that reaches ray_max_steps flags the ray and ends it
        flag = true;
        finish = true;
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
      const int rows = (!kShell && P.segment_rows) ? seg : final_num;
      if (!kShell && P.segment_rows) P.ray_rows[slot] = rows;
      P.ray_offset[slot] = (long long)atomicAdd(&P.counters[BL_CNT_SAMPLES], (unsigned long long)rows);
      atomicAdd(&P.counters[BL_CNT_COMMITTED], (unsigned long long)(-(long long)(P.ray_max_steps - (sample_num - (kShell ? skipped : 0)))));
      have_ray = false;
    }
  }
#ifdef BL_GEO_STATS
  atomicAdd(&P.counters[BL_CNT_DEBUG + 0], st_attempt);
  atomicAdd(&P.counters[BL_CNT_DEBUG + 1], st_accept);
  atomicAdd(&P.counters[BL_CNT_DEBUG + 2], st_emit);
  atomicAdd(&P.counters[BL_CNT_DEBUG + 6], st_busy);
  if (lane_here() == 0) {
    atomicAdd(&P.counters[BL_CNT_DEBUG + 3], st_iter);
    atomicAdd(&P.counters[BL_CNT_DEBUG + 4], st_emit_iter);
    atomicAdd(&P.counters[BL_CNT_DEBUG + 5], st_refill);
  }
#endif
}


// Rays predicted long (BlTraceArgs::split_b_lo): one ray of the chunk per lane, after bl_ray_init_kernel and the chunk's counter
// reset, before either stepper. A ray in the band is written to parked[] as it stands before its first step - what
// bl_geodesic_kernel would write with park_always - with its rows and its reservation, and its start state is marked.
__global__ void __launch_bounds__(256) bl_split_long_kernel(BlTraceArgs P) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= (long long)P.chunk_rays) return;
  const long long ray = P.ray_out_index[q];
  const long long pixel = P.pixel_map != nullptr ? (long long)P.pixel_map[ray] : ray;
  double u_ind, v_ind;
  bl_pixel_indices(P.cam, pixel, P.block_locs, &u_ind, &v_ind);
  const double scale = P.st.bh_m * P.cam.camera_width;
  const double b = blm_sqrt(u_ind * u_ind + v_ind * v_ind) * scale;
  if (!(b >= P.split_b_lo && b <= P.split_b_hi)) return;
  if (P.ray_start[q + 8 * P.ray_start_stride] < 0.0) return;   // (marked already: a ray is parked once)
  const long long at = (long long)atomicAdd(&P.counters[BL_CNT_PARKED], 1ull);
  if (at >= (long long)P.park_capacity) return;   // (the buffer holds every ray of the chunk: bl_render.hip)
  double *start = P.ray_start + q;
  const long long stride = P.ray_start_stride;
  double *pk = P.parked + at * BL_PARK_DOUBLES;
  for (int p = 0; p < 7; p++) pk[p] = start[p * stride];
  pk[7] = 0.0;
  for (int p = 0; p < 8; p++) pk[8 + p] = start[(9 + p) * stride];
  const double r = start[8 * stride];
  pk[16] = start[7 * stride];
  pk[17] = -P.ray_step * r;
  pk[18] = r;
  pk[19] = 0.0;
  pk[20] = __longlong_as_double((long long)(unsigned long long)(unsigned int)q);                      // slot, no samples so far
  pk[21] = __longlong_as_double((long long)(((unsigned long long)(unsigned int)-1) << 32));          // truncating sample -1, no retries
  pk[22] = __longlong_as_double(0ll);
  pk[23] = 0.0;
  P.ray_offset[q] = (long long)atomicAdd(&P.counters[BL_CNT_SAMPLES], (unsigned long long)P.ray_max_steps);
  atomicAdd(&P.counters[BL_CNT_COMMITTED], (unsigned long long)P.ray_max_steps);
  start[8 * stride] = -r;
}

// =================================================================================================
// Launch wrappers (called from bl_api.hip)
// =================================================================================================
extern "C" hipError_t bl_launch_split_long(const BlTraceArgs *args, hipStream_t stream) {
  if (args->parked == nullptr || args->ray_start == nullptr) return hipErrorInvalidValue;
  hipLaunchKernelGGL(bl_split_long_kernel, dim3((args->chunk_rays + 255) / 256), dim3(256), 0, stream, *args);
  return hipGetLastError();
}

// Start states of the rays [chunk_begin, chunk_begin + chunk_rays) (bl_ray_init_kernel)
extern "C" hipError_t bl_launch_ray_init(const BlTraceArgs *args, int integrator, hipStream_t stream) {
  const bool spin_zero = args->st.bh_a == 0.0;
  const int grid = (args->chunk_rays + 255) / 256;
  const bool dp = integrator == BL_INTEGRATOR_DP;
  if (dp && spin_zero) hipLaunchKernelGGL((bl_ray_init_kernel<true, true>), dim3(grid), dim3(256), 0, stream, *args);
  else if (dp) hipLaunchKernelGGL((bl_ray_init_kernel<true, false>), dim3(grid), dim3(256), 0, stream, *args);
  else hipLaunchKernelGGL((bl_ray_init_kernel<false, false>), dim3(grid), dim3(256), 0, stream, *args);
  return hipGetLastError();
}

// One expression per instantiation of the geodesic kernel. Dormand-Prince (the default, and what every BASELINE configuration uses):
// sample times x zero spin x empty shell, the last two only without sample times. The fixed-step steppers have the general
// instantiation with and without sample times only (zero spin runs through the general formulas, bit for bit the same; bl_render.hip
// leaves the shell's steps recorded for them).
#define BL_GEODESIC_CASES(I, DO)                                                      \
  do {                                                                                \
    if (with_time && spin_zero) DO((bl_geodesic_kernel<I, true, true, false>));       \
    else if (with_time) DO((bl_geodesic_kernel<I, true, false, false>));              \
    else if (spin_zero && shell) DO((bl_geodesic_kernel<I, false, true, true>));      \
    else if (spin_zero) DO((bl_geodesic_kernel<I, false, true, false>));              \
    else if (shell) DO((bl_geodesic_kernel<I, false, false, true>));                  \
    else DO((bl_geodesic_kernel<I, false, false, false>));                            \
  } while (0)
#define BL_GEODESIC_CASES_FIXED_STEP(I, DO)                                           \
  do {                                                                                \
    if (with_time) DO((bl_geodesic_kernel<I, true, false, false>));                   \
    else DO((bl_geodesic_kernel<I, false, false, false>));                            \
  } while (0)

// lds_pad: bytes of LDS a workgroup (= a wave) reserves without using them - 39 KiB keeps a CU to four waves, one per SIMD, where the
// dispatcher would otherwise fill a CU's eight slots before the next CU's first (BL_TAIL_SPLIT: a CU mask that leaves a shader engine
// one CU and another two makes it do that, and a wave that shares its SIMD steps its rays at half the speed)
extern "C" hipError_t bl_launch_geodesic(const BlTraceArgs *args, int integrator, int grid, hipStream_t stream, int lds_pad) {
  const bool with_time = args->sample_t != nullptr;
  const bool spin_zero = args->st.bh_a == 0.0;   // also true for -0.0: the instantiation never reads bh_a
  const bool shell = args->ray_skipped != nullptr;
  if (shell && (with_time || integrator != BL_INTEGRATOR_DP)) return hipErrorInvalidValue;
#define BL_LAUNCH_G(K) hipLaunchKernelGGL(K, dim3(grid), dim3(64), lds_pad, stream, *args)
  switch (integrator) {
    case BL_INTEGRATOR_DP: BL_GEODESIC_CASES(BL_INTEGRATOR_DP, BL_LAUNCH_G); break;
    case BL_INTEGRATOR_RK4: BL_GEODESIC_CASES_FIXED_STEP(BL_INTEGRATOR_RK4, BL_LAUNCH_G); break;
    default: BL_GEODESIC_CASES_FIXED_STEP(BL_INTEGRATOR_RK2, BL_LAUNCH_G); break;
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
    case BL_INTEGRATOR_RK4: BL_GEODESIC_CASES_FIXED_STEP(BL_INTEGRATOR_RK4, BL_OCCUPANCY_G); break;
    default: BL_GEODESIC_CASES_FIXED_STEP(BL_INTEGRATOR_RK2, BL_OCCUPANCY_G); break;
  }
#undef BL_OCCUPANCY_G
#undef BL_GEODESIC_CASES
#undef BL_GEODESIC_CASES_FIXED_STEP
  if (err != hipSuccess || blocks < 1) blocks = 4;
  return blocks;
}


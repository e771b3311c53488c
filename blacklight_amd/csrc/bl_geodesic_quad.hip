// bl_geodesic_quad.hip - the Dormand-Prince stepper with one ray per QUAD of lanes (gfx950): the last rays of a chunk.
//
//   bl_geodesic_quad_kernel   finishes the rays bl_geodesic_kernel parked (BlTraceArgs::parked, bl_device.h).
//
// Behind a measurement switch (BL_SWITCH_QUAD_TAIL / BL_SWITCH_QUAD_EVERY_RAY): bit-identical to the ray-per-lane kernel, 1.5 - 1.65
// times faster for a ray alone, 2.6 times dearer per ray-step in issue slots - see the end of this comment and docs/notebook.md section 5j.
//
// A ray is a chain of steps, a step a chain of six right-hand sides, and a wave issues its instructions one after the other
// however few of its lanes hold a ray: with a ray per lane the last rays of a chunk - the long rays of the 512^2 formula frame,
// the photon ring's rays of an eighth of the benchmark frame - cost ~6 000 instructions per step each, on SIMDs that have nothing
// else to run. Here the four lanes of a quad share a ray. Lane 0 carries (t, s), lanes 1, 2, 3 carry (x, k_x), (y, k_y),
// (z, k_z): the two components a lane owns go through the stage sums, the 5th/4th-order solutions, the error quotients, the
// midpoint and the dense output as the same instructions on all four lanes; of the right-hand side, each of lanes 1 - 3
// evaluates one row of g^{mu nu} k_nu, one column of the derivatives of (r, f, l), one of the three contractions
// -1/2 d_a g^{mu nu} k_mu k_nu and one row of the proper-distance sum (lane 0: the row of t), the rest - the Kerr-Schild
// scalars, the controller - is evaluated by all four alike. What a lane needs of its neighbours moves by DPP quad permutes
// (VALU, no LDS). Every floating-point operation is one the ray-per-lane kernel performs, on the same operands, in the same
// order within each sum (the contraction and the two distance pieces are the same functions, bl_geometry.h); which lane performs
// it is all that changes, so positions, step lengths, sample counts and flags are the same bits (tests/test_gpu_quad.py: whole
// frames through this kernel with BlTraceArgs::park_always against the same frames without it).   (geodesics.cpp:39-396, as
// bl_geodesic.hip)
//
// What it costs: 4 306 vector instructions in the kernel against 6 660 with zero spin, 4 854 against 7 277 with spin (static; the
// Kerr-Schild scalars with their square roots and reciprocals are evaluated by all four lanes alike, and selecting a lane's operands
// costs 300 - 430 v_cndmask) - two thirds, not the half hoped for. A wave alone on its SIMD issues ~one instruction per 5.3
// cycles whatever it holds, so a lone ray is stepped 1.5 - 1.65 times faster (tools/gpu_quad_latency.py: 49.8 -> 33.3 ms for 64
// rays around the photon ring at a = 0.9); but a wave holds 16 rays instead of 64, so where SIMDs are not idle it is 2.6 times
// dearer per ray-step. With every wave parking its rays once the queue is dry, configuration 2 takes 67 ms instead of 77 and the
// benchmark frame 55.6 instead of 49.2; an eighth of the benchmark frame is unchanged (docs/notebook.md section 5j has the table).
#include "bl_geodesic_common.h"

namespace {

// The value lane kSrc of the quad holds, on all four lanes (every lane of the quad is active wherever this is called: the
// kernel's control flow is uniform within a quad)
template <int kSrc>
__device__ __forceinline__ double quad_broadcast(double v) {
  constexpr int ctrl = kSrc * 0x55;   // quad_perm:[kSrc, kSrc, kSrc, kSrc]
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int kCtrl>
__device__ __forceinline__ double quad_permute(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), kCtrl, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), kCtrl, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int quad_broadcast0(int v) { return __builtin_amdgcn_update_dpp(0, v, 0, 0xf, 0xf, false); }

// Which component pair a lane owns
struct QuadRole {
  bool is0, is1, is2, is3;
};
// of three values, the one of the lane's spatial axis (lane 0: the last, unused)
__device__ __forceinline__ double own_of(const QuadRole &q, double v1, double v2, double v3) { return q.is1 ? v1 : (q.is2 ? v2 : v3); }

// The right-hand side at (x, y, z), (k_t, k_x, k_y, k_z) - the same on all four lanes - for the lane's two components:
//   *d0 = row of g^{mu nu} k_nu: d t (lane 0), d x^a (lanes 1 - 3);  *d1 = d s (lane 0), d k_a (lanes 1 - 3).
// own_pos / own_k: the lane's own x^a and k_a among the arguments. Follows bl_geodesic_rhs<true, kSpinZero> (bl_geometry.h)
// line by line; where that function computes three of a thing, this one computes the lane's.
template <bool kSpinZero>
__device__ __forceinline__ void rhs_quad(const BlSpacetime &st, const QuadRole &q, double x, double y, double z, double kt, double kx, double ky,
                                         double kz, double own_pos, double own_k, double *d0, double *d1, double *r_out) {
  const double kcov[4] = {kt, kx, ky, kz};
  if (st.ray_flat) {
    double acc = 0.0;
    for (int a = 1; a < 4; a++) acc += kcov[a] * kcov[a];
    const double ds = -bl_sqrt_g(acc);
    *d0 = q.is0 ? -kt : own_k;
    *d1 = q.is0 ? ds : 0.0;
    *r_out = bl_radial_coordinate<kSpinZero>(st, x, y, z);
    return;
  }
  const double bh_a = st.bh_a;
  BlKerrSchild ks;
  BlKerrSchildRecip rc;
  bl_kerr_schild_r<kSpinZero>(st, x, y, z, &ks, &rc);
  const double r = ks.r, r2 = ks.r2, f = ks.f, a2 = ks.a2, rr2 = ks.rr2;
  const double *l = ks.l;
  const double *fl = ks.fl;
  *r_out = r;

  // the lane's row of g^{mu nu}: (g^{00}, f l_j) on lane 0, (f l_a, g^{a j}) on the others
  const double g00 = -f - 1.0;
  const double fl_own = own_of(q, fl[0], fl[1], fl[2]);
  double row[3];
  for (int j = 0; j < 3; j++) row[j] = -(fl_own * l[j]);
  {
    const double d0_ = row[0] + 1.0, d1_ = row[1] + 1.0, d2_ = row[2] + 1.0;
    row[0] = q.is1 ? d0_ : row[0];
    row[1] = q.is2 ? d1_ : row[1];
    row[2] = q.is3 ? d2_ : row[2];
  }
  {
    const double lead = q.is0 ? g00 : fl_own;
    double acc = lead * kcov[0];
    for (int j = 0; j < 3; j++) acc += (q.is0 ? fl[j] : row[j]) * kcov[j + 1];
    *d0 = acc;
  }

  // the derivatives along the lane's axis a: d_a r, d_a f, d_a l_i (geodesic_geometry.cpp:199-220)
  double dr, df, dl[3];
  if (kSpinZero) {
    const BlRecip &rc_denom = rc.ra;
    dr = bl_div_r(r * own_pos, rc_denom);
    const double num_f = r2 * r2;
    const BlRecip rc_den_f = bl_recip(r * (r2 * r2));
    df = bl_div_r(-(num_f * dr), rc_den_f) * f;
    const double xl = x - 2.0 * r * l[0];
    const double yl = y - 2.0 * r * l[1];
    const double xd = xl * dr, yd = yl * dr;
    const double xd_r = xd + r, yd_r = yd + r;
    dl[0] = bl_div_r(q.is1 ? xd_r : xd, rc.ra);
    dl[1] = bl_div_r(q.is2 ? yd_r : yd, rc.ra);
    const double mz_r2 = bl_div_r(-z, rc.ra);
    const double zd = mz_r2 * dr;
    const double zd_r = zd + bl_div_r(1.0, rc.r);
    dl[2] = q.is3 ? zd_r : zd;
  } else {
    const BlRecip rc_denom = bl_recip(2.0 * r2 - rr2 + a2);
    const double num = r * own_pos;
    const double num_z = num + bl_div_r(a2 * z, rc.r);
    dr = bl_div_r(q.is3 ? num_z : num, rc_denom);
    const double num_f = r2 * r2 - 3.0 * a2 * z * z;
    const BlRecip rc_den_f = bl_recip(r * (r2 * r2 + a2 * z * z));
    const double t = num_f * dr;
    const double t_z = t + 2.0 * a2 * r * z;
    df = bl_div_r(-(q.is3 ? t_z : t), rc_den_f) * f;
    const double xl = x - 2.0 * r * l[0];
    const double yl = y - 2.0 * r * l[1];
    const double xd = xl * dr, yd = yl * dr;
    const double xd_r = xd + r, xd_a = xd + bh_a;
    const double yd_a = yd - bh_a, yd_r = yd + r;
    dl[0] = bl_div_r(q.is1 ? xd_r : (q.is2 ? xd_a : xd), rc.ra);
    dl[1] = bl_div_r(q.is1 ? yd_a : (q.is2 ? yd_r : yd), rc.ra);
    const double mz_r2 = bl_div_g(-z, r2);
    const double zd = mz_r2 * dr;
    const double zd_r = zd + bl_div_r(1.0, rc.r);
    dl[2] = q.is3 ? zd_r : zd;
  }
  const double dk = bl_momentum_rhs(df, dl, f, l, fl, kcov);

  // proper distance: the lane's row of the sum (:884-887), then the norm of the three rows on every lane (:888-891)
  const BlRecip rc_g00 = bl_recip(g00);
  const double temp_own = bl_distance_row(fl_own, row, fl, g00, rc_g00, kcov);
  const double temp_a[3] = {quad_broadcast<1>(temp_own), quad_broadcast<2>(temp_own), quad_broadcast<3>(temp_own)};
  const double ds = -bl_sqrt_g(bl_distance_norm(fl, l, temp_a));
  *d1 = q.is0 ? ds : dk;
}

}  // namespace

// =================================================================================================
// Geodesic kernel, a ray per quad
// =================================================================================================
template <bool kSpinZero>
__global__ void __launch_bounds__(64, 2) bl_geodesic_quad_kernel(BlTraceArgs P) {
  // With more than one wave per SIMD (BLACKLIGHT_AMD_QUAD_WAVES) the first round of waves - which takes the oldest parked rays,
  // the likely longest - goes first where two waves want the vector unit; the others fill the slots it leaves (a wave of this
  // kernel waits for a dependent result most of the time)
  if ((int)blockIdx.x < P.quad_first_round) __builtin_amdgcn_s_setprio(3);
  const int lane = wave_lane();
  const int role = lane & 3;
  const int quad_first = lane & ~3;
  QuadRole q;
  q.is0 = role == 0;
  q.is1 = role == 1;
  q.is2 = role == 2;
  q.is3 = role == 3;
  const int own1 = q.is0 ? 7 : 3 + role;   // the lane's second component in the ray-per-lane kernel's order t, x, y, z, k_x, k_y, k_z, s
  const BlSpacetime st = P.st;
  // the parked rays: the old ones from the front of the buffer, then the young ones from its back (BlTraceArgs::park_age)
  const unsigned long long n_parked_all = P.counters[BL_CNT_PARKED];
  const long long n_parked_old = (long long)(n_parked_all < (unsigned long long)P.park_capacity ? n_parked_all : (unsigned long long)P.park_capacity);
  const long long n_parked = n_parked_old + (long long)P.counters[BL_CNT_PARKED_YOUNG];   // (together no more than park_capacity)

  bool have_ray = false;
  bool exhausted = false;
  // per-ray state: the lane's two components of y and of the first stage, the scalars of the ray on all four lanes
  double y0 = 0.0, y1 = 0.0, kd0 = 0.0, kd1 = 0.0, kt = 0.0;
  double h_new = 0.0, r_cur = 0.0, r_prev_sample = 0.0;
  int num_retry = 0, sample_num = 0, trunc_at = -1, seg = 0;
  unsigned int slot = 0;
  bool previous_fail = false, flag = false;
  long long block_next = 0, block_end = 0;   // this wave's block of record slots (wave-uniform)

  while (true) {
    // ------------------------------------------------------------------ refill idle quads from the parked rays
    const bool need = !have_ray && !exhausted;
    const unsigned long long need_mask = __ballot(need && q.is0);
    if (need_mask != 0ull) {
      const int count = __popcll(need_mask);
      const int leader = __ffsll((long long)need_mask) - 1;
      unsigned long long base = 0ull;
      if (lane == leader) base = atomicAdd(&P.counters[BL_CNT_QUAD_NEXT], (unsigned long long)count);
      base = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(base >> 32), leader) << 32)
          | (unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)base, leader);
      if (need) {
        const int rank = __popcll(need_mask & ((1ull << quad_first) - 1ull));
        const long long at = (long long)base + rank;
        if (at >= n_parked) {
          exhausted = true;
        } else {
          have_ray = true;
          const double *pk = P.parked + (at < n_parked_old ? at : (long long)P.park_capacity - 1 - (at - n_parked_old)) * BL_PARK_DOUBLES;
          y0 = pk[role];
          y1 = pk[own1];
          kd0 = pk[8 + role];
          kd1 = pk[8 + own1];
          kt = pk[16];
          h_new = pk[17];
          r_cur = pk[18];
          r_prev_sample = pk[19];
          const long long w0 = __double_as_longlong(pk[20]), w1 = __double_as_longlong(pk[21]), w2 = __double_as_longlong(pk[22]);
          slot = (unsigned int)w0;
          sample_num = (int)(w0 >> 32);
          num_retry = (int)(unsigned int)w1;
          trunc_at = (int)(w1 >> 32);
          seg = (int)(unsigned int)w2;
          previous_fail = ((w2 >> 32) & 1) != 0;
          flag = ((w2 >> 32) & 2) != 0;
        }
      }
    }
    if (__ballot(have_ray) == 0ull) {
      retire_record_slots(P.records_hot, P.record_stride, block_next, block_end, lane);
      break;
    }

    // ------------------------------------------------------------------ one step attempt (geodesics.cpp:139-274)
    int emit = 0, num_steps = 0, num_steps_ideal = 1;
    bool accepted = false, finish = false;
    double h = 0.0, r_new = 0.0;
    double a5_0 = 0.0, a5_1 = 0.0, k6_0 = 0.0, k6_1 = 0.0, m0 = 0.0, m1 = 0.0;
    double rv0_0 = -0.0, rv1_0 = -0.0, rv2_0 = -0.0, rv3_0 = -0.0;   // dense-output coefficients of the lane's components;
    double rv0_1 = -0.0, rv1_1 = -0.0, rv2_1 = -0.0, rv3_1 = -0.0;   // -0: a midpoint step (bl_geodesic.hip)
    if (have_ray) {
      if (num_retry > P.ray_max_retries) {   // :139-143
        flag = true;
        finish = true;
      } else {
        h = h_new;
        double K0[7], K1[7];   // stage derivatives of the lane's two components
        double r_stage = 0.0;
        K0[0] = kd0;
        K1[0] = kd1;
        // stages 1..6 (:162-170): y_temp = y + sum_{q<s} a[s][q] * h * k[q], terms added in q order
#define BL_QUAD_STAGE(S)                                                                                      \
        {                                                                                                     \
          double t0 = y0, t1 = y1;                                                                            \
          _Pragma("unroll") for (int qq = 0; qq < S; qq++) {                                                  \
            t0 += kA[S][qq] * h * K0[qq];                                                                     \
            t1 += kA[S][qq] * h * K1[qq];                                                                     \
          }                                                                                                   \
          rhs_quad<kSpinZero>(st, q, quad_broadcast<1>(t0), quad_broadcast<2>(t0), quad_broadcast<3>(t0), kt, \
                              quad_broadcast<1>(t1), quad_broadcast<2>(t1), quad_broadcast<3>(t1), t0, t1, &K0[S], &K1[S], &r_stage); \
        }
        BL_QUAD_STAGE(1)
        BL_QUAD_STAGE(2)
        BL_QUAD_STAGE(3)
        BL_QUAD_STAGE(4)
        BL_QUAD_STAGE(5)
        BL_QUAD_STAGE(6)
#undef BL_QUAD_STAGE
        // 5th / 4th order solutions and error (:173-194); r_new is the r of stage 6 (bl_geodesic.hip)
        double a4_0 = y0, a4_1 = y1;
        a5_0 = y0;
        a5_1 = y1;
#pragma unroll
        for (int qq = 0; qq < 7; qq++) {
          a5_0 += kB5[qq] * h * K0[qq];
          a4_0 += kB4[qq] * h * K0[qq];
          a5_1 += kB5[qq] * h * K1[qq];
          a4_1 += kB4[qq] * h * K1[qq];
        }
        k6_0 = K0[6];
        k6_1 = K1[6];
        // the error norm: t, x, y, z (every lane's first component) and k_x, k_y, k_z (the second of lanes 1 - 3); a maximum of
        // non-negative quotients in which a NaN never wins (std_max keeps its first argument), so the order does not matter
        double error = 0.0;
        {
          const double y_abs = std_max(blm_abs(y0), blm_abs(a5_0));
          const double error_scale = P.ray_tol_abs + P.ray_tol_rel * y_abs;
          const double delta_y = blm_abs(a5_0 - a4_0);
          error = std_max(error, bl_div_g(delta_y, error_scale));
        }
        {
          const double y_abs = std_max(blm_abs(y1), blm_abs(a5_1));
          const double error_scale = P.ray_tol_abs + P.ray_tol_rel * y_abs;
          const double delta_y = blm_abs(a5_1 - a4_1);
          const double with_second = std_max(error, bl_div_g(delta_y, error_scale));
          error = q.is0 ? error : with_second;
        }
        error = std_max(error, quad_permute<0xb1>(error));   // quad_perm:[1, 0, 3, 2]
        error = std_max(error, quad_permute<0x4e>(error));   // quad_perm:[2, 3, 0, 1]
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

          // midpoint (:227-231), subdivision (:234-245)
          m0 = y0;
          m1 = y1;
#pragma unroll
          for (int qq = 0; qq < 7; qq++) {
            m0 += kB4m[qq] * h * K0[qq];
            m1 += kB4m[qq] * h * K1[qq];
          }
          const double r_mid = bl_radial_coordinate<kSpinZero>(st, quad_broadcast<1>(m0), quad_broadcast<2>(m0), quad_broadcast<3>(m0));
          const double delta_s_step = P.ray_step * r_mid;
          const double delta_s_full = quad_broadcast<0>(a5_1) - quad_broadcast<0>(y1);
          num_steps_ideal = (int)ceil(delta_s_full / delta_s_step);
          const int num_steps_max = P.ray_max_steps - sample_num;
          num_steps = num_steps_ideal;
          if (num_steps > num_steps_max) {
            num_steps = num_steps_max;
            flag = true;
          }
          emit = num_steps;
          if (num_steps_ideal > 1) {   // :262-274
            rv0_0 = a5_0 - y0;
            rv1_0 = y0 - a5_0 + h * kd0;
            rv2_0 = 2.0 * (a5_0 - y0) - h * (kd0 + k6_0);
            rv0_1 = a5_1 - y1;
            rv1_1 = y1 - a5_1 + h * kd1;
            rv2_1 = 2.0 * (a5_1 - y1) - h * (kd1 + k6_1);
            double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
            for (int qq = 0; qq < 7; qq++) {
              acc0 += kD[qq] * h * K0[qq];
              acc1 += kD[qq] * h * K1[qq];
            }
            rv3_0 = acc0;
            rv3_1 = acc1;
          }
        }
      }
    }

    // ------------------------------------------------------------------ allocate sample slots (as bl_geodesic.hip; a quad's samples
    // side by side, the quads' runs end to end)
    const int scan = wave_inclusive_scan(q.is0 ? emit : 0);
    const int total = __builtin_amdgcn_readlane(scan, 63);
    const int excl = quad_broadcast0(scan) - emit;
    long long old_base = block_next, new_base = 0;
    int old_room = 0x7fffffff;
    if (total > 0) {
      const long long remaining = block_end - block_next;
      if ((long long)total <= remaining) {
        block_next += total;
      } else {
        old_room = (int)remaining;
        const unsigned long long grab = ((unsigned long long)std_max_ll((long long)total - remaining, BL_RECORD_BLOCK) + 63ull) & ~63ull;
        unsigned long long fetched = 0ull;
        if (lane == 63) fetched = atomicAdd(&P.counters[BL_CNT_RECORDS], grab);
        fetched = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(fetched >> 32), 63) << 32)
            | (unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)fetched, 63);
        new_base = (long long)fetched;
        block_next = new_base + ((long long)total - remaining);
        block_end = (long long)(fetched + grab);
        if (block_end > P.record_capacity) {
          atomicExch(&P.counters[BL_CNT_OVERFLOW], 1ull);
          block_end = block_next = 0;
          emit = 0;
        }
      }
    }

    // ------------------------------------------------------------------ emit samples (:248-293, truncation :327-349 online)
    const int max_emit = wave_max_nonneg(emit);
    const bool dense_output = num_steps_ideal > 1;
    const double base0 = dense_output ? y0 : m0, base1 = dense_output ? y1 : m1;
    const BlRecip rc_steps = bl_recip((double)num_steps_ideal);
    const double len = bl_div_r(h, rc_steps);
    const int word = (role + 3) & 3;   // where the lane's 8 bytes go in either half of a record: x, y, z, (ray, n) | k_x, k_y, k_z, len
    double position = 0.5;
    for (int nn = 0; nn < max_emit; nn++, position += 1.0) {
      if (nn < emit) {
        const double frac = bl_div_r(position, rc_steps);
        const double smp0 = base0 + frac * (rv0_0 + (1.0 - frac) * (rv1_0 + frac * (rv2_0 + (1.0 - frac) * rv3_0)));
        const double smp1 = base1 + frac * (rv0_1 + (1.0 - frac) * (rv1_1 + frac * (rv2_1 + (1.0 - frac) * rv3_1)));
        const int index = sample_num + nn;
        const double r_s = bl_radial_coordinate<kSpinZero>(st, quad_broadcast<1>(smp0), quad_broadcast<2>(smp0), quad_broadcast<3>(smp0));
        const bool hit = index >= 1 && ((r_s > P.camera_r && r_s > r_prev_sample) || r_s < P.r_terminate);
        const bool dead = trunc_at >= 0 || hit;
        trunc_at = (trunc_at < 0 && hit) ? index : trunc_at;
        r_prev_sample = r_s;
        const int place = excl + nn;
        const long long at = place < old_room ? old_base + place : new_base + (place - old_room);
        unsigned int row = (unsigned int)index;
        if (P.segment_rows) {   // composed transfer maps: the record carries its segment's number (bl_geodesic.hip)
          const bool first_of_group = nn == 0 || ((unsigned int)at & 15u) == 0u || place == old_room;
          seg += (!dead && first_of_group) ? 1 : 0;
          row = (unsigned int)(seg - 1);
        }
        const unsigned long long id = ((unsigned long long)row << 32) | (unsigned long long)(dead ? BL_DEAD_RAY : slot);
        double *hot = reinterpret_cast<double *>(P.records_hot + at * P.record_stride);
        double *cold = reinterpret_cast<double *>(P.records_cold + at * P.record_stride);
        hot[word] = q.is0 ? __longlong_as_double((long long)id) : smp0;
        cold[word] = q.is0 ? len : smp1;
      }
    }

    // ------------------------------------------------------------------ finish the step
    if (have_ray && accepted) {
      // renormalise the spatial momentum at the new point (:296-309)
      const double factor = bl_renormalization_factor<kSpinZero>(st, quad_broadcast<1>(a5_0), quad_broadcast<2>(a5_0), quad_broadcast<3>(a5_0), kt,
                                                                 quad_broadcast<1>(a5_1), quad_broadcast<2>(a5_1), quad_broadcast<3>(a5_1));
      const double scaled = a5_1 * factor;
      a5_1 = q.is0 ? a5_1 : scaled;
      const double r_before = r_cur;
      sample_num += num_steps;
      const bool terminate_outer = r_new > P.camera_r && r_new > r_before;
      const bool terminate_inner = r_new < P.r_terminate;
      if (terminate_outer || terminate_inner) {
        finish = true;
      } else if (sample_num >= P.ray_max_steps) {   // :311-321
        flag = true;
        finish = true;
      }
      // FSAL (:149-154)
      y0 = a5_0;
      y1 = a5_1;
      kd0 = k6_0;
      kd1 = k6_1;
      r_cur = r_new;
    }
    if (have_ray && finish) {
      if (q.is0) {
        const int final_num = (trunc_at >= 0) ? trunc_at : sample_num;
        P.ray_sample_num[slot] = final_num;
        P.ray_flags[slot] = flag ? 1 : 0;
        const int rows = P.segment_rows ? seg : final_num;
        if (P.segment_rows) P.ray_rows[slot] = rows;
        // (the ray's rows were set aside when it was parked: ray_offset[slot])
        atomicAdd(&P.counters[BL_CNT_COMMITTED], (unsigned long long)(-(long long)(P.ray_max_steps - sample_num)));
      }
      have_ray = false;
    }
  }
}

// The parked rays of a chunk (BL_CNT_PARKED of them, known on the device only): a grid of waves that take them sixteen at a time
// (lds_pad: as for bl_launch_geodesic)
extern "C" hipError_t bl_launch_geodesic_quad(const BlTraceArgs *args, int grid, hipStream_t stream, int lds_pad) {
  if (args->parked == nullptr || args->sample_t != nullptr || args->ray_skipped != nullptr) return hipErrorInvalidValue;
  if (args->st.bh_a == 0.0) hipLaunchKernelGGL((bl_geodesic_quad_kernel<true>), dim3(grid), dim3(64), lds_pad, stream, *args);
  else hipLaunchKernelGGL((bl_geodesic_quad_kernel<false>), dim3(grid), dim3(64), lds_pad, stream, *args);
  return hipGetLastError();
}

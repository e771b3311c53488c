/* bl_fastmath.h - elementary functions of the tolerant arithmetic tier (bl_set_arithmetic(BL_ARITH_TOLERANT)): hardware
 * reciprocal / reciprocal square root with Newton steps, one exponential core for exp and expm1, cube root, logarithm, powers
 * of one base, K_0 / K_1 / K_2. Accurate to a few ulp (tests/test_gpu_tolerant.py::test_tolerant_functions_are_accurate), not
 * bit-reproducible by contract. Device only; to be included INSIDE a `#pragma clang fp contract(fast)` region of the including
 * file (bl_sampling_fast.h, bl_shade_fast.hip, bl_transfer.hip, bl_polarized.hip): the fused multiply-adds are part of the design. */
#ifndef BLACKLIGHT_AMD_BL_FASTMATH_H_
#define BLACKLIGHT_AMD_BL_FASTMATH_H_

#include "blmath.h"

namespace fastmath {

// FMK(c): a 64-bit literal of this header. BL_FAST_SCALAR_LITERALS: as a scalar operand (blmath.h BLM_K) instead of two v_mov_b32
#ifdef BL_FAST_SCALAR_LITERALS
#define FMK(c) blm_scalar_literal(c)
#else
#define FMK(c) (c)
#endif

// 1 / b: v_rcp_f64 (about 26 good bits) + one Newton step, relative error <= 2.1e-15 - far inside the tier's 1e-6 and
// six orders inside the guard band of the cut decisions; a second step (1.1e-16) costs two more multiply-adds per
// reciprocal, 0.7 ms of the benchmark frame. v_div_fixup restores 1 / 0 = inf, 1 / inf = 0 and NaN.
__device__ __forceinline__ double rcp(double b) {
  double y = __builtin_amdgcn_rcp(b);
  const double e = __builtin_fma(-b, y, 1.0);
  y = __builtin_fma(y, e, y);
  return __builtin_amdgcn_div_fixup(y, b, 1.0);
}
// 1 / sqrt(x) for finite x > 0
__device__ __forceinline__ double rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double h = 0.5 * x;
  double e = __builtin_fma(-h * y, y, 0.5);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-h * y, y, 0.5);
  return __builtin_fma(y, e, y);
}
// sqrt(x) for finite x >= 0 (a few units in the last place; 0 for 0)
__device__ __forceinline__ double sqrt(double x) { return x > 0.0 ? x * rsqrt(x) : 0.0; }
// exp(x) and expm1(x) share one core: x = k ln 2 + r, |r| <= ln 2 / 2, and e = expm1(r) = r + r^2 / 2 + r^3 q(r) with the
// polynomial of bl_expm1 (blmath.h); hardware rounding and ldexp. One set of coefficients for both keeps two dozen
// scalar registers free (a v_fma_f64 cannot take a 64-bit literal: every coefficient is a register pair).
__device__ __forceinline__ double expm1_core(double x, int *k) {
  const double kd = __builtin_rint(x * FMK(0x1.71547652b82fep+0));
  double r = __builtin_fma(-kd, FMK(0x1.62e42feep-1), x);
  r = __builtin_fma(-kd, FMK(0x1.a39ef35793c76p-33), r);
  double q = FMK(0x1.94328fcb8199cp-37);
  q = __builtin_fma(q, r, FMK(0x1.61bfaa228dde5p-33));
  q = __builtin_fma(q, r, FMK(0x1.1eed7a01fc8b7p-29));
  q = __builtin_fma(q, r, FMK(0x1.ae642c82e33d5p-26));
  q = __builtin_fma(q, r, FMK(0x1.27e4fb7a2782ap-22));
  q = __builtin_fma(q, r, FMK(0x1.71de3a5aa7bb7p-19));
  q = __builtin_fma(q, r, FMK(0x1.a01a01a019b63p-16));
  q = __builtin_fma(q, r, FMK(0x1.a01a01a0196acp-13));
  q = __builtin_fma(q, r, FMK(0x1.6c16c16c16c17p-10));
  q = __builtin_fma(q, r, FMK(0x1.1111111111111p-7));
  q = __builtin_fma(q, r, FMK(0x1.5555555555555p-5));
  q = __builtin_fma(q, r, FMK(0x1.5555555555555p-3));
  const double r2 = r * r;
  *k = (int)kd;
  return r + __builtin_fma(r2 * r, q, 0.5 * r2);
}
// ---- 64-bit constants as scalar operands. No fp64 vector instruction takes a 64-bit literal: a constant has to sit in a register
// pair, and the compiler's choice is two v_mov_b32 into vector registers in front of every use - for the addend of a Horner step
// always, because it prefers the two-address v_fmac_f64 and copies the constant into the destination first. fma_k / add_k are the
// three-address forms with the constant as the scalar operand (a literal handed in becomes two s_mov_b32; a value the caller keeps
// in scalar registers across a loop - resident_constant() - costs nothing per use).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double fma_k(double a, double b, double c_uniform) {   // a * b + c
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c_uniform));
  return d;
}
__device__ __forceinline__ double add_k(double a, double c_uniform) {   // a + c
  double d;
  asm("v_add_f64 %0, %1, %2" : "=v"(d) : "v"(a), "s"(c_uniform));
  return d;
}
__device__ __forceinline__ double resident_constant(double v) {   // a literal the optimiser can neither fold nor rematerialise
  asm volatile("" : "+s"(v));
  return v;
}
#else
__device__ __forceinline__ double fma_k(double a, double b, double c_uniform) { return __builtin_fma(a, b, c_uniform); }
__device__ __forceinline__ double add_k(double a, double c_uniform) { return a + c_uniform; }
__device__ __forceinline__ double resident_constant(double v) { return v; }
#endif
// expm1_core() with its fifteen constants handed in (kExpConstants through resident_constant(): a loop that evaluates one
// exponential per iteration keeps them in scalar registers): the same operations, bit-identical results
constexpr double kExpConstants[15] = {0x1.71547652b82fep+0, 0x1.62e42feep-1, 0x1.a39ef35793c76p-33, 0x1.94328fcb8199cp-37, 0x1.61bfaa228dde5p-33,
                                      0x1.1eed7a01fc8b7p-29, 0x1.ae642c82e33d5p-26, 0x1.27e4fb7a2782ap-22, 0x1.71de3a5aa7bb7p-19, 0x1.a01a01a019b63p-16,
                                      0x1.a01a01a0196acp-13, 0x1.6c16c16c16c17p-10, 0x1.1111111111111p-7, 0x1.5555555555555p-5, 0x1.5555555555555p-3};
__device__ __forceinline__ double expm1_core(double x, int *k, const double (&c)[15]) {
  const double kd = __builtin_rint(x * c[0]);
  double r = __builtin_fma(-kd, c[1], x);
  r = __builtin_fma(-kd, c[2], r);
  double q = add_k(r * c[3], c[4]);
#pragma unroll
  for (int i = 5; i < 15; i++) q = fma_k(q, r, c[i]);
  const double r2 = r * r;
  *k = (int)kd;
  return r + __builtin_fma(r2 * r, q, 0.5 * r2);
}
__device__ __forceinline__ double exp(double x, const double (&c)[15]) {
  x = x > 710.0 ? 710.0 : (x < -746.0 ? -746.0 : x);
  int k;
  const double e = expm1_core(x, &k, c);
  return __builtin_amdgcn_ldexp(1.0 + e, k);
}
// exp(x); x beyond the range of doubles saturates to 0 / inf
__device__ __forceinline__ double exp(double x) {
  x = x > 710.0 ? 710.0 : (x < -746.0 ? -746.0 : x);
  int k;
  const double e = expm1_core(x, &k);
  return __builtin_amdgcn_ldexp(1.0 + e, k);
}
// expm1(x); accurate for tiny |x| (k = 0: the core itself)
__device__ __forceinline__ double expm1(double x) {
  x = x > 710.0 ? 710.0 : (x < -40.0 ? -40.0 : x);
  int k;
  const double e = expm1_core(x, &k);
  const double t = __builtin_amdgcn_ldexp(1.0, k);   // 2^k; inf for k = 1024 (x > 709.78)
  return (t - 1.0) + t * e;
}
// cbrt(x) for x >= 0 (0, inf and NaN pass through): x = m 2^(3q), m in [0.5, 4); m^(-1/3) from a single-precision
// seed and two Newton steps z <- z + z (1 - m z^3) / 3, then m^(1/3) = m z^2 with one correction
__device__ __forceinline__ double cbrt(double x) {
  const int e = __builtin_amdgcn_frexp_exp(x);
  const double mant = __builtin_amdgcn_frexp_mant(x);              // [0.5, 1)
  const int q = (int)(((unsigned int)(e + 3072) * 43691u) >> 17) - 1024;   // floor(e / 3) for |e| < 3072
  const double m = __builtin_amdgcn_ldexp(mant, e - 3 * q);        // [0.5, 4)
  const float seed = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf((float)m) * -0.33333334f);
  double z = (double)seed;
  double h = __builtin_fma(-m, z * z * z, 1.0);
  z = __builtin_fma(z * h, FMK(0x1.5555555555555p-2), z);
  h = __builtin_fma(-m, z * z * z, 1.0);
  z = __builtin_fma(z * h, FMK(0x1.5555555555555p-2), z);
  double c = m * z * z;                                             // m^(1/3), ~2 ulp
  c = __builtin_fma(__builtin_fma(-c * c, c, m), z * z * FMK(0x1.5555555555555p-2), c);   // c + (m - c^3) / (3 c^2)
  const double res = __builtin_amdgcn_ldexp(c, q);
  return __builtin_amdgcn_class(x, 0x263) ? x : res;               // NaN, +-0, +inf
}

// log(x): x = m 2^e, m in [sqrt(1/2), sqrt(2)); log m = 2 s (1 + z / 3 + z^2 / 5 + ...), s = (m - 1) / (m + 1), z = s^2 <= 0.0295
// (ten terms: 1e-17). Zero gives -inf, negative arguments NaN, +inf and NaN pass through.
__device__ __forceinline__ double log(double x) {
  int e = __builtin_amdgcn_frexp_exp(x);
  double m = __builtin_amdgcn_frexp_mant(x);   // [0.5, 1)
  const bool low = m < FMK(0x1.6a09e667f3bcdp-1);
  m = low ? 2.0 * m : m;
  e = low ? e - 1 : e;
  const double s = (m - 1.0) * rcp(m + 1.0);
  const double z = s * s;
  double p = FMK(1.0 / 19.0);
  p = __builtin_fma(p, z, FMK(1.0 / 17.0));
  p = __builtin_fma(p, z, FMK(1.0 / 15.0));
  p = __builtin_fma(p, z, FMK(1.0 / 13.0));
  p = __builtin_fma(p, z, FMK(1.0 / 11.0));
  p = __builtin_fma(p, z, FMK(1.0 / 9.0));
  p = __builtin_fma(p, z, FMK(1.0 / 7.0));
  p = __builtin_fma(p, z, FMK(1.0 / 5.0));
  p = __builtin_fma(p, z, FMK(1.0 / 3.0));
  const double ed = (double)e;
  const double small = __builtin_fma(2.0 * s * z, p, ed * FMK(0x1.a39ef35793c76p-33));   // 2 s (z / 3 + ...) + e ln2_lo
  const double res = __builtin_fma(ed, FMK(0x1.62e42feep-1), 2.0 * s + small);
  // special arguments: +-0 -> -inf, negative -> NaN, +inf / NaN -> themselves
  if (__builtin_amdgcn_class(x, 0x060)) return -__builtin_inf();
  if (__builtin_amdgcn_class(x, 0x01c)) return __builtin_nan("");
  if (__builtin_amdgcn_class(x, 0x203)) return x;
  return res;
}
// pow(x, y) = exp(y log x) for x >= 0 (the coefficient formulas raise positive quantities to real powers); relative error
// ~|y log x| 2^-52. pow(x, 0) = 1 and pow(1, y) = 1 whatever the other argument, as std::pow has it.
__device__ __forceinline__ double pow(double x, double y) {
  const double r = exp(y * log(x));
  return (y == 0.0 || x == 1.0) ? 1.0 : r;
}
struct PowBase {
  double x, l;
};
__device__ __forceinline__ PowBase pow_base(double x) { return PowBase{x, log(x)}; }
__device__ __forceinline__ double pow_of(PowBase b, double y) {
  const double r = exp(y * b.l);
  return (y == 0.0 || b.x == 1.0) ? 1.0 : r;
}
__device__ __forceinline__ double div(double a, double b) { return a * rcp(b); }

// acos(x) for |x| <= 1 and atan2(y, x) for finite arguments, as the locate step of the tolerant tier uses them: the pinned
// library's polynomials (blmath.h: (asin(s) - s) / s^3 on s^2 <= 1/4; atan on |u| < 7/16) without its double-double corrections,
// with the tier's reciprocal and reciprocal square root, and one quotient in atan2 where bl_atan2 has two. Absolute error
// below 1e-15 (measured 6e-16 / 7e-16, tests/test_gpu_tolerant.py); the caller keeps away from decisions closer than that (locate_plain_sample).
__device__ __forceinline__ double acos(double x) {
  const double ax = __builtin_fabs(x);
  const bool small = ax < 0.5;
  const double z = small ? x * x : (1.0 - ax) * 0.5;
  double r = 0x1.e58a4f278e007p-6;
  r = __builtin_fma(r, z, -0x1.3bd7e353ddbc2p-6);
  r = __builtin_fma(r, z, 0x1.40c91fa8deb7ep-6);
  r = __builtin_fma(r, z, 0x1.8dcdf11997e0fp-9);
  r = __builtin_fma(r, z, 0x1.31777489dfd29p-7);
  r = __builtin_fma(r, z, 0x1.3b462d121c5d2p-7);
  r = __builtin_fma(r, z, 0x1.7b02ef007d23ep-7);
  r = __builtin_fma(r, z, 0x1.c990a42b32b03p-7);
  r = __builtin_fma(r, z, 0x1.1c4efd20ebb99p-6);
  r = __builtin_fma(r, z, 0x1.6e8ba121b9d5fp-6);
  r = __builtin_fma(r, z, 0x1.f1c71c7a5e151p-6);
  r = __builtin_fma(r, z, 0x1.6db6db6dac0eap-5);
  r = __builtin_fma(r, z, 0x1.3333333333389p-4);
  r = __builtin_fma(r, z, 0x1.5555555555555p-3);
  r *= z;                                                     // asin(s) = s + s r
  const double s = small ? x : (z > 0.0 ? z * rsqrt(z) : 0.0);
  const double asin_s = __builtin_fma(s, r, s);
  const double pio2 = 0x1.921fb54442d18p+0, pio2_lo = 0x1.1a62633145c07p-54;
  const double res_small = (pio2 - asin_s) + pio2_lo;
  const double res_neg = __builtin_fma(-2.0, asin_s, 2.0 * pio2) + 2.0 * pio2_lo;
  return small ? res_small : (x < 0.0 ? res_neg : 2.0 * asin_s);
}
// ... with the polynomial's coefficients handed in: a kernel that keeps them in registers across its sample loop (kAcosCoefficients
// through opaque_register(), bl_shade_fused_kernel) saves the two moves per coefficient that a literal costs at every use
constexpr double kAcosCoefficients[14] = {0x1.e58a4f278e007p-6, -0x1.3bd7e353ddbc2p-6, 0x1.40c91fa8deb7ep-6, 0x1.8dcdf11997e0fp-9,
                                          0x1.31777489dfd29p-7, 0x1.3b462d121c5d2p-7, 0x1.7b02ef007d23ep-7, 0x1.c990a42b32b03p-7,
                                          0x1.1c4efd20ebb99p-6, 0x1.6e8ba121b9d5fp-6, 0x1.f1c71c7a5e151p-6, 0x1.6db6db6dac0eap-5,
                                          0x1.3333333333389p-4, 0x1.5555555555555p-3};
__device__ __forceinline__ double opaque_register(double v) {
  asm volatile("" : "+v"(v));
  return v;
}
template <int N>
__device__ __forceinline__ double acos(double x, const double (&c)[N]) {
  const double ax = __builtin_fabs(x);
  const bool small = ax < 0.5;
  const double z = small ? x * x : (1.0 - ax) * 0.5;
  double r = c[0];
#pragma unroll
  for (int i = 1; i < 14; i++) r = __builtin_fma(r, z, c[i]);
  r *= z;
  const double s = small ? x : (z > 0.0 ? z * rsqrt(z) : 0.0);
  const double asin_s = __builtin_fma(s, r, s);
  const double pio2 = 0x1.921fb54442d18p+0, pio2_lo = 0x1.1a62633145c07p-54;
  const double res_small = (pio2 - asin_s) + pio2_lo;
  const double res_neg = __builtin_fma(-2.0, asin_s, 2.0 * pio2) + 2.0 * pio2_lo;
  return small ? res_small : (x < 0.0 ? res_neg : 2.0 * asin_s);
}
__device__ __forceinline__ double atan2(double y, double x) {
  const double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
  const double mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
  // t = mn / mx in [0, 1]: atan(t) = atan(c) + atan((t - c) / (1 + c t)) with c = 0, 1/2, 1 by t < 7/16, < 11/16, else
  const bool low = 16.0 * mn <= 7.0 * mx, mid = 16.0 * mn < 11.0 * mx;   // (<=: the origin itself gives 0)
  const double c = low ? 0.0 : (mid ? 0.5 : 1.0);
  const double hi = low ? 0.0 : (mid ? 0x1.dac670561bb4fp-2 : 0x1.921fb54442d18p-1);
  const double num = __builtin_fma(-c, mx, mn), den = __builtin_fma(c, mn, mx);
  const double u = den > 0.0 ? num * rcp(den) : 0.0;
  const double w = u * u;
  double a = -0x1.9a0e3d8214a3cp-7;
  a = __builtin_fma(a, w, 0x1.dde84abd3489ap-6);
  a = __builtin_fma(a, w, -0x1.4ac01ab40659fp-5);
  a = __builtin_fma(a, w, 0x1.812cf294b38a9p-5);
  a = __builtin_fma(a, w, -0x1.ae800c7915a0cp-5);
  a = __builtin_fma(a, w, 0x1.e1d239c838f12p-5);
  a = __builtin_fma(a, w, -0x1.1110907ae84ccp-4);
  a = __builtin_fma(a, w, 0x1.3b13abac1919fp-4);
  a = __builtin_fma(a, w, -0x1.745d171e2e854p-4);
  a = __builtin_fma(a, w, 0x1.c71c71c673bd9p-4);
  a = __builtin_fma(a, w, -0x1.24924924918e2p-3);
  a = __builtin_fma(a, w, 0x1.999999999998fp-3);
  a = __builtin_fma(a, w, -0x1.5555555555555p-2);
  double res = hi + __builtin_fma(u * w, a, u);               // atan(mn / mx)
  const double pio2 = 0x1.921fb54442d18p+0, pi = 0x1.921fb54442d18p+1;
  res = ay > ax ? pio2 - res : res;
  res = x < 0.0 ? pi - res : res;
  return y < 0.0 ? -res : res;
}

// K_0, K_1, K_2 of one argument: bl_cyl_bessel_k012's algorithm (Temme's series below 2, Steed's continued fraction above,
// the same stopping rules) with the tier's logarithm, exponential and reciprocals in place of the pinned functions and the
// IEEE divisions (a dozen per term there). x finite and positive (1 / Theta_e with Theta_e >= 0.01).
__device__ __forceinline__ void bessel_k012(double x, double *k0, double *k1, double *k2) {
  const double eps = 0x1p-52;
  const double xi = rcp(x);
  double kmu, knu1;
  if (x < 2.0) {
    const double x2 = 0.5 * x;
    const double d = -log(x2);
    double ff = -0.57721566490153286 + d;
    double sum = ff, p = 0.5, q = 0.5, c = 1.0, sum1 = 0.5;
    const double dd = x2 * x2;
    for (int i = 1; i <= 15000; ++i) {
      const double di = (double)i, inv_i = rcp(di);
      ff = (di * ff + p + q) * (inv_i * inv_i);
      c *= dd * inv_i;
      p *= inv_i;
      q *= inv_i;
      const double del = c * ff;
      sum += del;
      sum1 += c * (p - di * ff);
      if (__builtin_fabs(del) < eps * __builtin_fabs(sum)) break;
    }
    kmu = sum;
    knu1 = sum1 * (2.0 * xi);
  } else {
    double b = 2.0 * (1.0 + x);
    double d = rcp(b);
    double delh = d, h = d, q1 = 0.0, q2 = 1.0;
    const double a1 = 0.25;
    double c = a1, q = a1, a = -a1;
    double s = 1.0 + q * delh;
    for (int i = 2; i <= 15000; ++i) {
      a -= (double)(2 * (i - 1));
      c = -a * c * rcp((double)i);
      const double qnew = (q1 - b * q2) * rcp(a);
      q1 = q2;
      q2 = qnew;
      q += c * qnew;
      b += 2.0;
      d = rcp(b + a * d);
      delh = (b * d - 1.0) * delh;
      h += delh;
      const double dels = q * delh;
      s += dels;
      if (__builtin_fabs(dels) < eps * __builtin_fabs(s)) break;
    }
    h = a1 * h;
    kmu = bl_sqrt_g(3.141592653589793 * 0.5 * xi) * exp(-x) * rcp(s);
    knu1 = kmu * (x + 0.5 - h) * xi;
  }
  *k0 = kmu;
  *k1 = knu1;
  *k2 = 2.0 * xi * knu1 + kmu;
}

}  // namespace fastmath

#endif

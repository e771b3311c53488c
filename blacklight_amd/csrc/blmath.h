/* blmath.h - the build's own bit-reproducible double-precision math library.
 *
 * Why it exists: the reference's ray stepper is an adaptive ODE controller whose step sequence,
 * sample counts and termination flags depend on the last bit of std::hypot (reference
 * src/geodesic_integrator/geodesic_geometry.cpp:23,65,133,189) and std::pow
 * (src/geodesic_integrator/geodesics.cpp:202,215), and the reference gets those from whatever
 * libm the host has (glibc picks different code paths on FMA / non-FMA CPUs). "Bit-exact sample
 * counts" is therefore only well defined against a pinned math library. This header IS that pin:
 * the same source is compiled (a) into the gfx950 HIP kernels, (b) into the host-side C++ that
 * sets up the camera frame, (c) into oracle/libbl_oracle.so (the CPU restatement) and (d) into
 * oracle/_ref/libblmath_preload.so, which is LD_PRELOADed into the UNMODIFIED reference binary to
 * produce the tier-B golden vectors in tests/golden/.
 *
 * Every function uses only IEEE-754 binary64 +, -, *, /, sqrt, fma (all correctly rounded on
 * x86-64 and on gfx950) plus integer bit manipulation, so results are identical on every platform
 * provided the translation unit is compiled with -ffp-contract=off (no implicit contraction; the
 * explicit __builtin_fma calls below are the only fused operations).
 *
 * Accuracy (checked against mpmath in tests/test_blmath.py): hypot, cbrt, log, pow <= 0.51 ulp
 * in practice (double-double cores); exp, expm1, sin, cos, atan, atan2, acos <= ~1 ulp.
 * Polynomial coefficients come from tools/gen_blmath_coeffs.py (Chebyshev-node fits at 80 digits).
 *
 * C and C++ compatible; in HIP every function is __host__ __device__.
 */
#ifndef BLACKLIGHT_AMD_BLMATH_H_
#define BLACKLIGHT_AMD_BLMATH_H_

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define BLM_FN static inline __host__ __device__
#else
#define BLM_FN static inline
#endif

/* BLM_K(c): a 64-bit literal operand. The device compiler would put it in a vector register pair with two v_mov_b32 at
 * every use - as dear as the fused multiply-add that consumes it, which triples the cost of a Horner step; handed through
 * an empty scalar-register instruction it is two s_mov_b32 on the scalar unit (which runs beside the vector unit) and a
 * scalar operand of the vector instruction. Same value, same arithmetic. Host: the literal itself. */
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BLM_NO_SCALAR_LITERALS)
static inline __device__ double blm_scalar_literal(double v) {
  asm volatile("" : "+s"(v));
  return v;
}
#define BLM_K(c) blm_scalar_literal(c)
#else
#define BLM_K(c) (c)
#endif

/* blm_div(a, b), blm_sqrt_n(x): the IEEE quotient / square root for operands in the middle of the exponent range. The device
 * compiler expands `/` into v_div_scale x2, v_rcp, two Newton steps, a multiplication, a residual step, v_div_fmas, v_div_fixup,
 * where the scaling instructions only act when an exponent is near an end of the range (|b| or |a / b| beyond 2^+-1000,
 * 0 < |a| < 2^-900); without them the same sequence is five instructions shorter and bit-identical wherever no scaling would
 * have happened (tests/test_gpu_math.py). Likewise sqrt without its 2^256 pre-scaling of arguments below 2^-767. Used where the
 * operands are known to be ordinary: ratios of coordinates, polynomial arguments of the inverse trigonometric functions. Host:
 * the plain operations. */
#if defined(__HIP_DEVICE_COMPILE__)
static inline __device__ double blm_div(double a, double b) {
  double y = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-b, y, 1.0);
  y = __builtin_fma(y, e, y);
  double q = a * y;
  const double r = __builtin_fma(-b, q, a);
  q = __builtin_fma(r, y, q);
  return __builtin_amdgcn_div_fixup(q, b, a);
}
static inline __device__ double blm_sqrt_n(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = y * 0.5;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  return __builtin_amdgcn_class(x, 0x260) ? x : g;   /* +-0, +inf */
}
#else
#define blm_div(a, b) ((a) / (b))
#define blm_sqrt_n(x) __builtin_sqrt(x)
#endif

#define BLM_INF (__builtin_inf())
#define BLM_NAN (__builtin_nan(""))

typedef struct { double hi, lo; } blm_dd;

/* ---------------------------------------------------------------- bit helpers */
BLM_FN uint64_t blm_bits(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
BLM_FN double blm_from_bits(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }
BLM_FN double blm_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
BLM_FN double blm_abs(double x) { return blm_from_bits(blm_bits(x) & 0x7fffffffffffffffull); }
BLM_FN double blm_sqrt(double x) { return __builtin_sqrt(x); }
BLM_FN int blm_isnan(double x) { return x != x; }
BLM_FN int blm_isinf(double x) { return (blm_bits(x) & 0x7fffffffffffffffull) == 0x7ff0000000000000ull; }
BLM_FN double blm_copysign(double mag, double sgn) {
  return blm_from_bits((blm_bits(mag) & 0x7fffffffffffffffull) | (blm_bits(sgn) & 0x8000000000000000ull));
}
/* round to nearest integer (ties to even) for |v| < 2^51, without a libm call */
BLM_FN double blm_rint(double v) { return (v + BLM_K(0x1.8p52)) - BLM_K(0x1.8p52); }
/* 2^k for -1022 <= k <= 1023 */
BLM_FN double blm_pow2i(int k) { return blm_from_bits((uint64_t)(k + 1023) << 52); }
/* x * 2^k, any int k (results in the subnormal range are rounded once more; never on our path) */
BLM_FN double blm_scalbn(double x, int k) {
  if (k > 1023) { x *= 0x1p1023; k -= 1023; if (k > 1023) { x *= 0x1p1023; k -= 1023; if (k > 1023) k = 1023; } }
  else if (k < -1022) { x *= 0x1p-969; k += 969; if (k < -1022) { x *= 0x1p-969; k += 969; if (k < -1022) k = -1022; } }
  return x * blm_pow2i(k);
}

/* modf: integral part through *ip (truncation towards zero), fractional part returned, both with the sign of x; exact */
BLM_FN double bl_modf(double x, double *ip) {
  if (blm_isinf(x)) { *ip = x; return blm_copysign(0.0, x); }
  double t = __builtin_trunc(x);
  *ip = t;
  return blm_copysign(x - t, x);
}

/* ---------------------------------------------------------------- double-double toolkit */
BLM_FN blm_dd blm_two_sum(double a, double b) {
  blm_dd r; r.hi = a + b; double bb = r.hi - a; r.lo = (a - (r.hi - bb)) + (b - bb); return r;
}
BLM_FN blm_dd blm_fast_two_sum(double a, double b) { /* |a| >= |b| */
  blm_dd r; r.hi = a + b; r.lo = b - (r.hi - a); return r;
}
BLM_FN blm_dd blm_two_prod(double a, double b) {
  blm_dd r; r.hi = a * b; r.lo = blm_fma(a, b, -r.hi); return r;
}
BLM_FN blm_dd blm_dd_add(blm_dd a, blm_dd b) {
  blm_dd s = blm_two_sum(a.hi, b.hi);
  blm_dd t = blm_two_sum(a.lo, b.lo);
  s.lo += t.hi; s = blm_fast_two_sum(s.hi, s.lo);
  s.lo += t.lo; return blm_fast_two_sum(s.hi, s.lo);
}
BLM_FN blm_dd blm_dd_add_d(blm_dd a, double b) {
  blm_dd s = blm_two_sum(a.hi, b); s.lo += a.lo; return blm_fast_two_sum(s.hi, s.lo);
}
BLM_FN blm_dd blm_dd_mul(blm_dd a, blm_dd b) {
  blm_dd p = blm_two_prod(a.hi, b.hi);
  p.lo += a.hi * b.lo + a.lo * b.hi;
  return blm_fast_two_sum(p.hi, p.lo);
}
BLM_FN blm_dd blm_dd_mul_d(blm_dd a, double b) {
  blm_dd p = blm_two_prod(a.hi, b); p.lo = blm_fma(a.lo, b, p.lo); return blm_fast_two_sum(p.hi, p.lo);
}
BLM_FN blm_dd blm_dd_scale(blm_dd a, double pow2) { a.hi *= pow2; a.lo *= pow2; return a; }

/* ---------------------------------------------------------------- hypot */
/* sqrt(x^2+y^2): exact double-double sum of squares, sqrt, one fma-residual Newton step. */
BLM_FN double bl_hypot(double x, double y) {
  double ax = blm_abs(x), ay = blm_abs(y);
  if (blm_isinf(ax) || blm_isinf(ay)) return BLM_INF;
  if (blm_isnan(ax) || blm_isnan(ay)) return ax + ay;
  if (ax < ay) { double t = ax; ax = ay; ay = t; }
  if (ay == 0.0) return ax;
  double unscale = 1.0;
  if (ax > 0x1p510) { ax *= 0x1p-600; ay *= 0x1p-600; unscale = 0x1p600; }
  else if (ay < 0x1p-450) { ax *= 0x1p600; ay *= 0x1p600; unscale = 0x1p-600; }
  if (ay < ax * 0x1p-54) return ax * unscale;
  double p = ax * ax, pe = blm_fma(ax, ax, -p);
  double q = ay * ay, qe = blm_fma(ay, ay, -q);
  double hi = p + q;
  double lo = (q - (hi - p)) + (pe + qe);
  double h = blm_sqrt(hi);
  double r = blm_fma(-h, h, hi) + lo;
  h = h + r / (h + h);
  return h * unscale;
}
/* three-argument form (libstdc++ std::hypot(x,y,z) as used by reference camera.cpp:620) */
BLM_FN double bl_hypot3(double x, double y, double z) {
  double ax = blm_abs(x), ay = blm_abs(y), az = blm_abs(z);
  double m = ax > ay ? ax : ay; m = m > az ? m : az;
  if (m == 0.0 || blm_isinf(m)) return m;
  if (blm_isnan(ax) || blm_isnan(ay) || blm_isnan(az)) return ax + ay + az;
  ax /= m; ay /= m; az /= m;
  return m * blm_sqrt(ax * ax + ay * ay + az * az);
}

/* ---------------------------------------------------------------- exp */
#define BLM_LN2_HI BLM_K(0x1.62e42feep-1)          /* 32 significant bits: k*LN2_HI exact for |k| < 2^21 */
#define BLM_LN2_LO BLM_K(0x1.a39ef35793c76p-33)
#define BLM_LN2_TAIL BLM_K(0x1.cc01f97b57a08p-87)
#define BLM_INV_LN2 BLM_K(0x1.71547652b82fep+0)

BLM_FN double bl_exp(double x) {
  if (blm_isnan(x)) return x;
  if (x > BLM_K(0x1.62e42fefa39efp+9)) return BLM_INF;
  if (x < -BLM_K(0x1.74910d52d3051p+9)) return 0.0;
  double kd = blm_rint(x * BLM_INV_LN2);
  int k = (int)kd;
  double r = blm_fma(-kd, BLM_LN2_HI, x);
  r = blm_fma(-kd, BLM_LN2_LO, r);
  double p = BLM_K(0x1.61bfaa228dde5p-33);
  p = blm_fma(p, r, BLM_K(0x1.1f7f2776cfaf2p-29));
  p = blm_fma(p, r, BLM_K(0x1.ae642c82e33d5p-26));
  p = blm_fma(p, r, BLM_K(0x1.27e4d41966f2fp-22));
  p = blm_fma(p, r, BLM_K(0x1.71de3a5aa7bb7p-19));
  p = blm_fma(p, r, BLM_K(0x1.a01a01a9e991bp-16));
  p = blm_fma(p, r, BLM_K(0x1.a01a01a0196acp-13));
  p = blm_fma(p, r, BLM_K(0x1.6c16c16c15a68p-10));
  p = blm_fma(p, r, BLM_K(0x1.1111111111111p-7));
  p = blm_fma(p, r, BLM_K(0x1.5555555555557p-5));
  p = blm_fma(p, r, BLM_K(0x1.5555555555555p-3));
  p = blm_fma(p, r, BLM_K(0x1.0000000000000p-1));
  blm_dd s1 = blm_fast_two_sum(1.0, r);            /* 1 + r exactly */
  double e = s1.hi + blm_fma(r * r, p, s1.lo);
  return blm_scalbn(e, k);
}

/* ---------------------------------------------------------------- expm1 */
BLM_FN double bl_expm1(double x) {
  if (blm_isnan(x)) return x;
  if (x > BLM_K(0x1.62e42fefa39efp+9)) return BLM_INF;
  if (x < -38.0) return -1.0;
  if (blm_abs(x) < 0x1p-54) return x;
  double kd = blm_rint(x * BLM_INV_LN2);
  int k = (int)kd;
  double rh = blm_fma(-kd, BLM_LN2_HI, x);
  double r = blm_fma(-kd, BLM_LN2_LO, rh);
  double c = blm_fma(-kd, BLM_LN2_LO, rh - r);   /* r + c == rh - kd*LN2_LO to ~2^-106 */
  double q = BLM_K(0x1.94328fcb8199cp-37);
  q = blm_fma(q, r, BLM_K(0x1.61bfaa228dde5p-33));
  q = blm_fma(q, r, BLM_K(0x1.1eed7a01fc8b7p-29));
  q = blm_fma(q, r, BLM_K(0x1.ae642c82e33d5p-26));
  q = blm_fma(q, r, BLM_K(0x1.27e4fb7a2782ap-22));
  q = blm_fma(q, r, BLM_K(0x1.71de3a5aa7bb7p-19));
  q = blm_fma(q, r, BLM_K(0x1.a01a01a019b63p-16));
  q = blm_fma(q, r, BLM_K(0x1.a01a01a0196acp-13));
  q = blm_fma(q, r, BLM_K(0x1.6c16c16c16c17p-10));
  q = blm_fma(q, r, BLM_K(0x1.1111111111111p-7));
  q = blm_fma(q, r, BLM_K(0x1.5555555555555p-5));
  q = blm_fma(q, r, BLM_K(0x1.5555555555555p-3));
  double r2 = r * r;
  double tail = blm_fma(r2 * r, q, 0.5 * r2);
  tail += blm_fma(c, r, c);
  double e = r + tail;
  if (k >= 54) return blm_scalbn(1.0 + e, k) - 1.0;
  /* (2^k - 1) + 2^k e: 2^k - 1 and 2^k e are exact, one rounding. For k = 0, -1, +1 this is
     e, 0.5 e - 0.5 and 1 + 2 e bit for bit, so no special cases (and no lane divergence). */
  double t = blm_pow2i(k);
  return (t - 1.0) + t * e;
}

/* ---------------------------------------------------------------- log (double-double core) */
BLM_FN blm_dd blm_log_dd(double x) { /* x finite, > 0 */
  int e = 0;
  uint64_t u = blm_bits(x);
  if (u < 0x0010000000000000ull) { x *= 0x1p54; e = -54; u = blm_bits(x); }
  e += (int)(u >> 52) - 1023;
  double m = blm_from_bits((u & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
  if (m > BLM_K(0x1.6a09e667f3bcdp+0)) { m *= 0.5; e += 1; }
  double f = m - 1.0;
  blm_dd den = blm_two_sum(m, 1.0);
  blm_dd s;
  s.hi = f / den.hi;
  s.lo = (blm_fma(-s.hi, den.hi, f) - s.hi * den.lo) / den.hi;
  blm_dd z;
  z.hi = s.hi * s.hi;
  z.lo = blm_fma(s.hi, s.hi, -z.hi) + 2.0 * s.hi * s.lo;
  double zz = z.hi;
  double l = BLM_K(0x1.615a2da208f78p-5);
  l = blm_fma(l, zz, BLM_K(0x1.441667b6cb80dp-5));
  l = blm_fma(l, zz, BLM_K(0x1.64515e682719ap-5));
  l = blm_fma(l, zz, BLM_K(0x1.861778358d1c4p-5));
  l = blm_fma(l, zz, BLM_K(0x1.af286f7c52d6ep-5));
  l = blm_fma(l, zz, BLM_K(0x1.e1e1e1d845c7ap-5));
  l = blm_fma(l, zz, BLM_K(0x1.1111111118d87p-4));
  l = blm_fma(l, zz, BLM_K(0x1.3b13b13b13a9ep-4));
  l = blm_fma(l, zz, BLM_K(0x1.745d1745d1746p-4));
  l = blm_fma(l, zz, BLM_K(0x1.c71c71c71c71cp-4));
  l = blm_fma(l, zz, BLM_K(0x1.2492492492492p-3));
  double tail = zz * zz * zz * l;
  blm_dd third = {BLM_K(0x1.5555555555555p-2), BLM_K(0x1.5555555555555p-56)};
  blm_dd fifth = {BLM_K(0x1.999999999999ap-3), -BLM_K(0x1.999999999999ap-57)};
  blm_dd t1 = blm_dd_mul(z, third);
  blm_dd t2 = blm_dd_mul(blm_dd_mul(z, z), fifth);
  blm_dd sum = blm_dd_add_d(t2, tail);
  sum = blm_dd_add(t1, sum);
  sum = blm_dd_add_d(sum, 1.0);
  blm_dd lm = blm_dd_mul(s, sum);
  lm = blm_dd_scale(lm, 2.0);
  double ed = (double)e;
  blm_dd ln2lo = {BLM_LN2_LO, BLM_LN2_TAIL};
  blm_dd el = blm_dd_mul_d(ln2lo, ed);
  blm_dd res = blm_dd_add(el, lm);
  return blm_dd_add_d(res, ed * BLM_LN2_HI);
}

BLM_FN double bl_log(double x) {
  if (blm_isnan(x)) return x;
  if (x < 0.0) return BLM_NAN;
  if (x == 0.0) return -BLM_INF;
  if (blm_isinf(x)) return x;
  blm_dd l = blm_log_dd(x);
  return l.hi + l.lo;
}

/* ---------------------------------------------------------------- exp of a double-double */
/* returns exp(a.hi + a.lo) rounded to double; |a| < 745 assumed */
BLM_FN double blm_exp_dd(blm_dd a) {
  double kd = blm_rint(a.hi * BLM_INV_LN2);
  int k = (int)kd;
  blm_dd r = blm_two_sum(a.hi, -(kd * BLM_LN2_HI));
  blm_dd p2 = blm_two_prod(-kd, BLM_LN2_LO);
  r = blm_dd_add(r, p2);
  r = blm_dd_add_d(r, a.lo);
  r = blm_dd_add_d(r, -kd * BLM_LN2_TAIL);
  r = blm_dd_scale(r, 0.125);
  double rr = r.hi;
  double ep = BLM_K(0x1.1eef4361593cbp-29);
  ep = blm_fma(ep, rr, BLM_K(0x1.ae67522acca97p-26));
  ep = blm_fma(ep, rr, BLM_K(0x1.27e4fb764f679p-22));
  ep = blm_fma(ep, rr, BLM_K(0x1.71de3a5345fa0p-19));
  ep = blm_fma(ep, rr, BLM_K(0x1.a01a01a01a025p-16));
  ep = blm_fma(ep, rr, BLM_K(0x1.a01a01a01a02cp-13));
  ep = blm_fma(ep, rr, BLM_K(0x1.6c16c16c16c17p-10));
  ep = blm_fma(ep, rr, BLM_K(0x1.1111111111111p-7));
  ep = blm_fma(ep, rr, BLM_K(0x1.5555555555555p-5));
  double r2d = rr * rr;
  double tail = r2d * r2d * ep;
  blm_dd r2 = blm_dd_mul(r, r);
  blm_dd r3 = blm_dd_mul(r2, r);
  blm_dd sixth = {BLM_K(0x1.5555555555555p-3), BLM_K(0x1.5555555555555p-57)};
  blm_dd e = blm_dd_add_d(blm_dd_mul(r3, sixth), tail);
  e = blm_dd_add(blm_dd_scale(r2, 0.5), e);
  e = blm_dd_add(r, e);
  for (int i = 0; i < 3; i++) e = blm_dd_add(blm_dd_scale(e, 2.0), blm_dd_mul(e, e));
  blm_dd one = blm_dd_add_d(e, 1.0);
  return blm_scalbn(one.hi + one.lo, k);
}

/* ---------------------------------------------------------------- pow */
BLM_FN double bl_pow(double x, double y) {
  if (y == 0.0) return 1.0;
  if (x == 1.0) return 1.0;
  if (blm_isnan(x) || blm_isnan(y)) return x + y;
  double ay = blm_abs(y);
  /* classify y: 0 = non-integer, 1 = odd integer, 2 = even integer */
  int yint = 0;
  if (ay >= 0x1p53) yint = 2;
  else if (ay >= 1.0) {
    double fl = blm_rint(ay);
    if (fl == ay) { double h = fl * 0.5; yint = (blm_rint(h) == h) ? 2 : 1; }
  }
  double ax = blm_abs(x);
  int neg = (blm_bits(x) >> 63) != 0;
  if (blm_isinf(y)) {
    if (ax == 1.0) return 1.0;
    return ((ax > 1.0) == (y > 0.0)) ? BLM_INF : 0.0;
  }
  if (ax == 0.0 || blm_isinf(ax)) {
    double z = ((ax == 0.0) == (y > 0.0)) ? 0.0 : BLM_INF;
    return (neg && yint == 1) ? -z : z;
  }
  if (neg && yint == 0) return BLM_NAN;
  double sign = (neg && yint == 1) ? -1.0 : 1.0;
  blm_dd l = blm_log_dd(ax);
  blm_dd a = blm_two_prod(l.hi, y);
  a.lo = blm_fma(l.lo, y, a.lo);
  a = blm_fast_two_sum(a.hi, a.lo);
  if (a.hi > 709.79) return sign * BLM_INF;
  if (a.hi < -745.2) return sign * 0.0;
  return sign * blm_exp_dd(a);
}

/* bl_pow(x, -0.2) - the Dormand-Prince step controller's only power (geodesics.cpp:202, :215), once per step attempt - by a
   routine of its own on the device: a power correctly rounded is one number, so any routine that rounds correctly returns
   bl_pow's bits. The exponent is the DOUBLE -0.2 = -(1 + 2^-54) / 5, not -1/5: the result is Y^(1 + 2^-54) with Y = x^(-1/5).
   x = m 2^(5q) with m in [1, 32); y ~ m^(-1/5) from a single-precision seed and two Newton steps on 1 - m y^5 in double; one more
   step with m y^5 in double-double arithmetic leaves y + delta within 2^-98 of the root; the factor Y^(2^-54) = 1 + 2^-54 ln Y
   enters delta with ln Y from the hardware's single-precision logarithm (1e-7 absolute, i.e. 2^-77 of the result). The sum
   is rounded once, and the rounding is accepted only when the exact remainder t of that sum keeps 2^-12 half-units (2^-66 of the
   result) away from a rounding boundary. That band is set by bl_pow, not by this routine: the contract is bl_pow's bits, and
   bl_pow is NOT correctly rounded everywhere - the tail of its logarithm's series is summed in plain double, which leaves it
   ~2^-71 of the result in the worst cases seen (x = 5.853689088059315: the true power lies 2.3e-6 units above a midpoint and
   bl_pow rounds down) - so this routine only answers where a 2^-66 error cannot change the rounding, and asks bl_pow itself
   otherwise (2^-12 of the calls, and arguments outside 2^+-900). ~80 instructions where bl_pow takes ~520 (7 % of a step).
   Host builds: bl_pow. tests/test_gpu_math.py compares the two on 10^8 arguments. */
#if defined(__HIP_DEVICE_COMPILE__)
static inline __device__ double bl_pow_neg_fifth(double x) {
  const uint64_t u = blm_bits(x);
  const int e_biased = (int)(u >> 52);                       /* sign bit set (x < 0) lands far outside the window below */
  double result = 0.0;
  bool fast = (unsigned)(e_biased - 123) < 1800u;            /* x finite, positive, exponent in [-900, 899] */
  if (fast) {
    const int e = e_biased - 1023;
    const int q = (int)(((unsigned)(e + 5000) * 52429u) >> 18) - 1000;   /* floor(e / 5) */
    const int r = e - 5 * q;                                 /* 0 ... 4 */
    const double m = blm_from_bits((u & 0x000fffffffffffffull) | ((uint64_t)(1023 + r) << 52));   /* [1, 32) */
    double y = (double)__builtin_amdgcn_exp2f(-0.2f * __builtin_amdgcn_logf((float)m));        /* ~2^-21 */
    for (int step = 0; step < 2; step++) {                   /* y <- y + y (1 - m y^5) / 5 */
      const double y2 = y * y, y4 = y2 * y2;
      const double rho = __builtin_fma(-(y4 * y), m, 1.0);
      y = __builtin_fma(y * 0.2, rho, y);
    }
    blm_dd p = blm_two_prod(y, y);                           /* y^2 exactly */
    p = blm_dd_mul(p, p);                                    /* y^4 */
    p = blm_dd_mul_d(p, y);                                  /* y^5 */
    p = blm_dd_mul_d(p, m);                                  /* m y^5 = 1 + O(2^-51) */
    const double rho = (1.0 - p.hi) - p.lo;
    /* ln Y = ln y - q ln 2 */
    const double ln_y = __builtin_fma(-(double)q, 0x1.62e42fefa39efp-1, (double)__builtin_amdgcn_logf((float)y) * 0x1.62e42fefa39efp-1);
    const double delta = __builtin_fma(y * 0x1p-54, ln_y, (y * 0.2) * rho);
    const double s = y + delta;
    const double t = (y - s) + delta;                        /* y + delta = s + t exactly (|delta| << |y|) */
    /* Half a unit in the last place of s: 2^-54 for s in [1/2, 1] (m >= 1 keeps s <= 1) - and 2^-55 below 1/2, where the factor
       Y^(2^-54) can push s for x just under a power of 32, and for s = 1/2 itself when the remainder points downwards (the
       neighbour below is half as far away). Without the case the guard never fired in that binade (|t| <= 2^-55 there). */
    const double half_ulp = (s < 0.5 || (s == 0.5 && t < 0.0)) ? 0x1p-55 : 0x1p-54;
    fast = !(blm_abs(t) >= half_ulp * (1.0 - 0x1p-12));
    result = s * blm_pow2i(-q);                              /* exact: |q| <= 180 */
  }
  return fast ? result : bl_pow(x, -0.2);
}
#else
#define bl_pow_neg_fifth(x) bl_pow((x), -0.2)
#endif

/* Several powers of one base: bl_pow(x, y) spends a third of its work on log(x). bl_pow_base() takes that
   logarithm once; bl_pow_of(base, y) is bl_pow(x, y) bit for bit - for a finite positive x other than 1 it is the
   tail of bl_pow on the same double-double logarithm, for every other x it calls bl_pow. */
typedef struct { double x; blm_dd l; int regular; } blm_powbase;
BLM_FN blm_powbase bl_pow_base(double x) {
  blm_powbase b;
  b.x = x;
  b.regular = x > 0.0 && !blm_isinf(x) && x != 1.0;   /* false for NaN */
  b.l.hi = b.l.lo = 0.0;
  if (b.regular) b.l = blm_log_dd(x);
  return b;
}
BLM_FN double bl_pow_of(blm_powbase b, double y) {
  if (!b.regular) return bl_pow(b.x, y);
  if (y == 0.0) return 1.0;
  if (blm_isnan(y)) return b.x + y;
  if (blm_isinf(y)) return ((b.x > 1.0) == (y > 0.0)) ? BLM_INF : 0.0;
  blm_dd a = blm_two_prod(b.l.hi, y);
  a.lo = blm_fma(b.l.lo, y, a.lo);
  a = blm_fast_two_sum(a.hi, a.lo);
  if (a.hi > 709.79) return BLM_INF;
  if (a.hi < -745.2) return 0.0;
  return blm_exp_dd(a);
}

/* ---------------------------------------------------------------- cbrt */
BLM_FN double bl_cbrt(double x) {
  if (x == 0.0 || blm_isnan(x) || blm_isinf(x)) return x;
  double ax = blm_abs(x);
  int adj = 0;
  uint64_t u = blm_bits(ax);
  if (u < 0x0010000000000000ull) { ax *= 0x1p54; adj = -18; u = blm_bits(ax); }
  int e = (int)(u >> 52) - 1023;
  double m = blm_from_bits((u & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
  int q = (e >= 0) ? e / 3 : -((2 - e) / 3);
  int rem = e - 3 * q;                       /* 0, 1, 2 */
  double xs = rem == 0 ? m : (rem == 1 ? m * 2.0 : m * 4.0);   /* in [1, 8) */
  double t = BLM_K(0x1.4c7608a04eba1p-8);
  t = blm_fma(t, m, -BLM_K(0x1.8bd2dce403128p-5));
  t = blm_fma(t, m, BLM_K(0x1.92bfc00e33108p-3));
  t = blm_fma(t, m, -BLM_K(0x1.d758498b983bcp-2));
  t = blm_fma(t, m, BLM_K(0x1.a9da3cc66f245p-1));
  t = blm_fma(t, m, BLM_K(0x1.e68ceb1fc3429p-2));
  t *= rem == 0 ? 1.0 : (rem == 1 ? BLM_K(0x1.428a2f98d728bp+0) : BLM_K(0x1.965fea53d6e3dp+0));
  double t3 = t * t * t;
  t = t * (t3 + 2.0 * xs) / (2.0 * t3 + xs);            /* Halley */
  double t2 = t * t, t2e = blm_fma(t, t, -t2);
  double res = blm_fma(t2, t, -xs) + t2e * t;           /* t^3 - xs, exact to ~2^-100 */
  t = t - res / (3.0 * t2);
  return blm_copysign(blm_scalbn(t, q + adj), x);
}

/* ---------------------------------------------------------------- sin / cos */
#define BLM_PIO2_HI BLM_K(0x1.921fb54442d18p+0)
#define BLM_PIO2_LO BLM_K(0x1.1a62633145c07p-54)
#define BLM_PI_HI BLM_K(0x1.921fb54442d18p+1)
#define BLM_PI_LO BLM_K(0x1.1a62633145c07p-53)

/* argument reduction: x = n*(pi/2) + (r.hi + r.lo), |r| <= ~pi/4; returns n mod 4.
   pi/2 is carried as three full doubles (~159 bits); products are exact via fma. */
BLM_FN int blm_rem_pio2(double x, blm_dd *r) {
  double fn = blm_rint(x * BLM_K(0x1.45f306dc9c883p-1));
  blm_dd p1 = blm_two_prod(-fn, BLM_K(0x1.921fb54442d18p+0));
  blm_dd p2 = blm_two_prod(-fn, BLM_K(0x1.1a62633145c07p-54));
  double p3 = -fn * -BLM_K(0x1.f1976b7ed8fbcp-110);
  blm_dd acc = blm_two_sum(x, p1.hi);
  blm_dd t = blm_two_sum(p1.lo, p2.hi);
  acc = blm_dd_add(acc, t);
  acc = blm_dd_add_d(acc, p2.lo + p3);
  *r = acc;
  double q = fn - 4.0 * blm_rint(fn * 0.25);   /* in [-2, 2] */
  int n = (int)q;
  return n & 3;
}
BLM_FN double blm_ksin(double r, double lo) {
  double z = r * r;
  double s = -BLM_K(0x1.ab17a79237a19p-41);
  s = blm_fma(s, z, BLM_K(0x1.61217ec01749dp-33));
  s = blm_fma(s, z, -BLM_K(0x1.ae64541266378p-26));
  s = blm_fma(s, z, BLM_K(0x1.71de3a54605eep-19));
  s = blm_fma(s, z, -BLM_K(0x1.a01a01a019936p-13));
  s = blm_fma(s, z, BLM_K(0x1.1111111111110p-7));
  s = blm_fma(s, z, -BLM_K(0x1.5555555555555p-3));
  double corr = lo * (1.0 - 0.5 * z);
  return r + blm_fma(r * z, s, corr);
}
BLM_FN double blm_kcos(double r, double lo) {
  double z = r * r;
  double c = BLM_K(0x1.ab783376962cfp-45);
  c = blm_fma(c, z, -BLM_K(0x1.9394b9c9c20a4p-37));
  c = blm_fma(c, z, BLM_K(0x1.1eed8deb6d561p-29));
  c = blm_fma(c, z, -BLM_K(0x1.27e4fb7712bdfp-22));
  c = blm_fma(c, z, BLM_K(0x1.a01a01a019d0ap-16));
  c = blm_fma(c, z, -BLM_K(0x1.6c16c16c16c16p-10));
  c = blm_fma(c, z, BLM_K(0x1.5555555555555p-5));
  double hz = 0.5 * z;
  double w = 1.0 - hz;
  double tail = ((1.0 - w) - hz) + blm_fma(z * z, c, -r * lo);
  return w + tail;
}
BLM_FN double bl_sin(double x) {
  if (blm_isnan(x) || blm_isinf(x)) return BLM_NAN;
  blm_dd r; int n = blm_rem_pio2(x, &r);
  switch (n) {
    case 0: return blm_ksin(r.hi, r.lo);
    case 1: return blm_kcos(r.hi, r.lo);
    case 2: return -blm_ksin(r.hi, r.lo);
    default: return -blm_kcos(r.hi, r.lo);
  }
}
BLM_FN double bl_cos(double x) {
  if (blm_isnan(x) || blm_isinf(x)) return BLM_NAN;
  blm_dd r; int n = blm_rem_pio2(x, &r);
  switch (n) {
    case 0: return blm_kcos(r.hi, r.lo);
    case 1: return -blm_ksin(r.hi, r.lo);
    case 2: return -blm_kcos(r.hi, r.lo);
    default: return blm_ksin(r.hi, r.lo);
  }
}
BLM_FN void bl_sincos(double x, double *s, double *c) {
  if (blm_isnan(x) || blm_isinf(x)) { *s = BLM_NAN; *c = BLM_NAN; return; }
  blm_dd r; int n = blm_rem_pio2(x, &r);
  double sn = blm_ksin(r.hi, r.lo), cs = blm_kcos(r.hi, r.lo);
  switch (n) {
    case 0: *s = sn; *c = cs; break;
    case 1: *s = cs; *c = -sn; break;
    case 2: *s = -sn; *c = -cs; break;
    default: *s = -cs; *c = sn; break;
  }
}

/* ---------------------------------------------------------------- atan / atan2 */
BLM_FN double bl_atan(double x) {
  if (blm_isnan(x)) return x;
  double ax = blm_abs(x);
  if (ax >= 0x1p66) return blm_copysign(BLM_PIO2_HI + BLM_PIO2_LO, x);
  if (ax < 0x1p-27) return x;
  /* range selection without divergent branches: t = num / den, atan(ax) = hi + lo + atan(t) */
  double num = ax, den = 1.0, hi = 0.0, lo = 0.0;          /* ax < 7/16: t = ax / 1 = ax exactly */
  if (ax >= 0.4375) { num = 2.0 * ax - 1.0; den = 2.0 + ax; hi = BLM_K(0x1.dac670561bb4fp-2); lo = BLM_K(0x1.a2b7f222f65e2p-56); }
  if (ax >= 0.6875) { num = ax - 1.0; den = ax + 1.0; hi = BLM_K(0x1.921fb54442d18p-1); lo = BLM_K(0x1.1a62633145c07p-55); }
  if (ax >= 1.1875) { num = ax - 1.5; den = 1.0 + 1.5 * ax; hi = BLM_K(0x1.f730bd281f69bp-1); lo = BLM_K(0x1.007887af0cbbdp-56); }
  if (ax >= 2.4375) { num = -1.0; den = ax; hi = BLM_PIO2_HI; lo = BLM_PIO2_LO; }
  double t = blm_div(num, den);             /* |num| <= 2 |x| + 1, 1 <= den: ordinary operands */
  double z = t * t;
  double a = -BLM_K(0x1.9a0e3d8214a3cp-7);
  a = blm_fma(a, z, BLM_K(0x1.dde84abd3489ap-6));
  a = blm_fma(a, z, -BLM_K(0x1.4ac01ab40659fp-5));
  a = blm_fma(a, z, BLM_K(0x1.812cf294b38a9p-5));
  a = blm_fma(a, z, -BLM_K(0x1.ae800c7915a0cp-5));
  a = blm_fma(a, z, BLM_K(0x1.e1d239c838f12p-5));
  a = blm_fma(a, z, -BLM_K(0x1.1110907ae84ccp-4));
  a = blm_fma(a, z, BLM_K(0x1.3b13abac1919fp-4));
  a = blm_fma(a, z, -BLM_K(0x1.745d171e2e854p-4));
  a = blm_fma(a, z, BLM_K(0x1.c71c71c673bd9p-4));
  a = blm_fma(a, z, -BLM_K(0x1.24924924918e2p-3));
  a = blm_fma(a, z, BLM_K(0x1.999999999998fp-3));
  a = blm_fma(a, z, -BLM_K(0x1.5555555555555p-2));
  double tp = t * (z * a);                    /* atan(t) = t + tp */
  double res = hi + ((tp + lo) + t);          /* hi = lo = 0 in the direct range: exactly t + tp */
  return blm_copysign(res, x);
}
BLM_FN double bl_atan2(double y, double x) {
  if (blm_isnan(x) || blm_isnan(y)) return x + y;
  int sy = (blm_bits(y) >> 63) != 0, sx = (blm_bits(x) >> 63) != 0;
  double pi = BLM_PI_HI, pio2 = BLM_PIO2_HI;
  if (y == 0.0) return sx ? (sy ? -pi : pi) : y;
  if (x == 0.0) return sy ? -pio2 : pio2;
  if (blm_isinf(x)) {
    if (blm_isinf(y)) { double v = sx ? 3.0 * BLM_K(0x1.921fb54442d18p-1) : BLM_K(0x1.921fb54442d18p-1); return sy ? -v : v; }
    return sx ? (sy ? -pi : pi) : (sy ? -0.0 : 0.0);
  }
  if (blm_isinf(y)) return sy ? -pio2 : pio2;
  double ax = blm_abs(x), ay = blm_abs(y);
  double z;
  if (ay > ax * 0x1p64) z = BLM_PIO2_HI + 0.5 * BLM_PI_LO;
  else if (sx && ay < ax * 0x1p-64) z = 0.0;
  else {
    /* 2^-64 <= ay / ax <= 2^64 here; the short division needs both operands away from the ends of the exponent range */
    const unsigned ex = (unsigned)(blm_bits(ax) >> 52), ey = (unsigned)(blm_bits(ay) >> 52);
    const int ordinary = (ex - 223u) < 1600u && (ey - 223u) < 1600u;   /* 2^-800 <= |x|, |y| < 2^800 */
    z = bl_atan(ordinary ? blm_div(ay, ax) : ay / ax);
  }
  double res = sx ? (pi - (z - BLM_PI_LO)) : z;
  return sy ? -res : res;
}

/* ---------------------------------------------------------------- acos */
BLM_FN double blm_asin_r(double z) { /* (asin(x)-x)/x^3 with z = x^2 <= 0.25 */
  double r = BLM_K(0x1.e58a4f278e007p-6);
  r = blm_fma(r, z, -BLM_K(0x1.3bd7e353ddbc2p-6));
  r = blm_fma(r, z, BLM_K(0x1.40c91fa8deb7ep-6));
  r = blm_fma(r, z, BLM_K(0x1.8dcdf11997e0fp-9));
  r = blm_fma(r, z, BLM_K(0x1.31777489dfd29p-7));
  r = blm_fma(r, z, BLM_K(0x1.3b462d121c5d2p-7));
  r = blm_fma(r, z, BLM_K(0x1.7b02ef007d23ep-7));
  r = blm_fma(r, z, BLM_K(0x1.c990a42b32b03p-7));
  r = blm_fma(r, z, BLM_K(0x1.1c4efd20ebb99p-6));
  r = blm_fma(r, z, BLM_K(0x1.6e8ba121b9d5fp-6));
  r = blm_fma(r, z, BLM_K(0x1.f1c71c7a5e151p-6));
  r = blm_fma(r, z, BLM_K(0x1.6db6db6dac0eap-5));
  r = blm_fma(r, z, BLM_K(0x1.3333333333389p-4));
  r = blm_fma(r, z, BLM_K(0x1.5555555555555p-3));
  return r;
}
BLM_FN double bl_acos(double x) {
  if (blm_isnan(x)) return x;
  double ax = blm_abs(x);
  if (ax > 1.0) return BLM_NAN;
  if (ax == 1.0) return x > 0.0 ? 0.0 : BLM_PI_HI + BLM_PI_LO;
  if (ax < 0x1p-57) return BLM_PIO2_HI + BLM_PIO2_LO;
  /* Three ranges, evaluated without divergent branches (a wave usually holds lanes of all three):
       |x| < 0.5 : pi/2 - (x + x^3 R(x^2))
       x <= -0.5 : pi - 2 (s + s^3 R(z)),  z = (1 + x)/2, s = sqrt(z)
       x >=  0.5 : 2 (s + s^3 R(z)),       z = (1 - x)/2, with s split as s + c exactly
     1 + x = 1 - |x| exactly for x < 0, so z has a single expression in the outer ranges. */
  int small = ax < 0.5;
  double z = small ? x * x : (1.0 - ax) * 0.5;
  double r = z * blm_asin_r(z);
  double s = blm_sqrt_n(z);                    /* z >= 2^-114 */
  double res_small = BLM_PIO2_HI - (x - (BLM_PIO2_LO - x * r));
  double w = r * s - BLM_PIO2_LO;
  double res_neg = BLM_PI_HI - 2.0 * (s + w);
  double c = blm_div(blm_fma(-s, s, z), s + s);   /* sqrt(z) = s + c; the numerator is zero or within 2^-54 of z */
  double res_pos = 2.0 * (s + (r * s + c));
  return small ? res_small : (x < 0.0 ? res_neg : res_pos);
}

/* ---------------------------------------------------------------- hyperbolic (polarized path) */
BLM_FN double bl_sinh(double x) {
  double ax = blm_abs(x);
  if (blm_isnan(x) || blm_isinf(x)) return x;
  double r;
  if (ax < 0x1p-28) return x;
  if (ax < 22.0) { double t = bl_expm1(ax); r = 0.5 * (t + t / (t + 1.0)); }
  else if (ax < 709.0) r = 0.5 * bl_exp(ax);
  else { double w = bl_exp(0.5 * ax); r = (0.5 * w) * w; }
  return blm_copysign(r, x);
}
BLM_FN double bl_cosh(double x) {
  double ax = blm_abs(x);
  if (blm_isnan(x)) return x;
  if (blm_isinf(x)) return BLM_INF;
  if (ax < 22.0) { double t = bl_exp(ax); return 0.5 * t + 0.5 / t; }
  if (ax < 709.0) return 0.5 * bl_exp(ax);
  double w = bl_exp(0.5 * ax); return (0.5 * w) * w;
}
BLM_FN double bl_tanh(double x) {
  double ax = blm_abs(x);
  if (blm_isnan(x)) return x;
  double r;
  if (ax >= 22.0) r = 1.0;
  else if (ax < 0x1p-55) return x;
  else if (ax >= 1.0) { double t = bl_expm1(2.0 * ax); r = 1.0 - 2.0 / (t + 2.0); }
  else { double t = bl_expm1(-2.0 * ax); r = -t / (t + 2.0); }
  return blm_copysign(r, x);
}

#endif /* BLACKLIGHT_AMD_BLMATH_H_ */

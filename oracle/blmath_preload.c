/* blmath_preload.c - exports the build's own math library (blacklight_amd/csrc/blmath.h) under the
 * libm symbol names so that it can be LD_PRELOADed into the UNMODIFIED reference binary
 * (oracle/_ref/blacklight). That run is "the reference with a pinned math library" and produces the
 * tier-B golden vectors (SURVEY.md section 7 hard part 1, section 8c). TEST INFRASTRUCTURE ONLY.
 *
 * The reference imports acos atan atan2 cbrt cos cosh exp expm1 hypot log modf pow sin sincos sinh
 * tanh tgamma from libm (nm -D); modf is exact and tgamma is only used in one-time non-thermal
 * precalculations, so they are left to glibc.
 */
#include "blmath.h"

#define EXPORT __attribute__((visibility("default")))

EXPORT double hypot(double x, double y) { return bl_hypot(x, y); }
EXPORT double pow(double x, double y) { return bl_pow(x, y); }
EXPORT double exp(double x) { return bl_exp(x); }
EXPORT double expm1(double x) { return bl_expm1(x); }
EXPORT double log(double x) { return bl_log(x); }
EXPORT double cbrt(double x) { return bl_cbrt(x); }
EXPORT double sin(double x) { return bl_sin(x); }
EXPORT double cos(double x) { return bl_cos(x); }
EXPORT void sincos(double x, double *s, double *c) { bl_sincos(x, s, c); }
EXPORT double acos(double x) { return bl_acos(x); }
EXPORT double atan(double x) { return bl_atan(x); }
EXPORT double atan2(double y, double x) { return bl_atan2(y, x); }
EXPORT double sinh(double x) { return bl_sinh(x); }
EXPORT double cosh(double x) { return bl_cosh(x); }
EXPORT double tanh(double x) { return bl_tanh(x); }

/* array entry points used by tests/test_blmath.py through ctypes (prefixed, no clash with libm) */
#define VEC1(name, fn) EXPORT void name(const double *x, double *out, long n) { for (long i = 0; i < n; i++) out[i] = fn(x[i]); }
#define VEC2(name, fn) EXPORT void name(const double *x, const double *y, double *out, long n) { for (long i = 0; i < n; i++) out[i] = fn(x[i], y[i]); }
VEC2(blv_hypot, bl_hypot)
VEC2(blv_pow, bl_pow)
static double pow_through_base(double x, double y) { return bl_pow_of(bl_pow_base(x), y); }
VEC2(blv_pow_of, pow_through_base)
VEC2(blv_atan2, bl_atan2)
VEC1(blv_exp, bl_exp)
VEC1(blv_expm1, bl_expm1)
VEC1(blv_log, bl_log)
VEC1(blv_cbrt, bl_cbrt)
VEC1(blv_sin, bl_sin)
VEC1(blv_cos, bl_cos)
VEC1(blv_acos, bl_acos)
VEC1(blv_atan, bl_atan)
VEC1(blv_sinh, bl_sinh)
VEC1(blv_cosh, bl_cosh)
VEC1(blv_tanh, bl_tanh)

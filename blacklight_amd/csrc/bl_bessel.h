/* bl_bessel.h - modified Bessel functions of the second kind K_0, K_1, K_2 as the reference obtains them
 * from std::cyl_bessel_k (simulation_coefficients.cpp:537-539), i.e. as libstdc++ 11 (GCC 11.4,
 * <tr1/modified_bessel_func.tcc>, __bessel_ik lines 75-258 behind __cyl_bessel_k lines 305-318) evaluates
 * them: Temme's series for x < 2, Steed's continued fraction (CF2) for x >= 2, then the upward recurrence
 * K_{mu+i+1} = (mu + i) (2 / x) K_{mu+i} + K_{mu+i-1}. The header computes I_nu in the same call (a second
 * continued fraction and a downward recurrence); nothing of that enters K_nu, so it is left out. For integer
 * order the fractional part mu of the order is 0 and __gamma_temme (bessel_function.tcc:100-119) returns
 * gampl = gammi = 1 / tgamma(1) = 1, gam1 = -(Euler's constant), gam2 = 1.
 * Same operations in the same order as the header, on the pinned elementary functions of blmath.h.
 * Host and device (BLM_FN). */
#ifndef BLACKLIGHT_AMD_BL_BESSEL_H_
#define BLACKLIGHT_AMD_BL_BESSEL_H_

#include "blmath.h"

/* K_mu and K_{mu+1} for mu = 0 (the part of __bessel_ik every integer order shares); x is finite, non-zero */
BLM_FN void bl_bessel_k_core(double x, double *kmu_out, double *knu1_out) {
  const double eps = 0x1p-52;
  const int max_iter = 15000;
  const double mu = 0.0, mu2 = 0.0;
  const double xi = 1.0 / x;
  const double xi2 = 2.0 * xi;
  double kmu, knu1;
  if (x < 2.0) {
    const double x2 = x / 2.0;
    const double fact = 1.0;               /* |pi mu| < eps */
    double d = -bl_log(x2);
    double e = mu * d;
    const double fact2 = 1.0;              /* |e| < eps */
    const double gam1 = -0.57721566490153286, gam2 = 1.0, gampl = 1.0, gammi = 1.0;
    double ff = fact * (gam1 * bl_cosh(e) + gam2 * fact2 * d);
    double sum = ff;
    e = bl_exp(e);
    double p = e / (2.0 * gampl);
    double q = 1.0 / (2.0 * e * gammi);
    double c = 1.0;
    d = x2 * x2;
    double sum1 = p;
    for (int i = 1; i <= max_iter; ++i) {
      ff = (i * ff + p + q) / (i * i - mu2);
      c *= d / i;
      p /= i - mu;
      q /= i + mu;
      const double del = c * ff;
      sum += del;
      const double del1 = c * (p - i * ff);
      sum1 += del1;
      if (blm_abs(del) < eps * blm_abs(sum)) break;
    }
    kmu = sum;
    knu1 = sum1 * xi2;
  } else {
    double b = 2.0 * (1.0 + x);
    double d = 1.0 / b;
    double delh = d;
    double h = delh;
    double q1 = 0.0;
    double q2 = 1.0;
    const double a1 = 0.25 - mu2;
    double c = a1;
    double q = c;
    double a = -a1;
    double s = 1.0 + q * delh;
    for (int i = 2; i <= max_iter; ++i) {
      a -= 2 * (i - 1);
      c = -a * c / i;
      const double qnew = (q1 - b * q2) / a;
      q1 = q2;
      q2 = qnew;
      q += c * qnew;
      b += 2.0;
      d = 1.0 / (b + a * d);
      delh = (b * d - 1.0) * delh;
      h += delh;
      const double dels = q * delh;
      s += dels;
      if (blm_abs(dels / s) < eps) break;
    }
    h = a1 * h;
    kmu = blm_sqrt(3.141592653589793 / (2.0 * x)) * bl_exp(-x) / s;
    knu1 = kmu * (mu + x + 0.5 - h) * xi;
  }
  *kmu_out = kmu;
  *knu1_out = knu1;
}

BLM_FN double bl_cyl_bessel_k(int nl, double x) {
  if (x != x) return x;
  if (x == 0.0) return blm_from_bits(0x7ff0000000000000ull);
  const double mu = 0.0;
  const double xi2 = 2.0 * (1.0 / x);
  double kmu, knu1;
  bl_bessel_k_core(x, &kmu, &knu1);
  for (int i = 1; i <= nl; ++i) {
    const double knutemp = (mu + i) * xi2 * knu1 + kmu;
    kmu = knu1;
    knu1 = knutemp;
  }
  return kmu;
}

/* K_0, K_1, K_2 of one argument: what three calls of bl_cyl_bessel_k(0 / 1 / 2, x) return, bit for bit - each of
 * them starts from the same (K_0, K_1) pair and the upward recurrence only reads it - for a third of the work. */
BLM_FN void bl_cyl_bessel_k012(double x, double *k0, double *k1, double *k2) {
  if (x != x) { *k0 = *k1 = *k2 = x; return; }
  if (x == 0.0) { *k0 = *k1 = *k2 = blm_from_bits(0x7ff0000000000000ull); return; }
  const double xi2 = 2.0 * (1.0 / x);
  double kmu, knu1;
  bl_bessel_k_core(x, &kmu, &knu1);
  *k0 = kmu;
  *k1 = knu1;
  *k2 = (0.0 + 1) * xi2 * knu1 + kmu;
}

#endif

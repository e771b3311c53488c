// bl_sampling_fast.h - device functions of the tolerant arithmetic tier (fused multiply-adds, bl_fastmath.h): the locate step's
// angles and the coefficient formulas of bl_coefficients.inc in their second inclusion. Shared by bl_shade_fast.hip and
// bl_coefficients_freq.hip. Everything in here is compiled with fp contract(fast); the pragma is switched off at the end.
#pragma once
#include "bl_sampling.h"

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

#ifndef BLV_NOCONTRACT
#pragma clang fp contract(fast)
#endif
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

// locate_sample() (bl_sampling.h) for the locate kernels of the tolerant tier - meshes with refinement, inter-block interpolation,
// slow light, grids in blocks with holes: the angles by the tier's functions (a third of the pinned ones' instructions; the exact
// tier's locate kernel spends two thirds of its 1 000 vector instructions per sample on them), the search on them, and where they do
// not decide it - a face, a centre or an end of the azimuth's range within `band` - the exact tier's angles and the search again.
// Status, cell, anchors and every count are the exact tier's; the fractions differ from its by the angles' 1e-15.
template <bool kRefined, bool kSpinZero, int kWhere = kTableHbm, bool kNearby = true>
__device__ __forceinline__ void locate_sample_tolerant(const BlShadeArgs &P, const GridTables &tab, const BlSpacetime &st, double x1, double x2, double x3,
                                                       double r, LocatedSample *out, unsigned long long *gathers, unsigned int *anchors,
                                                       const RefinedTables *refined, double band, bool defer_nearby = false) {
  if (band > 0.0 && P.plasma.simulation_coord == BL_COORD_SKS && !P.grid.fmks) {
    const double th = fastmath::acos(blm_div(x3, r));
    const double ph_unwrapped = kSpinZero ? fastmath::atan2(x2, x1) : fastmath::atan2(x2, x1) - fastmath::atan2(st.bh_a, r);
    double ph = ph_unwrapped;
    ph += ph < 0.0 ? 2.0 * kPi : 0.0;
    const double ph_once = ph;
    ph -= ph >= 2.0 * kPi ? 2.0 * kPi : 0.0;
    const double e0 = __builtin_fabs(ph_unwrapped), e1 = __builtin_fabs(ph_once - 2.0 * kPi);
    out->ph = ph_unwrapped;
    out->f_i = out->f_j = out->f_k = 0.0;
    out->cell = 0u;
    if (e0 > band && e1 > band && locate_from_coordinates<kRefined, kWhere, kNearby>(P, tab, refined, r, th, ph, out, gathers, anchors, band, defer_nearby)) return;
  }
  locate_sample<kRefined, kSpinZero, kWhere, kNearby>(P, tab, st, x1, x2, x3, r, out, gathers, anchors, refined, defer_nearby);
}

// tolerant arithmetic tier of the coefficient formulas (sin, cos, tanh keep the pinned versions: few calls, and their
// arguments need a real range reduction)
#define BLC_NAME(f) f##_fast
#define BLC_SQRT bl_sqrt_g
#define BLC_SQRT_M bl_sqrt_g
// (BLV_*: variants for A/B runs that put the exact tier's function back, one at a time - tools/build_variant.sh)
#ifdef BLV_CBRT
#define BLC_CBRT bl_cbrt
#else
#define BLC_CBRT fastmath::cbrt
#endif
#ifdef BLV_EXP
#define BLC_EXP bl_exp
#else
#define BLC_EXP fastmath::exp
#endif
#ifdef BLV_EXPM1
#define BLC_EXPM1 bl_expm1
#else
#define BLC_EXPM1 fastmath::expm1
#endif
#ifdef BLV_LOG
#define BLC_LOG bl_log
#else
#define BLC_LOG fastmath::log
#endif
#ifdef BLV_POW
#define BLC_POW bl_pow
#define BLC_POWBASE_T blm_powbase
#define BLC_POW_BASE bl_pow_base
#define BLC_POW_OF bl_pow_of
#else
#define BLC_POW fastmath::pow
#define BLC_POWBASE_T fastmath::PowBase
#define BLC_POW_BASE fastmath::pow_base
#define BLC_POW_OF fastmath::pow_of
#endif
#ifdef BLV_DIV
#define BLC_DIV_G bl_div_g
#else
#define BLC_DIV_G fastmath::div
#endif
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


#pragma clang fp contract(off)

// bl_kernel_util.h - what every kernel file of the hot path shares: the physical constants of the reference, its
// std::max / std::min semantics, and the wave-level scans (DPP, VALU only). Device code, gfx950.
//
// The kernels (fp64 throughout, grid primitives fp32 in HBM; no MFMA: the work is per-ray ODE integration and gathers, not
// a contraction), one file per stage of the pipeline:
//   bl_geodesic.hip           bl_ray_init_kernel, bl_geodesic_kernel: camera pixel -> ray, stepping, sample records
//   bl_shade.hip              exact tier: bl_locate_kernel / bl_locate_plain_kernel, bl_shade_kernel / bl_shade_exact_kernel
//   bl_shade_fast.hip         tolerant tier: bl_shade_fused_kernel / bl_shade_fast_kernel / bl_shade_formula_fast_kernel
//   bl_coefficients_freq.hip  per-frequency coefficient kernels of polarized and many-frequency runs
//   bl_transfer.hip           bl_transfer_kernel / _quad / _freq / _aux, bl_tau_kernel
//   bl_polarized.hip          polarized transport
// with the device functions they share in bl_sampling.h (exact tier) and bl_sampling_fast.h (tolerant tier).
//
// Compile with -ffp-contract=off: bit-exact sample counts depend on it (see blmath.h).
#pragma once
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

// A kernel's argument block where it lies in the kernel-argument segment, behind a pointer the optimiser cannot trace back to the
// kernel's entry. A by-value struct argument is copied into registers in the entry block - every field any path of the kernel reads,
// live from the first instruction on: the general locate and coefficient kernels spilled 116 ... 249 scalar registers to vector lanes
// that way. Fields read through this reference are loaded (s_load: the address space is still known, the loads are scalar) where
// they are used. For kernels whose only parameter is the struct (it starts at offset 0 of the segment).
template <typename Args>
__device__ __forceinline__ const Args &kernel_arguments_in_place() {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const __attribute__((address_space(4))) Args *InSegment;
  InSegment p = (InSegment)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return *(const Args *)p;
#else
  return *static_cast<const Args *>(nullptr);
#endif
}

}  // namespace

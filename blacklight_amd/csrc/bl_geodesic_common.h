// bl_geodesic_common.h - what the two geodesic kernels share: the Dormand-Prince 5(4) tableau (bl_geodesic.hip: a ray per lane;
// bl_geodesic_quad.hip: a ray per quad of lanes), and the retirement of unused record slots (the layout of a parked ray, the
// hand-over between the two, is BL_PARK_DOUBLES in bl_device.h).
#pragma once
#include "bl_kernel_util.h"

namespace {

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

// Mark the record slots [first, last) as dead (only the id word is written)
__device__ __forceinline__ void retire_record_slots(BlSampleHot *records, int stride, long long first, long long last, int lane) {
  for (long long at = first + lane; at < last; at += 64) {
    records[at * stride].ray = BL_DEAD_RAY;
    records[at * stride].n = 0u;
  }
}

}  // namespace

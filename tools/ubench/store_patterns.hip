// Cost of a wave's 16-byte stores by address pattern, gfx950: every wave of the chip stores `iters` x 4 dwordx4 per lane,
// lane l of a wave writing at base + (l * stride + k * 16) bytes, k = 0..3 - the geodesic kernel's record stores are the
// stride = emit * 32 case. Reports ns per wave store instruction per CU (8 waves per CU in flight).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/store_patterns.hip -o tools/ubench/store_patterns
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(64, 2) store_kernel(double2 *out, long long stride_units, long long wave_span_units, int iters, int per_lane) {
  const int lane = threadIdx.x & 63;
  const long long wave = blockIdx.x;
  double2 v = make_double2(1.0 + lane, 2.0);
  long long at = wave * wave_span_units * iters + lane * stride_units;
  for (int i = 0; i < iters; i++) {
    for (int k = 0; k < per_lane; k++) out[at + k] = v;
    at += wave_span_units;
  }
}

int main() {
  const int waves = 256 * 8, iters = 500;
  const long long max_span = 40 * 64 + 4;                    // the widest case below (last lane's four stores included), in 16-byte units per wave and iteration
  const size_t units = (size_t)waves * iters * max_span + 64;   // every case stays inside waves * iters * span <= this
  double2 *d = nullptr;
  if (hipMalloc(&d, units * 16) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  hipMemset(d, 0, units * 16);
  struct Case { const char *name; long long stride; int per_lane; } cases[] = {
      {"contiguous 16 B per lane (1 store)", 1, 1},
      {"32-B records side by side (2 stores: halves)", 2, 2},
      {"64-B records side by side (4 stores)", 4, 4},
      {"stride 320 B, 2 stores of 16 B (one 32-B record half pair)", 20, 2},
      {"stride 320 B, 4 stores", 20, 4},
      {"stride 640 B, 4 stores", 40, 4},
  };
  for (auto &c : cases) {
    const long long span = (c.stride * 64 > 64 * (long long)c.per_lane ? c.stride * 64 : 64 * (long long)c.per_lane) + 4;
    if (span > max_span) { printf("span too large\n"); return 1; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(store_kernel, dim3(waves), dim3(64), 0, 0, d, c.stride, span, 10, c.per_lane);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(store_kernel, dim3(waves), dim3(64), 0, 0, d, c.stride, span, iters, c.per_lane);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_cu = (double)iters * c.per_lane * 8;
    const double bytes = (double)waves * iters * c.per_lane * 64 * 16;
    printf("%-62s %.3f ms  %.1f ns per wave store per CU  %.2f TB/s of stored bytes\n", c.name, ms, ms * 1e6 / instr_per_cu, bytes / ms / 1e9);
  }
  return 0;
}

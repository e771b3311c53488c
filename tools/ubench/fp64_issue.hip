// fp64 issue-rate / latency microbenchmark for gfx950: dependent chains vs independent chains, at 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int kChains, int kOp>
__global__ void chain_kernel(double *out, double a, double b, int iters) {
  double x[kChains];
#pragma unroll
  for (int c = 0; c < kChains; c++) x[c] = a + c + threadIdx.x * 1e-9;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
#pragma unroll
      for (int c = 0; c < kChains; c++) {
        if (kOp == 0) x[c] = __builtin_fma(x[c], b, a);
        else if (kOp == 1) x[c] = x[c] * b;
        else if (kOp == 2) x[c] = x[c] + b;
        else x[c] = __builtin_amdgcn_rcp(x[c]);
      }
    }
  }
  double s = 0.0;
#pragma unroll
  for (int c = 0; c < kChains; c++) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int kChains, int kOp>
void run(const char *name, int waves_per_simd, double *d_out) {
  int cus = 256;
  int iters = 20000;
  dim3 grid(cus * waves_per_simd), block(256);   // 256 threads = 4 waves = one per SIMD per block
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((chain_kernel<kChains, kOp>), grid, block, 0, 0, d_out, 1.0000001, 0.9999999, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((chain_kernel<kChains, kOp>), grid, block, 0, 0, d_out, 1.0000001, 0.9999999, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double instr_per_wave = (double)iters * 16 * kChains;
  double ns_per_instr_per_simd = ms * 1e6 / (instr_per_wave * waves_per_simd);
  printf("%-6s chains=%d waves/SIMD=%d : %.3f ms, %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n", name, kChains, waves_per_simd, ms,
         ns_per_instr_per_simd, ns_per_instr_per_simd * 2.4);
}

int main() {
  double *d_out; hipMalloc(&d_out, sizeof(double) * 256 * 8 * 256);
  for (int w = 1; w <= 4; w *= 2) {
    run<1, 0>("fma", w, d_out); run<2, 0>("fma", w, d_out); run<4, 0>("fma", w, d_out); run<8, 0>("fma", w, d_out);
    run<1, 1>("mul", w, d_out); run<4, 1>("mul", w, d_out);
    run<1, 2>("add", w, d_out); run<4, 2>("add", w, d_out);
    run<1, 3>("rcp", w, d_out); run<4, 3>("rcp", w, d_out);
  }
  return 0;
}

// Do two kernels from two streams share SIMDs? A: dependent fp64 chain, 1 wave/SIMD, padded to ~330 registers.
// B: same chain, small register footprint. Time A alone, B alone, A||B.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int kPad>
__global__ void __launch_bounds__(256, 1) chain_kernel(double *out, double a, double b, int iters) {
  double x = a + threadIdx.x * 1e-9;
  double pad[kPad > 0 ? kPad : 1];
#pragma unroll
  for (int c = 0; c < kPad; c++) pad[c] = a * (c + 1);
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) x = __builtin_fma(x, b, a);
    if (kPad > 0) {
#pragma unroll
      for (int c = 0; c < kPad; c++) asm volatile("" : "+v"(pad[c]));   // keep the padding registers live
    }
  }
  double s = x;
#pragma unroll
  for (int c = 0; c < kPad; c++) s += pad[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  double *d_a, *d_b; hipMalloc(&d_a, 8 * 256 * 2048); hipMalloc(&d_b, 8 * 256 * 2048);
  hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  hipEvent_t e0, e1, e2, e3; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2); hipEventCreate(&e3);
  const int iters = 40000;
  auto A = [&](hipStream_t s) { hipLaunchKernelGGL(chain_kernel<150>, dim3(256), dim3(256), 0, s, d_a, 1.0000001, 0.9999999, iters); };
  auto B = [&](hipStream_t s) { hipLaunchKernelGGL(chain_kernel<0>, dim3(256), dim3(256), 0, s, d_b, 1.0000001, 0.9999999, iters); };
  A(sa); B(sb); hipDeviceSynchronize();
  float ms;
  hipEventRecord(e0, sa); A(sa); hipEventRecord(e1, sa); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); printf("A alone (big regs)   %.3f ms\n", ms);
  hipEventRecord(e0, sb); B(sb); hipEventRecord(e1, sb); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); printf("B alone (small regs) %.3f ms\n", ms);
  hipDeviceSynchronize();
  hipEventRecord(e0, sa); hipEventRecord(e2, sb);
  A(sa); B(sb);
  hipEventRecord(e1, sa); hipEventRecord(e3, sb);
  hipDeviceSynchronize();
  float ma, mb, mab;
  hipEventElapsedTime(&ma, e0, e1); hipEventElapsedTime(&mb, e2, e3); hipEventElapsedTime(&mab, e0, e3);
  printf("A||B: A %.3f ms, B %.3f ms, first start to last end %.3f ms\n", ma, mb, mab);
  // same stream back to back for reference
  hipEventRecord(e0, sa); A(sa); B(sa); hipEventRecord(e1, sa); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); printf("A;B serial %.3f ms\n", ms);
  return 0;
}

// Issue cost of the VALU operations the hot kernels are made of, gfx950: 8 independent chains per lane, 2 waves per SIMD,
// cycles per wave-instruction per SIMD (clock taken from the fma line = 4 cycles nominal).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/op_rates.hip -o tools/ubench/op_rates && tools/ubench/op_rates
#include <hip/hip_runtime.h>
#include <cstdio>

enum Op { FMA, MUL, ADD, RCP, RSQ, SQRT, CVT_F64_F32, CVT_F32_F64, CVT_F64_I32, CVT_I32_F64, CNDMASK64, CMP_SEL, DIV_FIXUP, LDEXP, FREXP_MANT, RNDNE, FMA32, MAX64,
          TRIG_PREOP, MUL_LO_U32, ADD_U32, MAD_U64, FLOOR64, FRACT64, MIN64, CMP_CLASS, MOV64, PK_FMA32, RCP32, N_OPS };
const char *kNames[N_OPS] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_cvt_f64_f32", "v_cvt_f32_f64", "v_cvt_f64_i32",
                             "v_cvt_i32_f64", "v_cndmask x2 (64-bit select)", "v_cmp_f64 + 2 cndmask", "v_div_fixup_f64", "v_ldexp_f64", "v_frexp_mant_f64", "v_rndne_f64", "v_fma_f32",
                             "v_max_f64", "v_trig_preop_f64", "v_mul_lo_u32", "v_add_u32", "v_mad_u64_u32", "v_floor_f64", "v_fract_f64", "v_min_f64", "v_cmp_class_f64+sel", "v_mov_b64", "v_pk_fma_f32", "v_rcp_f32"};

template <int kOp>
__global__ void __launch_bounds__(256, 2) rate_kernel(double *out, double a, double b, int iters) {
  constexpr int kChains = 8;
  double x[kChains];
  float xf[kChains];
  int xi[kChains];
#pragma unroll
  for (int c = 0; c < kChains; c++) {
    x[c] = a + c + threadIdx.x * 1e-9;
    xf[c] = (float)x[c];
    xi[c] = c + threadIdx.x;
  }
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
#pragma unroll
      for (int c = 0; c < kChains; c++) {
        if (kOp == FMA) x[c] = __builtin_fma(x[c], b, a);
        else if (kOp == MUL) x[c] = x[c] * b;
        else if (kOp == ADD) x[c] = x[c] + b;
        else if (kOp == RCP) x[c] = __builtin_amdgcn_rcp(x[c]);
        else if (kOp == RSQ) x[c] = __builtin_amdgcn_rsq(x[c]);
        else if (kOp == SQRT) x[c] = __builtin_amdgcn_sqrt(x[c]);
        else if (kOp == CVT_F64_F32) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x[c]) : "v"(xf[c])); asm volatile("" : "+v"(xf[c]) : "v"(x[c])); }
        else if (kOp == CVT_F32_F64) { asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(xf[c]) : "v"(x[c])); asm volatile("" : "+v"(x[c]) : "v"(xf[c])); }
        else if (kOp == CVT_F64_I32) { asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(x[c]) : "v"(xi[c])); asm volatile("" : "+v"(xi[c]) : "v"(x[c])); }
        else if (kOp == CVT_I32_F64) { asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(xi[c]) : "v"(x[c])); asm volatile("" : "+v"(x[c]) : "v"(xi[c])); }
        else if (kOp == CNDMASK64) { int lo = __double2loint(x[c]), hi = __double2hiint(x[c]); asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %2, vcc" : "+v"(lo), "+v"(hi) : "v"(u)); x[c] = __hiloint2double(hi, lo); }
        else if (kOp == CMP_SEL) x[c] = x[c] < b ? x[c] + 0.0 * a : a;
        else if (kOp == DIV_FIXUP) x[c] = __builtin_amdgcn_div_fixup(x[c], b, a);
        else if (kOp == LDEXP) x[c] = __builtin_amdgcn_ldexp(x[c], 1);
        else if (kOp == FREXP_MANT) x[c] = __builtin_amdgcn_frexp_mant(x[c]);
        else if (kOp == RNDNE) { asm volatile("v_rndne_f64 %0, %1" : "=v"(x[c]) : "v"(x[c])); }
        else if (kOp == FMA32) xf[c] = __builtin_fmaf(xf[c], (float)b, (float)a);
        else if (kOp == MAX64) { asm volatile("v_max_f64 %0, %1, %2" : "=v"(x[c]) : "v"(x[c]), "v"(b)); }
        else if (kOp == MIN64) { asm volatile("v_min_f64 %0, %1, %2" : "=v"(x[c]) : "v"(x[c]), "v"(b)); }
        else if (kOp == TRIG_PREOP) x[c] = __builtin_amdgcn_trig_preop(x[c], 1);
        else if (kOp == MUL_LO_U32) xi[c] = xi[c] * 3;
        else if (kOp == ADD_U32) { asm volatile("v_add_u32 %0, %1, %2" : "=v"(xi[c]) : "v"(xi[c]), "v"(u)); }
        else if (kOp == MAD_U64) { unsigned long long t; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(t) : "v"(xi[c]), "v"(u), "v"((unsigned long long)xi[c]) : "vcc"); xi[c] = (int)t; }
        else if (kOp == FLOOR64) { asm volatile("v_floor_f64 %0, %1" : "=v"(x[c]) : "v"(x[c])); }
        else if (kOp == FRACT64) { asm volatile("v_fract_f64 %0, %1" : "=v"(x[c]) : "v"(x[c])); }
        else if (kOp == CMP_CLASS) { x[c] = __builtin_isnan(x[c]) ? a : x[c] + 0.0; }
        else if (kOp == MOV64) { double y; asm volatile("v_mov_b64 %0, %1" : "=v"(y) : "v"(x[c])); x[c] = y; }
        else if (kOp == PK_FMA32) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(x[c]) : "v"(x[c]), "v"(b), "v"(a)); }
        else if (kOp == RCP32) xf[c] = __builtin_amdgcn_rcpf(xf[c]);
      }
    }
  }
  double s = 0.0;
#pragma unroll
  for (int c = 0; c < kChains; c++) s += x[c] + xf[c] + xi[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int kOp>
double run(double *d_out) {
  const int cus = 256, waves_per_simd = 2, iters = 4000;
  dim3 grid(cus * waves_per_simd), block(256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((rate_kernel<kOp>), grid, block, 0, 0, d_out, 1.0000001, 0.9999999, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((rate_kernel<kOp>), grid, block, 0, 0, d_out, 1.0000001, 0.9999999, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double ops_per_wave = (double)iters * 8 * 8;
  return ms * 1e6 / (ops_per_wave * waves_per_simd);   // ns per op (group) per SIMD
}

int main() {
  double *d_out; hipMalloc(&d_out, sizeof(double) * 256 * 2 * 256);
  double ns[N_OPS];
#define R(O) ns[O] = run<O>(d_out);
  R(FMA) R(MUL) R(ADD) R(RCP) R(RSQ) R(SQRT) R(CVT_F64_F32) R(CVT_F32_F64) R(CVT_F64_I32) R(CVT_I32_F64) R(CNDMASK64) R(CMP_SEL) R(DIV_FIXUP) R(LDEXP) R(FREXP_MANT) R(RNDNE)
  R(FMA32) R(MAX64) R(TRIG_PREOP) R(MUL_LO_U32) R(ADD_U32) R(MAD_U64) R(FLOOR64) R(FRACT64) R(MIN64) R(CMP_CLASS) R(MOV64) R(PK_FMA32) R(RCP32)
  for (int o = 0; o < N_OPS; o++) printf("%-30s %.3f ns = %.2f x fma\n", kNames[o], ns[o], ns[o] / ns[FMA]);
  return 0;
}

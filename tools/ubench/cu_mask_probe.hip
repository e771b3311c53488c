// Which compute units a CU-masked stream reaches on this device (hipExtStreamCreateWithCUMask): for masks with the lowest K bits
// set - and for their complements - the set of (XCC, shader engine, CU) its waves report (s_getreg HW_ID / XCC_ID).
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench/cu_mask_probe tools/ubench/cu_mask_probe.hip && tools/ubench/cu_mask_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <map>
#include <set>
#include <vector>

__global__ void probe(unsigned int *out, int spin) {
  unsigned int hw_id, xcc_id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
  double x = threadIdx.x;
  for (int i = 0; i < spin; i++) x = x * 1.0000001 + 1e-9;   // keep the wave resident while the others are dispatched
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = hw_id;
    out[2 * blockIdx.x + 1] = (xcc_id & 0xf) | (x > 1e300 ? 16u : 0u);
  }
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount, words = (cus + 31) / 32, blocks = cus * 8;
  unsigned int *d = nullptr;
  hipMalloc(&d, blocks * 2 * sizeof(unsigned int));
  std::vector<unsigned int> h(blocks * 2);
  for (int k : {8, 24, 32, 40, 64, 128}) {
    for (int complement = 0; complement < 2; complement++) {
      std::vector<uint32_t> mask(words, 0u);
      for (int c = 0; c < cus; c++)
        if ((c < k) != (complement != 0)) mask[c / 32] |= 1u << (c % 32);
      hipStream_t s;
      if (hipExtStreamCreateWithCUMask(&s, words, mask.data()) != hipSuccess) { std::printf("mask refused\n"); return 1; }
      hipMemsetAsync(d, 0xff, blocks * 2 * sizeof(unsigned int), s);
      hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 0, s, d, 200000);
      hipStreamSynchronize(s);
      hipMemcpy(h.data(), d, blocks * 2 * sizeof(unsigned int), hipMemcpyDeviceToHost);
      std::map<int, std::set<int>> per_xcc;   // xcc -> (se, sh, cu) ids seen
      for (int b = 0; b < blocks; b++) {
        const unsigned int hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        const int cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;   // gfx9 HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
        per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
      }
      int total = 0;
      std::printf("%s %3d bits: ", complement ? "all but the lowest" : "the lowest        ", k);
      for (auto &kv : per_xcc) { std::printf("xcc%d:%zu ", kv.first, kv.second.size()); total += (int)kv.second.size(); }
      std::printf(" = %d distinct CUs\n", total);
      hipStreamDestroy(s);
    }
  }
  hipFree(d);
  return 0;
}

// Microbenchmark: do one wave's VALU instructions issue beside another wave's MFMAs on the SAME SIMD?
// 8 waves per workgroup, one workgroup per CU: waves 0-3 land on SIMDs 0-3, waves 4-7 are their partners.
// mode 0: all 8 waves run MFMA; 1: all run VALU; 2: waves 0-3 MFMA, 4-7 VALU; 3: only 0-3 MFMA; 4: only 4-7 VALU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ void __launch_bounds__(512) k(float* out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = (mode == 0) || ((mode == 2 || mode == 3) && wave < 4);
  const bool do_valu = (mode == 1) || ((mode == 2 || mode == 4) && wave >= 4);
  f32x4 acc = {0, 0, 0, 0};
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(j + 1); }
  float v0 = threadIdx.x, v1 = 1.0f, v2 = 2.0f, v3 = 3.0f;
  if (do_mfma) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    }
  }
  if (do_valu) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {   // 4 independent fma chains = 64 VALU per iteration
        v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 0.9999f, 0.25f); v2 = fmaf(v2, 1.0002f, 0.125f); v3 = fmaf(v3, 0.9998f, 0.0625f);
      }
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + v0 + v1 + v2 + v3;
}
int main() {
  float* d; hipMalloc(&d, 256 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int mode = 0; mode < 5; ++mode) {
    k<<<256, 512>>>(d, 100, mode); hipDeviceSynchronize();
    hipEventRecord(e0); k<<<256, 512>>>(d, iters, mode); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d: %.3f ms  (per iteration: %.1f ns; 16 MFMA 16x16x32 = 256 pipe cycles, 64 VALU)\n", mode, ms, ms * 1e6 / iters);
  }
  return 0;
}

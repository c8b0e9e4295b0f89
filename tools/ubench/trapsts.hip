// Does a saturating f32 -> f16 conversion (v_cvt_pkrtz_f16_f32) leave the sticky OVERFLOW bit in TRAPSTS.EXCP on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* in, unsigned* out, int mode) {
  float v = in[threadIdx.x];
  unsigned before = __builtin_amdgcn_s_getreg((3) | (0 << 6) | ((9 - 1) << 11));   // HW_REG_TRAPSTS id 3, offset 0, size 9
  unsigned r = 0;
  if (mode == 0) { auto h = __builtin_amdgcn_cvt_pkrtz(v, v); r = __builtin_bit_cast(unsigned, h); }
  else if (mode == 1) { r = __float_as_uint(v * v); }
  else { r = __float_as_uint(v + 1.0f); }
  asm volatile("" ::"v"(r));
  unsigned after = __builtin_amdgcn_s_getreg((3) | (0 << 6) | ((9 - 1) << 11));
  if (threadIdx.x == 0) { out[0] = before; out[1] = after; out[2] = r; }
}
int main() {
  float* d; unsigned* o; hipMalloc(&d, 256); hipMalloc(&o, 64);
  for (int mode = 0; mode < 3; ++mode) for (float x : {1.0f, 1e6f, 1e30f}) {
    float h[64]; for (int i = 0; i < 64; ++i) h[i] = x;
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o, mode);
    unsigned r[3]; hipMemcpy(r, o, 12, hipMemcpyDeviceToHost);
    printf("mode %d x %g: TRAPSTS.EXCP before 0x%x after 0x%x result 0x%x\n", mode, x, r[0], r[1], r[2]);
  }
  return 0;
}

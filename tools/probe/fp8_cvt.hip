#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* x, unsigned* o, int n) {
  int i = threadIdx.x;
  if (i < n) {
    unsigned r = 0;
    r = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], r, false);   // low 16 bits: two fp8
    o[i] = r;
  }
}
int main() {
  float h[16] = {1.f, -1.f, 448.f, 500.f, 0.5f, 0.015625f, 240.f, 256.f, 1e-3f, 3.3f, 0.f, -0.f, 1000.f, -1000.f, 7.5f, 0.0019f};
  float* d; unsigned* o; hipMalloc(&d, sizeof h); hipMalloc(&o, 8 * 4);
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, o, 8);
  unsigned r[8]; hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
  for (int i = 0; i < 8; ++i) printf("%g -> 0x%02x   %g -> 0x%02x\n", h[2 * i], r[i] & 0xff, h[2 * i + 1], (r[i] >> 8) & 0xff);
  return 0;
}

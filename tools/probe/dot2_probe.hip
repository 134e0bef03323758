// Probe: numerical behaviour of v_dot2c_f32_bf16 on gfx950 (used for the softmax row sums).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
__global__ void k(const unsigned* a, const unsigned* b, const float* c, float* o, int n) {
  int i = threadIdx.x;
  if (i < n) o[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, a[i]), __builtin_bit_cast(bf2, b[i]), c[i], false);
}
static unsigned short f2b(float f) { union { float f; unsigned u; } c; c.f = f; return (unsigned short)(c.u >> 16); }
int main() {
  const int n = 8;
  float lo[n] = {1.f, 0.5f, 0.25f, 3.f, 1e-3f, 100.f, 0.75f, 2.f};
  float hi[n] = {2.f, 0.125f, 4.f, 5.f, 2e-3f, 0.5f, 0.015625f, 2.f};
  float cc[n] = {0.f, 1.f, 10.f, 0.f, 0.f, 1000.f, 3.f, 31.f};
  unsigned ha[n], hb[n]; 
  for (int i = 0; i < n; ++i) { ha[i] = f2b(lo[i]) | ((unsigned)f2b(hi[i]) << 16); hb[i] = 0x3f803f80u; }
  unsigned *da, *db; float *dc, *dout;
  hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dout, n * 4);
  hipMemcpy(da, ha, n * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb, n * 4, hipMemcpyHostToDevice); hipMemcpy(dc, cc, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, dout, n);
  float out[n]; hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("lo=%g hi=%g c=%g -> %g (expect %g)\n", lo[i], hi[i], cc[i], out[i], lo[i] + hi[i] + cc[i]);
  return 0;
}

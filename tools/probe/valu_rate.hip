// Probe: issue cost of the VALU instructions that share the attention loop with the MFMAs (gfx950), in cycles per instruction and wave,
// with 1, 2 and 4 waves per SIMD of one CU: 8 independent chains per wave, 2^20 instructions, HIP-event time (clock64 is a fixed-rate counter).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ void k(float* out, long* cyc, int iters) {
  float a[8]; f32x2 p[8];
  for (int i = 0; i < 8; ++i) { a[i] = 0.001f * (threadIdx.x + i); p[i] = (f32x2){a[i], a[i] + 1.f}; }
  const long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#define X0(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
#define X1(i) asm volatile("v_exp_legacy_f32 %0, %0" : "+v"(a[i]));
#define X2(i) asm volatile("v_exp_f16 %0, %0" : "+v"(a[i]));
#define X3(i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(a[i]));
#define X4(i) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p[i]));
#define X5(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
#define X6(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(a[i]));
#define X7(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define X8(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
    if (OP == 0) { REP8(X0) REP8(X0) } else if (OP == 1) { REP8(X1) REP8(X1) } else if (OP == 2) { REP8(X2) REP8(X2) }
    else if (OP == 3) { REP8(X3) REP8(X3) } else if (OP == 4) { REP8(X4) REP8(X4) } else if (OP == 5) { REP8(X5) REP8(X5) }
    else if (OP == 6) { REP8(X6) REP8(X6) } else if (OP == 7) { REP8(X7) REP8(X7) } else { REP8(X8) REP8(X8) }
  }
  const long t1 = clock64();
  float s = 0.f; for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP> void run(const char* name, float* out, long* cyc) {
  for (int threads : {256, 512, 1024}) {           // 1, 2, 4 waves per SIMD on one CU
    const int iters = 1 << 16;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<1, threads>>>(out, cyc, iters); hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<1, threads>>>(out, cyc, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    (void)h;
    printf("%-18s %d waves/SIMD: %6.2f ns per instruction and wave, %5.2f ns per instruction and SIMD\n", name, threads / 256,
           ms * 1e6 / (iters * 16.0), ms * 1e6 / (iters * 16.0) / (threads / 256));
  }
}
int main() {
  float* out; long* cyc; hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 8);
  run<0>("v_exp_f32", out, cyc); run<1>("v_exp_legacy_f32", out, cyc); run<2>("v_exp_f16", out, cyc); run<7>("v_rcp_f32", out, cyc);
  run<3>("v_add_f32", out, cyc); run<8>("v_fma_f32", out, cyc); run<4>("v_pk_add_f32", out, cyc); run<5>("v_pk_fma_f32", out, cyc);
  run<6>("v_cvt_pk_bf16_f32", out, cyc);
  return 0;
}

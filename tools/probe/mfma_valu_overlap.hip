// Probe: how well do VALU work and MFMA work of the attention loop's mix overlap on one SIMD, as a function of the number of
// resident waves?  Per iteration and wave: 36 x v_mfma_f32_16x16x32_bf16 (independent accumulators), 32 x v_exp_f32, 16 x
// v_cvt_pk_bf16_f32, interleaved one MFMA : one exp (: half a cvt) as ld_attn_q64.hip does; registers only, no memory.
// Modes: 0 MFMA only, 1 VALU only, 2 both.  One workgroup per CU (256 CUs), 4 / 8 / 12 / 16 waves = 1 .. 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
#define FENCE() __builtin_amdgcn_sched_barrier(0)
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
  f32x4_t acc[8];
  bf16x8_t a, b;
  float e[32];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x % 7); b[i] = (short)(0x3f00 + i); }
  for (int i = 0; i < 32; ++i) e[i] = 0.001f * (threadIdx.x + i);
  unsigned pk = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 36; ++i) {
      if (MODE != 1) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 7], 0, 0, 0);
      if (MODE != 0 && i < 32) {
        asm volatile("v_exp_f32 %0, %0" : "+v"(e[i]));
        if (i & 1) { unsigned r; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(e[i - 1]), "v"(e[i])); pk ^= r; }
      }
      FENCE();
    }
  }
  float s = __uint_as_float(pk & 0x3f800000u);
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 32; ++i) s += e[i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> float run(int threads, float* out, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<256, threads>>>(out, iters); hipDeviceSynchronize();
  hipEventRecord(e0); k<MODE><<<256, threads>>>(out, iters); hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  float* out; hipMalloc(&out, 256 * 1024 * 4);
  const int iters = 20000;
  printf("whole chip busy (256 workgroups): ns per iteration and SIMD-resident wave (one iteration = 36 MFMA 16x16x32 + 32 exp + 16 cvt_pk)\n");
  for (int threads : {256, 512, 768, 1024}) {
    const int wps = threads / 256;
    const float m = run<0>(threads, out, iters), v = run<1>(threads, out, iters), b = run<2>(threads, out, iters);
    const double f = 1e6 / iters / wps;        // ns of SIMD time per wave-iteration
    printf("%d waves/SIMD: MFMA only %7.1f  VALU only %7.1f  both %7.1f   -> both = MFMA + %.2f x VALU\n", wps, m * f, v * f, b * f,
           (b - m) / v);
  }
  return 0;
}

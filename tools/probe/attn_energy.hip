// What the attention kernel's non-MFMA work costs under the power / current governor: the 16x16x32 loop of mfma_power.hip
// (one wave per SIMD, 128x128 fp32 accumulators, operands re-read from LDS every k-step) with, per 64 MFMAs,
//   NREAD ds_read_b128 (16 = each fragment read once; the attention kernel reads 28 per 64 MFMAs),
//   NEXP  v_exp_f32    (the attention kernel issues 57 per 64 MFMAs),
//   NCVT  v_cvt_pk_bf16_f32 (28 per 64 MFMAs in the attention kernel).
// Caveat (measured): with one wave per SIMD the inline-asm additions make the loop issue-bound (2.39 GHz, below the power cap),
// so the variants do not read as energy; the attention ablation (tools/attn_ablate.sh) replaced this probe.
// No global-memory traffic.  Build: hipcc --offload-arch=gfx950 -O3 -o attn_energy attn_energy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NREAD, int NEXP, int NCVT>
__global__ __launch_bounds__(256, 1) void loop(const bf16x8* __restrict__ src, float* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 65536 / 16; i += 256) ((bf16x8*)smem)[i] = src[(blockIdx.x * 4096 + i) % (1 << 16)];
  __syncthreads();
  const char* lbase = smem + wave * 16384 + lane * 16;
  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
  bf16x8 a[8], b[8];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        a[i] = *(const bf16x8*)(lbase + ((p * 16 + i) & 15) * 1024);
        b[i] = *(const bf16x8*)(lbase + ((p * 16 + 8 + i) & 15) * 1024);
      }
#pragma unroll
      for (int x = 0; x < NREAD - 16; ++x) {                      // extra fragment reads, kept alive but unused
        bf16x8 t = *(const volatile bf16x8*)(lbase + ((p * 16 + x) & 15) * 1024);
        asm volatile("" ::"v"(t));
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
          const int k = i * 8 + j;
          if ((k * NEXP) / 64 != ((k + 1) * NEXP) / 64) {          // NEXP of the 64 slots
            float y;
            asm volatile("v_exp_f32 %0, %1" : "=v"(y) : "v"(__builtin_bit_cast(f32x4, a[(i + 4) & 7])[k & 3]));
            asm volatile("" ::"v"(y));
          }
          if ((k * NCVT) / 64 != ((k + 1) * NCVT) / 64) {
            unsigned y;
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(y) : "v"(__builtin_bit_cast(f32x4, b[(j + 3) & 7])[k & 3]), "v"(__builtin_bit_cast(f32x4, a[(i + 5) & 7])[(k + 1) & 3]));
            asm volatile("" ::"v"(y));
          }
        }
    }
  }
  float total = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) total += acc[i][j][r];
  if (total == 1234.5678f) sink[tid] = total;
}

template <int NREAD, int NEXP, int NCVT>
void run(const bf16x8* src, float* sink) {
  const int iters = 2000, wgs = 256 * 4;
  hipFuncSetAttribute((const void*)loop<NREAD, NEXP, NCVT>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const double flop_launch = (double)wgs * 4 * iters * 2.0 * 128 * 128 * 64;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) loop<NREAD, NEXP, NCVT><<<wgs, 256, 65536>>>(src, sink, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0); loop<NREAD, NEXP, NCVT><<<wgs, 256, 65536>>>(src, sink, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms1; hipEventElapsedTime(&ms1, e0, e1);
  const int n = (int)(1500.f / ms1) + 1;
  hipEventRecord(e0);
  for (int i = 0; i < n; ++i) loop<NREAD, NEXP, NCVT><<<wgs, 256, 65536>>>(src, sink, iters);
  hipEventRecord(e1);
  if (system("sleep 0.9; rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'sclk|Package Power' | sed 's/^/      /'")) {}
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("per 64 MFMA: %2d ds_read_b128, %2d v_exp_f32, %2d v_cvt_pk_bf16_f32   %7.3f ms/launch  %6.0f TFLOP/s\n", NREAD, NEXP, NCVT, ms / n,
         flop_launch * n / (ms * 1e-3) / 1e12);
  fflush(stdout);
}

int main() {
  std::vector<unsigned short> h((size_t)(1 << 16) * 8);
  srand(1);
  for (auto& x : h) {
    float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = rand() / (float)RAND_MAX;
    float g = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2) * 0.05f;
    unsigned bits; memcpy(&bits, &g, 4);
    x = (unsigned short)(bits >> 16);
  }
  bf16x8* src; float* sink;
  hipMalloc(&src, h.size() * 2); hipMalloc(&sink, 4096);
  hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  run<16, 0, 0>(src, sink);
  run<32, 0, 0>(src, sink);
  run<48, 0, 0>(src, sink);
  run<16, 32, 0>(src, sink);
  run<16, 64, 0>(src, sink);
  run<16, 0, 32>(src, sink);
  run<32, 64, 32>(src, sink);
  run<16, 0, 0>(src, sink);
  return 0;
}

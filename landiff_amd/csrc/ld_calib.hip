// Calibration loops for bench.py -- NOT on the product path.  Two launches that tell a reader of a bench line what the box it
// ran on sustains: an MFMA-only loop (v_mfma_f32_16x16x32_bf16 on caller-supplied operands, no memory traffic inside the loop)
// and a 16-bytes-per-lane streaming read.  bench.py runs each for ~1 s before the timed region and prints the rates next to
// the roofline fractions (`calibration`, `frac_of_box_ceiling`): the boxes of the pool differ by +-4-5 % on power-limited
// kernels, which is more than most single optimisations move the number.
#include "ld_common.h"
#include "../../include/landiff_hip.h"

namespace {

// One wave per SIMD, an 8 x 8 grid of 16x16 accumulators per wave (256 registers), operands from a two-set register pool that is
// loaded once: per loop trip 2 x 64 MFMAs = 2 x 64 x 16384 FLOP per wave.  (tools/probe/mfma_power.hip, variant 1.)
__global__ __launch_bounds__(256, 1) void ld_calib_mfma_kernel(const bf16x8_t* __restrict__ src, long n_frag, float* sink, int iters) {
  const int tid = threadIdx.x;
  f32x4_t acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  bf16x8_t a[2][8], b[2][8];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      a[p][i] = src[(((long)blockIdx.x * 32 + p * 16 + i) * 256 + tid) % n_frag];
      b[p][i] = src[(((long)blockIdx.x * 32 + p * 16 + 8 + i) * 256 + tid + 7777) % n_frag];
    }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[p][i], b[p][j], acc[i][j], 0, 0, 0);
  }
  float total = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) total += acc[i][j][r];
  if (total == 123.456f) sink[0] = total;          // keeps the loop alive, practically never stores
}

// grid-stride read, four 16-byte loads in flight per lane (non-temporal: nothing is reused)
__global__ __launch_bounds__(256) void ld_calib_read_kernel(const u32x4_t* __restrict__ p, long n16, uint32_t* sink) {
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    u32x4_t v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
  }
  for (; i < n16; i += stride) { const u32x4_t v = p[i]; acc += v[0] ^ v[1] ^ v[2] ^ v[3]; }
  if (acc == 0x12345678u) *sink = acc;
}

}  // namespace

LD_API int ld_calib_mfma_bf16(const void* operands, int64_t operand_bytes, float* sink, int64_t n_workgroups, int64_t iters,
                              double* flops_out, void* stream) {
  LD_REQUIRE(operands && sink && operand_bytes >= (1 << 20) && operand_bytes % 16 == 0, "ld_calib_mfma_bf16: >= 1 MiB of bf16 operands, 16-byte multiple");
  LD_REQUIRE(n_workgroups > 0 && iters > 0 && iters < (1 << 30), "ld_calib_mfma_bf16: bad launch size");
  hipLaunchKernelGGL(ld_calib_mfma_kernel, dim3((unsigned)n_workgroups), dim3(256), 0, (hipStream_t)stream,
                     (const bf16x8_t*)operands, (long)(operand_bytes / 16), sink, (int)iters);
  if (flops_out) *flops_out = (double)n_workgroups * 4.0 * (double)iters * 128.0 * 16384.0;      // 4 waves x 128 MFMAs x 2*16*16*32
  return ld_check_launch("ld_calib_mfma_bf16");
}

LD_API int ld_calib_stream_read(const void* buf, int64_t bytes, uint32_t* sink, void* stream) {
  LD_REQUIRE(buf && sink && bytes >= 16 && bytes % 16 == 0 && ((uintptr_t)buf & 15) == 0, "ld_calib_stream_read: 16-byte aligned buffer, 16-byte multiple");
  hipLaunchKernelGGL(ld_calib_read_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, (const u32x4_t*)buf, (long)(bytes / 16), sink);
  return ld_check_launch("ld_calib_stream_read");
}

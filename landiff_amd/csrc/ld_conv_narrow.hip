// Causal 3x3x3 convolution with a handful of output channels (the VAE's conv_out: 128 -> 3 at 480 x 720), channels-last.
//
// Replaces, for this shape, the implicit-GEMM route of ld_conv_cl_bf16 (landiff/diffusion/vae_modules/cp_enc_dec.py:416-473 via
// ContextParallelDecoder3D.conv_out, :1066).  As a GEMM it is M = T*H*W x N = 3 x K = 27*128: the 256 x 128 MFMA tile computes 128
// columns for 3, and -- worse -- walks the 891 MB padded input once per filter tap (27 x 1 KB LDS-DMA pieces per 8 output
// positions): 2.1 ms per chunk, 27 TFLOP/s, 13 ms of a video's VAE decode (profiles/r03_vae_conv_shapes.txt).  The operation is
// a read of the input; this kernel reads it about once:
//   * a workgroup (4 waves) owns a 4 x 16 spatial tile for ALL frames and walks time: a ring of four LDS slots holds the input
//     frames t, t+1, t+2 (+ the one being fetched) of its 6 x 18 halo window, 27 KB each, fetched by LDS-DMA (27 pieces of four
//     256-byte pixels; per-lane source addresses put 16-byte chunk c of pixel column x at position c ^ (x & 15), so that the 16
//     pixels of an MFMA operand, 256 bytes apart, read conflict-free);
//   * wave w owns input channels 32 w .. 32 w + 31 and keeps its 27 weight fragments (one per tap: [cout <= 4 (of 16)] x 32
//     channels) in 108 registers for the life of the workgroup; per frame it issues 27 taps x 4 rows of
//     v_mfma_f32_16x16x32_bf16 (weights x pixels: the accumulator is C^T, lane = pixel, register = output channel), 108 operand
//     reads of 1 KB from LDS;
//   * the four waves' partial sums (a split of K) meet in LDS in a fixed order, bias is added, the bf16 row goes out.
// Summation order differs from the GEMM route (K is split in four), so results agree with it to fp32 rounding before the bf16
// rounding of the output -- not bit for bit; run-to-run the kernel is deterministic.
#include "ld_common.h"
#include <stdlib.h>

namespace {

constexpr int CN_C = 128;                         // input channels
constexpr int CN_TH = 4, CN_TW = 16;              // output tile
constexpr int CN_HR = CN_TH + 2, CN_HC = CN_TW + 2;
constexpr int CN_NPIX = CN_HR * CN_HC;            // 108 halo pixels per frame
constexpr int CN_SLAB = CN_NPIX * CN_C * 2;       // 27 648 B
constexpr int CN_NSLOT = 4;
constexpr int CN_NPIECE = CN_NPIX / 4;            // 27 LDS-DMA pieces of 1 KB
constexpr int CN_RED = 4 * CN_TH * 4 * CN_TW * 4; // [wave][row][cout 0..3][pixel] fp32 = 4 KB
constexpr int CN_SMEM = CN_NSLOT * CN_SLAB + CN_RED;

struct NarrowParams {
  const bf16_t* in; const bf16_t* w; const bf16_t* bias; bf16_t* out;
  int T, H, W, Cout; long ldo;
  int in_bytes;
};

__global__ __launch_bounds__(256, 1) void ld_conv_narrow_kernel(NarrowParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = (float*)(smem + CN_NSLOT * CN_SLAB);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntx = p.W / CN_TW;
  const int ty = blockIdx.x / ntx, tx = blockIdx.x - ty * ntx;
  const int y0 = ty * CN_TH, x0 = tx * CN_TW;
  const int Hp = p.H + 2, Wp = p.W + 2;
  const long frame_bytes = (long)Hp * Wp * CN_C * 2;

  // ---- LDS-DMA sources: wave w stages pieces w, w + 4, ...; lane l of a piece = pixel 4 * piece + (l >> 4), position l & 15 ----
  uint32_t src_off[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int piece = wave + 4 * k;
    const int q = (piece < CN_NPIECE ? piece : 0) * 4 + (lane >> 4);
    const int r = q / CN_HC, c = q - r * CN_HC;
    const int chunk = (lane & 15) ^ (c & 15);
    src_off[k] = (uint32_t)((((long)(y0 + r) * Wp + x0 + c) * CN_C + chunk * 8) * 2);
  }
  auto stage_frame = [&](int f) {                          // padded input frame f -> slot f % 4
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const int soff = (int)(f * frame_bytes);
    char* slot = smem + (f & (CN_NSLOT - 1)) * CN_SLAB;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const int piece = wave + 4 * k;
      if (piece < CN_NPIECE)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(slot + piece * 1024), 16, src_off[k], soff, 0, 0);
    }
  };

  // ---- this wave's weight fragments: cout (lane & 15) < Cout, channels 32 * wave + 8 * (lane >> 4) .. + 7 of every tap ----
  bf16x8_t wf[27];
  {
    const int n = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      wf[tap] = (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
      if (n < p.Cout) wf[tap] = *(const bf16x8_t*)(p.w + ((long)n * 27 + tap) * CN_C + wave * 32 + kq * 8);
    }
  }
  // operand reads: pixel column i + dw of the halo window, chunk 4 * wave + kq at position chunk ^ (column & 15)
  int rd[3];
#pragma unroll
  for (int dw = 0; dw < 3; ++dw) {
    const int c = (lane & 15) + dw;
    rd[dw] = c * (CN_C * 2) + (((wave * 4 + (lane >> 4)) ^ (c & 15)) << 4);
  }
  const int o_y = (tid >> 4) & 3, o_x = tid & 15;              // reduce phase: thread = (row, pixel), all output channels

  const int nframes = p.T + 2;
  stage_frame(0); stage_frame(1); stage_frame(2);
  for (int t = 0; t < p.T; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of frame t + 2 (requested one step ago) have landed
    __syncthreads();                                        // ... everybody's; and slot (t + 3) % 4 was last read in step t - 1
    if (t + 3 < nframes) stage_frame(t + 3);
    f32x4_t acc[CN_TH];
#pragma unroll
    for (int y = 0; y < CN_TH; ++y) acc[y] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < 3; ++dt) {
      const char* slot = smem + ((t + dt) & (CN_NSLOT - 1)) * CN_SLAB;
#pragma unroll
      for (int dh = 0; dh < 3; ++dh)
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) {
          const int tap = (dt * 3 + dh) * 3 + dw;
#pragma unroll
          for (int y = 0; y < CN_TH; ++y) {
            const bf16x8_t a = *(const bf16x8_t*)(slot + rd[dw] + (y + dh) * (CN_HC * CN_C * 2));
            acc[y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[tap], a, acc[y], 0, 0, 0);
          }
        }
    }
    // C^T blocks: register r of lane l = output channel (l >> 4) * 4 + r of pixel l & 15: lanes 0-15 hold channels 0..3
    if (lane < 16) {
#pragma unroll
      for (int y = 0; y < CN_TH; ++y)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wave * CN_TH + y) * 4 + r) * CN_TW + lane] = acc[y][r];
    }
    __syncthreads();
    if (tid < CN_TH * CN_TW) {
      bf16_t* o = p.out + ((long)(t * p.H + y0 + o_y) * p.W + x0 + o_x) * p.ldo;
      for (int c = 0; c < p.Cout; ++c) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += red[((w * CN_TH + o_y) * 4 + c) * CN_TW + o_x];      // fixed order
        if (p.bias) s += bf2f(p.bias[c]);
        o[c] = f2bf(s);
      }
    }
  }
}

}  // namespace

// Called by ld_conv_cl_bf16 (ld_gemm.hip).  Returns 1 when the shape is not this kernel's (the caller takes the GEMM route), 0 after
// a launch, negative on error.  LD_CONV_NARROW=0 disables the route (A/B timing).
int ld_conv_narrow_try(const void* in_padded, const void* w, const void* bias, void* out, long ldo, long T, long H, long W, long Cin,
                       long Cout, long kT, long kH, long kW, bool plain_bias_epilogue, hipStream_t stream, bool dry_run) {
  static int k_on = LD_KNOB_UNSET;
  if (ld_knob("LD_CONV_NARROW", 1, &k_on) == 0) return 1;
  if (!(kT == 3 && kH == 3 && kW == 3 && Cin == CN_C && Cout >= 1 && Cout <= 4 && H % CN_TH == 0 && W % CN_TW == 0 && plain_bias_epilogue))
    return 1;
  const long bytes = (T + 2) * (H + 2) * (W + 2) * CN_C * 2;
  if (bytes >= 0x7fffffffL || ldo < Cout) return 1;
  if (dry_run) return 0;
  NarrowParams p{(const bf16_t*)in_padded, (const bf16_t*)w, (const bf16_t*)bias, (bf16_t*)out, (int)T, (int)H, (int)W, (int)Cout, ldo, (int)bytes};
  static thread_local LdSmemCache cache{};
  if (int rc = ld_ensure_dyn_smem((const void*)ld_conv_narrow_kernel, CN_SMEM, &cache)) return rc;
  hipLaunchKernelGGL(ld_conv_narrow_kernel, dim3((unsigned)((H / CN_TH) * (W / CN_TW))), dim3(256), CN_SMEM, stream, p);
  return ld_check_launch("ld_conv_cl_bf16(narrow)");
}

// bf16 MFMA GEMM / implicit-GEMM convolution for gfx950 with fused epilogues.
//
//   out[m][n] = epilogue( sum_k A[m][k] * W[n][k] )        (W in nn.Linear layout, K contiguous)
//
// Replaces (reference op sites):
//   sat ColumnParallelLinear/RowParallelLinear + bias + GELU-tanh + gated residual
//   (landiff/diffusion/dit_video_concat.py:568-629,1234-1237,1357-1370),
//   nn.Linear in the TiTok decoder (landiff/tokenizer/modules/blocks.py:164-219,253-261),
//   ContextParallelCausalConv3d / Conv2d (landiff/diffusion/vae_modules/cp_enc_dec.py:416-473,
//   590-633; landiff/diffusion/semantic_models/modules/vq_gan_blocks.py:90-148).
//
// Structure (MI355X-first, not a translation of a warp-32 tiling):
//   * 128x128 output tile per 256-thread workgroup, 4 wave64 as 2x2, each wave 64x64 =
//     2x2 v_mfma_f32_32x32x16_bf16 accumulators (64 acc VGPRs), BK = 64.
//   * A and W tiles go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, 16 B/lane, no VGPR
//     round trip), double buffered (2 x 32 KB), one barrier per K-tile.
//   * LDS image is lane-linear (DMA constraint); bank conflicts on the ds_read_b128 fragment
//     reads are removed by XOR-swizzling the 16-B chunk index with ((row>>1)&7) on the
//     *source* address and on the read (both-sides rule).
//   * Convolution is the same kernel with a different A-row address generator: the input is a
//     zero-bordered channels-last tensor [T+kT-1][H+kH-1][W+kW-1][Cin], so every tap is a plain
//     128-byte row read: no bounds checks, coalesced (B,T,H,W,C) loads.
//   * Epilogue goes through wave-private LDS so that global stores / residual loads are
//     16-byte, row-contiguous.
//   * blockIdx is remapped so that consecutive tiles of one A panel sit on one XCD (private L2).
#include "ld_common.h"
#include "../../include/landiff_hip.h"
#include <stdlib.h>
#include <type_traits>

// ld_conv_narrow.hip: 3x3x3 convolutions with <= 4 output channels; 1 = not its shape, 0 = launched (or would be: dry_run)
int ld_conv_narrow_try(const void* in_padded, const void* w, const void* bias, void* out, long ldo, long T, long H, long W, long Cin,
                       long Cout, long kT, long kH, long kW, bool plain_bias_epilogue, hipStream_t stream, bool dry_run);

namespace {

constexpr int BK = 64;
typedef int i32x8_t __attribute__((ext_vector_type(8)));      // operand of the f8f6f4 MFMAs
constexpr int CW_STRIDE = 68;                    // fp32 row stride of the epilogue staging tile (64 cols + pad)

struct GemmParams {
  const bf16_t* A;
  const bf16_t* W;
  void* out;
  const bf16_t* bias;
  const bf16_t* mul;
  const void* resid;
  const bf16_t* gate;
  const bf16_t* add2;
  int M, N, K;
  long lda, ldo, ldr, ldmul, ldadd;
  int act;
  int out_f32, resid_f32;
  int rows_per_batch, text_len;
  long gate_bstride, gate_off_img, gate_off_txt;
  // conv (channels-last, zero-bordered input)
  int H, W_, Hp, Wp, Cin, kH, kW;   // output H,W; padded input Hp,Wp
  int group_m;                      // tile-raster group height (L2 locality)
  int m_begin;                      // first output row of this launch (rows stay absolute: M is the end row)
  // fp8 (e4m3) operands: A and W are byte matrices (lda in bytes), dequantised by per-row / per-output-channel scales
  const float* scale_a;             // [M]
  const float* scale_w;             // [N]
  // MXFP8 form: one E8M0 scale byte per 32 consecutive K elements, stored K-tile-major [K / 128][rows][4] so that the 256
  // rows of a tile and K-tile are 1 KB contiguous (a [rows][K / 32] strip cost one cache line per row and K-tile)
  const unsigned char* mx_a;
  const unsigned char* mx_w;
  unsigned char* mx_out;            // non-null: the output itself is MXFP8 (out = e4m3 bytes, ldo in bytes; scales here)
  long ld_mx_out;
  // fused qkv head split (EPI_QKV): N = 3 * heads * 64 columns [q | k | v]; out is unused
  bf16_t* q_out; bf16_t* k_out; bf16_t* vt_out;       // Q, K [B][heads][Npad][64], V^T [B][heads][64][Npad]
  const bf16_t* qn_w; const bf16_t* qn_b; const bf16_t* kn_w; const bf16_t* kn_b;   // QK-LayerNorm(64) weights
  int heads, Ntok, Npad;
  float qk_eps;
  // 8-phase kernels: the launch covers tiles [tile_begin, tile_end) of the 256 x 256 raster (tile_end == 0: all of them).  The
  // whole rounds of the chip go to ld_gemm8p_kernel, the partial last round to ld_gemm8p_n128_kernel as 256 x 128 half tiles.
  int tile_begin, tile_end;
  // convolutions only: GroupNorm partial statistics of the bf16 output, [ceil(M / 64)][N / 4][2] fp32 = (sum, sum of squares) of
  // every 64-row x 4-channel patch, written by the epilogue that holds the values anyway (ld_conv_cl_bf16_gn); null: none
  float* gn_part;
};

__device__ __forceinline__ void glds16(const bf16_t* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void*)g,
      (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// epilogue on 8 consecutive columns of one output row
__device__ __forceinline__ void epilogue_store8(const GemmParams& p, float (&v)[8], int gm, int gn0, bool vec_ok) {
  const int nvalid = (p.N - gn0) < 8 ? (p.N - gn0) : 8;
  const bf16_t* gate_row = nullptr;
  if (p.gate) {
    const int b = gm / p.rows_per_batch;
    const int rin = gm - b * p.rows_per_batch;
    gate_row = p.gate + b * p.gate_bstride + (rin < p.text_len ? p.gate_off_txt : p.gate_off_img);
  }
  if (vec_ok) {
    float bias[8], mulv[8], gt[8], rs[8], ad[8];
    if (p.bias) {
      const u32x4_t bw = *(const u32x4_t*)(p.bias + gn0);
#pragma unroll
      for (int e = 0; e < 4; ++e) { bias[2 * e] = bf_lo(bw[e]); bias[2 * e + 1] = bf_hi(bw[e]); }
    }
    if (p.mul) {
      const u32x4_t mw = *(const u32x4_t*)(p.mul + (long)gm * p.ldmul + gn0);
#pragma unroll
      for (int e = 0; e < 4; ++e) { mulv[2 * e] = bf_lo(mw[e]); mulv[2 * e + 1] = bf_hi(mw[e]); }
    }
    if (gate_row) {
      const u32x4_t gw = *(const u32x4_t*)(gate_row + gn0);
#pragma unroll
      for (int e = 0; e < 4; ++e) { gt[2 * e] = bf_lo(gw[e]); gt[2 * e + 1] = bf_hi(gw[e]); }
    }
    if (p.resid) {
      if (p.resid_f32) {
        const f32x4_t r0 = *(const f32x4_t*)((const float*)p.resid + (long)gm * p.ldr + gn0);
        const f32x4_t r1 = *(const f32x4_t*)((const float*)p.resid + (long)gm * p.ldr + gn0 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { rs[e] = r0[e]; rs[4 + e] = r1[e]; }
      } else {
        const u32x4_t rw = *(const u32x4_t*)((const bf16_t*)p.resid + (long)gm * p.ldr + gn0);
#pragma unroll
        for (int e = 0; e < 4; ++e) { rs[2 * e] = bf_lo(rw[e]); rs[2 * e + 1] = bf_hi(rw[e]); }
      }
    }
    if (p.add2) {
      const u32x4_t aw = *(const u32x4_t*)(p.add2 + (long)gm * p.ldadd + gn0);
#pragma unroll
      for (int e = 0; e < 4; ++e) { ad[2 * e] = bf_lo(aw[e]); ad[2 * e + 1] = bf_hi(aw[e]); }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = v[e];
      if (p.bias) x += bias[e];
      x = rbf(x);                                   // bf16 Linear/conv output
      if (p.act) x = rbf(apply_act(p.act, x));
      if (p.mul) x = rbf(x * mulv[e]);
      if (gate_row) x = rbf(x * gt[e]);
      if (p.resid) { x = rs[e] + x; if (!p.out_f32) x = rbf(x); }
      if (p.add2) { x = x + ad[e]; if (!p.out_f32) x = rbf(x); }
      v[e] = x;
    }
    if (p.out_f32) {
      float* o = (float*)p.out + (long)gm * p.ldo + gn0;
      *(f32x4_t*)o = (f32x4_t){v[0], v[1], v[2], v[3]};
      *(f32x4_t*)(o + 4) = (f32x4_t){v[4], v[5], v[6], v[7]};
    } else {
      u32x4_t ow;
#pragma unroll
      for (int e = 0; e < 4; ++e) ow[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
      *(u32x4_t*)((bf16_t*)p.out + (long)gm * p.ldo + gn0) = ow;
    }
  } else {
    for (int e = 0; e < nvalid; ++e) {
      const int gn = gn0 + e;
      float x = v[e];
      if (p.bias) x += bf2f(p.bias[gn]);
      x = rbf(x);
      if (p.act) x = rbf(apply_act(p.act, x));
      if (p.mul) x = rbf(x * bf2f(p.mul[(long)gm * p.ldmul + gn]));
      if (gate_row) x = rbf(x * bf2f(gate_row[gn]));
      if (p.resid) {
        const float r = p.resid_f32 ? ((const float*)p.resid)[(long)gm * p.ldr + gn]
                                    : bf2f(((const bf16_t*)p.resid)[(long)gm * p.ldr + gn]);
        x = r + x; if (!p.out_f32) x = rbf(x);
      }
      if (p.add2) { x = x + bf2f(p.add2[(long)gm * p.ldadd + gn]); if (!p.out_f32) x = rbf(x); }
      if (p.out_f32) ((float*)p.out)[(long)gm * p.ldo + gn] = x;
      else ((bf16_t*)p.out)[(long)gm * p.ldo + gn] = f2bf(x);
    }
  }
}

// Epilogue shared by both main loops: per MFMA row-block, accumulators -> wave-private LDS (fp32) -> row-contiguous
// 16-byte stores (wave tile = MI x NI MFMA 32x32 tiles, NI * 32 == 64 columns).  Must be entered with all main-loop
// LDS traffic of the whole workgroup retired (a barrier); inside, every wave works on its own staging tile, so the only
// ordering needed is the in-order execution of one wave's own DS instructions -- no workgroup barriers.
//
// Code size is the constraint here: the epilogue is straight-line code that every wave walks once per tile, and a
// body that carries every runtime feature (four activations inlined per element) grew the kernel past 160 KB -- more
// than the instruction cache, so each tile paid tens of microseconds of instruction fetch.  The three epilogues of
// the DiT layer are therefore compile-time specialisations (a few KB each, fully unrolled, operands of a row-block
// requested before its accumulators are staged); everything else takes the compact generic path.
enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_GATE = 2, EPI_GENERIC = 3, EPI_GELU_MX = 4, EPI_QKV = 5 };   // 4: bias + GELU, MXFP8 output; 5: qkv head split

// stage_block(ic) writes the 32 x 64 fp32 values of 32-row block ic of the wave tile into cw[32][CW_STRIDE] -- the only part
// that depends on the MFMA shape the accumulators came from (gemm_epilogue: 32x32x16, gemm_epilogue16: 16x16x32).
// hook(): called once, right after the epilogue's FIRST global loads have been issued (bias; gate / residual / control add of
// the first row block) and before anything waits on them.  The persistent 8-phase kernel issues the next tile's first K-tile
// there: LDS-DMA and loads retire in order, so anything the epilogue loads after that would wait for the DMA to land.
struct NoHook { __device__ __forceinline__ void operator()() const {} };
//
// GN (the convolution kernels): with p.gn_part set, every lane also sums the FINAL bf16 values it stores -- 8 consecutive channels
// of 4 rows per 32-row block -- as two 4-channel quads (sum, sum of squares), and after every second row block the 8 lanes that
// hold the same columns meet in a fixed butterfly and lane 0..7 store the 64-row patch's four numbers.  Each (64-row unit, quad)
// is written exactly once per launch, by a fixed sequence of fp32 additions: deterministic; ld_gn_stats_from_partials_kernel
// (ld_norm.hip) sums the units in double in index order.  Replaces the separate read of the whole activation by
// ld_gn_stats_kernel for the VAE's GroupNorms, all of which normalise a convolution's output (cp_enc_dec.py:546-569, 745-782).
template <int MI, int EPI, typename StageFn, typename Hook = NoHook, bool GN = false>
__device__ __forceinline__ void gemm_epilogue_core(const GemmParams& p, StageFn&& stage_block, float* cw, int lane,
                                                   int row0, int col0w, Hook&& hook = Hook{}) {
  const int col0 = (lane & 7) * 8;
  const int gn0 = col0w + col0;
  static_assert(!GN || (MI % 2 == 0 && (EPI == EPI_BIAS || EPI == EPI_GENERIC)), "GroupNorm partials: 64-row units, conv epilogues");
  float gq[4] = {0.f, 0.f, 0.f, 0.f};                       // quad 0 (sum, sumsq), quad 1 (sum, sumsq)
  auto gn_add = [&](const float (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { gq[0] += v[e]; gq[1] += v[e] * v[e]; }
#pragma unroll
    for (int e = 4; e < 8; ++e) { gq[2] += v[e]; gq[3] += v[e] * v[e]; }
  };
  auto gn_flush = [&](int unit_row0) {                       // all 64 lanes get here together
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float a = gq[q];
      a += __shfl_xor(a, 8, 64); a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
      gq[q] = a;
    }
    if (lane < 8 && gn0 < p.N && unit_row0 < p.M)
      *(f32x4_t*)(p.gn_part + ((long)(unit_row0 >> 6) * (p.N >> 2) + (gn0 >> 2)) * 2) = (f32x4_t){gq[0], gq[1], gq[2], gq[3]};
    gq[0] = gq[1] = gq[2] = gq[3] = 0.f;
  };
  if constexpr (EPI == EPI_GENERIC) {
    hook();
    const bool vec_ok = ((p.N & 7) == 0) && ((p.ldo & 7) == 0) &&
                        (p.resid == nullptr || (p.ldr & 7) == 0) &&
                        (p.mul == nullptr || (p.ldmul & 7) == 0) &&
                        (p.add2 == nullptr || (p.ldadd & 7) == 0);
    auto row_block = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      stage_block(ic);
#pragma unroll 1
      for (int ps = 0; ps < 4; ++ps) {
        const int row = ps * 8 + (lane >> 3);
        const int gm = row0 + i * 32 + row;
        if (gm < p.M && gn0 < p.N) {
          float v[8];
          const f32x4_t lo = *(const f32x4_t*)(cw + row * CW_STRIDE + col0);
          const f32x4_t hi = *(const f32x4_t*)(cw + row * CW_STRIDE + col0 + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
          epilogue_store8(p, v, gm, gn0, vec_ok);
          if constexpr (GN) if (p.gn_part) gn_add(v);       // vec_ok (the launcher checks): v holds the stored, rounded values
        }
      }
      if constexpr (GN && (i & 1)) if (p.gn_part) gn_flush(row0 + (i - 1) * 32);
    };
    row_block(std::integral_constant<int, 0>{});
    if constexpr (MI > 1) row_block(std::integral_constant<int, 1>{});
    if constexpr (MI > 2) row_block(std::integral_constant<int, 2>{});
    if constexpr (MI > 3) row_block(std::integral_constant<int, 3>{});
  } else {
    // specialised: N % 8 == 0, all leading dimensions % 8 == 0, bf16 output (checked by the launcher)
    const bool col_ok = gn0 < p.N;
    float bias[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[e] = 0.f;
    if (p.bias && col_ok) {
      const u32x4_t bw = *(const u32x4_t*)(p.bias + gn0);
#pragma unroll
      for (int e = 0; e < 4; ++e) { bias[2 * e] = bf_lo(bw[e]); bias[2 * e + 1] = bf_hi(bw[e]); }
    }
    // gate row selection without a division per row: the tile's first batch and the next batch boundary
    int bnd = 0, b0 = 0;
    if constexpr (EPI == EPI_GATE) {
      b0 = row0 / p.rows_per_batch;
      bnd = (b0 + 1) * p.rows_per_batch;
    }
    const int rsub = lane >> 3;
    auto row_block = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      u32x4_t g[4], rs[4], ad[4];
      if constexpr (EPI == EPI_GATE) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          const int gm = row0 + i * 32 + ps * 8 + rsub;
          g[ps] = rs[ps] = ad[ps] = (u32x4_t){0u, 0u, 0u, 0u};
          if (gm < p.M && col_ok) {
            const int b = gm >= bnd ? b0 + 1 : b0;
            const int rin = gm - b * p.rows_per_batch;
            g[ps] = *(const u32x4_t*)(p.gate + b * p.gate_bstride + (rin < p.text_len ? p.gate_off_txt : p.gate_off_img) + gn0);
            rs[ps] = *(const u32x4_t*)((const bf16_t*)p.resid + (long)gm * p.ldr + gn0);
            if (p.add2) ad[ps] = *(const u32x4_t*)(p.add2 + (long)gm * p.ldadd + gn0);
          }
        }
      }
      if constexpr (i == 0) hook();
      stage_block(ic);
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const int row = ps * 8 + rsub;
        const int gm = row0 + i * 32 + row;
        const f32x4_t lo = *(const f32x4_t*)(cw + row * CW_STRIDE + col0);
        const f32x4_t hi = *(const f32x4_t*)(cw + row * CW_STRIDE + col0 + 4);
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {                       // element pairs: one cvt_pk per bf16 rounding of two values
          ld_f32x2_t x = rbf2((ld_f32x2_t){v[2 * e], v[2 * e + 1]} + (ld_f32x2_t){bias[2 * e], bias[2 * e + 1]});      // bf16 Linear output
          if constexpr (EPI == EPI_GELU || EPI == EPI_GELU_MX) x = act_gelu_tanh2(x);
          if constexpr (EPI == EPI_GATE) {
            x = rbf2(x * unpack_bf16x2(g[ps][e]));
            x = unpack_bf16x2(rs[ps][e]) + x;               // rounded by the pack below (or here, when another term follows)
            if (p.add2) x = rbf2(x) + unpack_bf16x2(ad[ps][e]);
          }
          v[2 * e] = x[0]; v[2 * e + 1] = x[1];
        }
        if constexpr (EPI == EPI_GELU_MX) {
          // the bf16 activation, quantised where it is produced: a 32-column MX block is the 8 columns of four adjacent
          // lanes of the same row (lane bits 0-1); scale = smallest power of two >= amax / 448 (ld_quant_mxfp8_kernel)
          float amax = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) {       // round to bf16 pairwise (one cvt_pk + two unpacks per pair)
            const uint32_t pk = pack_bf16x2(v[2 * e], v[2 * e + 1]);
            v[2 * e] = bf_lo(pk); v[2 * e + 1] = bf_hi(pk);
            amax = fmaxf(amax, fmaxf(fabsf(v[2 * e]), fabsf(v[2 * e + 1])));
          }
          amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
          amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
          const uint32_t tb = __float_as_uint(amax * (1.0f / 448.0f));
          int sb = (int)((tb >> 23) & 0xffu) + ((tb & 0x7fffffu) != 0u ? 1 : 0);
          sb = amax > 0.f ? (sb < 1 ? 1 : (sb > 254 ? 254 : sb)) : 0;
          const float inv = __uint_as_float((uint32_t)(254 - sb) << 23);      // exact power of two: |v| * inv <= 448, no clamp
          if (gm < p.M && col_ok) {
            u32x2_t o;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              unsigned w = 0;
              w = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * h] * inv, v[4 * h + 1] * inv, w, false);
              w = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * h + 2] * inv, v[4 * h + 3] * inv, w, true);
              o[h] = w;
            }
            *(u32x2_t*)((unsigned char*)p.out + (long)gm * p.ldo + gn0) = o;
            if ((lane & 3) == 0) p.mx_out[(((long)(gn0 >> 7)) * p.ld_mx_out + gm) * 4 + ((gn0 >> 5) & 3)] = (unsigned char)sb;
          }
        } else if (gm < p.M && col_ok) {
          u32x4_t ow;
#pragma unroll
          for (int e = 0; e < 4; ++e) ow[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
          __builtin_nontemporal_store(ow, (u32x4_t*)((bf16_t*)p.out + (long)gm * p.ldo + gn0));
          if constexpr (GN) if (p.gn_part) gn_add(v);       // EPI_BIAS: v = rbf2(acc + bias), already the stored values
        }
      }
      if constexpr (GN && (i & 1)) if (p.gn_part) gn_flush(row0 + (i - 1) * 32);
    };
    // explicit expansion: a `#pragma unroll` over a body this large is silently dropped and acc[] lands in scratch
    row_block(std::integral_constant<int, 0>{});
    if constexpr (MI > 1) row_block(std::integral_constant<int, 1>{});
    if constexpr (MI > 2) row_block(std::integral_constant<int, 2>{});
    if constexpr (MI > 3) row_block(std::integral_constant<int, 3>{});
  }
  static_assert(MI <= 4, "extend the expansion");
}

template <int MI, int NI, int EPI, bool GN = false>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x16_t (&acc)[MI][NI], char* smem, int wave, int lane,
                                              int row0, int col0w) {
  float* cw = (float*)smem + wave * (32 * CW_STRIDE);
  auto stage_block = [&](auto ic) {
    constexpr int i = decltype(ic)::value;
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        cw[row * CW_STRIDE + j * 32 + (lane & 31)] = acc[i][j][r];
      }
  };
  gemm_epilogue_core<MI, EPI, decltype(stage_block)&, NoHook, GN>(p, stage_block, cw, lane, row0, col0w);
}

// Accumulators of v_mfma_f32_16x16x32_bf16: acc[i][j][r] = C[i * 16 + (lane >> 4) * 4 + r][j * 16 + (lane & 15)], a wave tile of
// (MI * 32) rows x 64 columns = [2 * MI][4] blocks starting at column block j0.
// SWAP: the accumulators came from MFMAs with the operands exchanged (W fragment first), i.e. blocks of C^T:
//   acc[i][j][r] = C[i * 16 + (lane & 15)][j * 16 + (lane >> 4) * 4 + r]
// -- a lane's four registers are four consecutive COLUMNS of one row, so staging a block is ONE ds_write_b128 per lane instead of
// four ds_write_b32 (128 -> 32 LDS store instructions per wave tile; conflict-free: the 8 lanes of a store group are 8 rows,
// 68 dwords apart).  Same dot products, same results.
template <int MI, int EPI, int NJ, bool SWAP = false, typename Hook = NoHook, bool GN = false>
__device__ __forceinline__ void gemm_epilogue16(const GemmParams& p, f32x4_t (&acc)[2 * MI][NJ], int j0, char* smem, int wave,
                                                int lane, int row0, int col0w, Hook&& hook = Hook{}) {
  float* cw = (float*)smem + wave * (32 * CW_STRIDE);
  auto stage_block = [&](auto ic) {
    constexpr int i = decltype(ic)::value;
#pragma unroll
    for (int di = 0; di < 2; ++di)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (SWAP) {
          *(f32x4_t*)(cw + (di * 16 + (lane & 15)) * CW_STRIDE + j * 16 + (lane >> 4) * 4) = acc[2 * i + di][j0 + j];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            cw[(di * 16 + (lane >> 4) * 4 + r) * CW_STRIDE + j * 16 + (lane & 15)] = acc[2 * i + di][j0 + j][r];
        }
      }
  };
  gemm_epilogue_core<MI, EPI, decltype(stage_block)&, Hook, GN>(p, stage_block, cw, lane, row0, col0w, static_cast<Hook&&>(hook));
}

// ------------------------------------------------------------------------------------------------
// EPI_QKV: the DiT's qkv Linear with the head split fused into its epilogue (16x16x32 accumulators only).  Replaces the
// Linear output + sat's _transpose_for_scores + query/key_layernorm of AdaLNMixin.attention_fn
// (landiff/diffusion/dit_video_concat.py:636-653) -- i.e. ld_gemm_bf16 followed by ld_qkv_split mode 0 -- without the
// [M][3*heads*64] round trip through HBM.  A wave's 64 output columns are exactly one head of q, k or v:
//   q / k:  32-row blocks through the fp32 staging tile; the 8 lanes that hold a row's 64 columns do LayerNorm(64) on the
//           bf16-rounded Linear output (same operation order as ld_qkv_split_kernel) and store the 128-byte row of
//           Q / K [B][heads][Npad][64];
//   v:      the accumulators go (bias added, rounded) straight into a TRANSPOSED bf16 tile [64 d][rows] in LDS -- a lane's four
//           accumulator registers are four consecutive rows of one column, one ds_write_b64 -- and leave as 16-byte chunks
//           of V^T [B][heads][64][Npad] rows, 8 tokens each (batch boundary and M are multiples of 8 rows).
// Rows [Ntok, Npad) of Q / K / V^T are never written: the caller zero-fills those workspaces once.
// LDS: QKV_REGION bytes per wave (wave-private: only the in-order execution of a wave's own DS instructions orders it).
constexpr int QKV_REGION = 9216;       // >= 32 * CW_STRIDE * 4 (q/k staging) and 64 * (64 * 2 + 16) (v tile: 64 rows of a wave tile at a time)
template <int MI, typename Hook = NoHook>
__device__ __forceinline__ void qkv_epilogue16(const GemmParams& p, f32x4_t (&acc)[2 * MI][4], char* smem, int wave, int lane,
                                               int row0, int col0w, Hook&& hook = Hook{}) {
  if (col0w >= p.N) { hook(); return; }
  const int head = col0w >> 6;
  const int which = head / p.heads, h = head - which * p.heads;       // 0 = q, 1 = k, 2 = v
  char* reg = smem + wave * QKV_REGION;
  const int b0 = row0 / p.Ntok;
  const int bnd = (b0 + 1) * p.Ntok;           // a wave tile (<= 128 rows, Ntok >= 256) crosses at most one batch boundary
  if (which < 2) {
    float* cw = (float*)reg;
    const int sub = lane & 7, rsub = lane >> 3;
    const bf16_t* nw = which ? p.kn_w : p.qn_w;
    const bf16_t* nb = which ? p.kn_b : p.qn_b;
    bf16_t* dst = which ? p.k_out : p.q_out;
    float bias[8], wv[8], bv[8];
    {
      const u32x4_t bw = *(const u32x4_t*)(p.bias + col0w + sub * 8), ww = *(const u32x4_t*)(nw + sub * 8), nbw = *(const u32x4_t*)(nb + sub * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bias[2 * e] = bf_lo(bw[e]); bias[2 * e + 1] = bf_hi(bw[e]);
        wv[2 * e] = bf_lo(ww[e]); wv[2 * e + 1] = bf_hi(ww[e]);
        bv[2 * e] = bf_lo(nbw[e]); bv[2 * e + 1] = bf_hi(nbw[e]);
      }
    }
    hook();
    auto row_block = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
#pragma unroll
      for (int di = 0; di < 2; ++di)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            cw[(di * 16 + (lane >> 4) * 4 + r) * CW_STRIDE + j * 16 + (lane & 15)] = acc[2 * i + di][j][r];
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const int row = ps * 8 + rsub;
        const int gm = row0 + i * 32 + row;
        const f32x4_t lo = *(const f32x4_t*)(cw + row * CW_STRIDE + sub * 8);
        const f32x4_t hi = *(const f32x4_t*)(cw + row * CW_STRIDE + sub * 8 + 4);
        float v[8];
#pragma unroll
        for (int e = 0; e < 2; ++e) {                    // the bf16 Linear output, rounded pairwise
          const ld_f32x2_t a = rbf2((ld_f32x2_t){lo[2 * e], lo[2 * e + 1]} + (ld_f32x2_t){bias[2 * e], bias[2 * e + 1]});
          const ld_f32x2_t c = rbf2((ld_f32x2_t){hi[2 * e], hi[2 * e + 1]} + (ld_f32x2_t){bias[4 + 2 * e], bias[5 + 2 * e]});
          v[2 * e] = a[0]; v[2 * e + 1] = a[1]; v[4 + 2 * e] = c[0]; v[5 + 2 * e] = c[1];
        }
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[e];
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        const float mean = s * (1.0f / 64.0f);
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[e] - mean; ss += d * d; }
        ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 4, 64);
        const float rstd = rsqrtf(ss * (1.0f / 64.0f) + p.qk_eps);
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const ld_f32x2_t m2 = {mean, mean}, r2 = {rstd, rstd};
          o[e] = pack_bf16x2(((ld_f32x2_t){v[2 * e], v[2 * e + 1]} - m2) * r2 * (ld_f32x2_t){wv[2 * e], wv[2 * e + 1]} + (ld_f32x2_t){bv[2 * e], bv[2 * e + 1]});
        }
        if (gm < p.M) {
          const int b = gm >= bnd ? b0 + 1 : b0;
          const int n = gm - b * p.Ntok;
          __builtin_nontemporal_store(o, (u32x4_t*)(dst + (((long)b * p.heads + h) * p.Npad + n) * 64 + sub * 8));
        }
      }
    };
    row_block(std::integral_constant<int, 0>{});
    if constexpr (MI > 1) row_block(std::integral_constant<int, 1>{});
    if constexpr (MI > 2) row_block(std::integral_constant<int, 2>{});
    if constexpr (MI > 3) row_block(std::integral_constant<int, 3>{});
  } else {
    // (round 6) a 128-row wave tile goes through the transposed tile in two 64-row halves: 9 KB instead of 17 KB per wave, so that
    // the whole epilogue staging (8 x QKV_REGION) stays clear of K-tile buffer 0 and the persistent kernel can request the next
    // tile's first K-tile from inside this epilogue too (PREFETCH in ld_gemm8p_kernel).  Wave-private LDS: the second half's
    // stores follow the first half's loads in the wave's own DS queue, which executes in order.
    constexpr int VH = MI >= 4 ? 2 : 1;          // halves
    constexpr int MH = MI / VH;                  // 32-row blocks per half
    constexpr int ROWB = MH * 64 + 16;           // bytes per d row of the transposed tile (MH * 32 rows + pad, 16-byte aligned)
    static_assert(MI % VH == 0 && 64 * ROWB <= QKV_REGION, "v tile does not fit its LDS region");
    float bj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bj[j] = bf2f(p.bias[col0w + j * 16 + (lane & 15)]);
    hook();
#pragma unroll
    for (int vh = 0; vh < VH; ++vh) {
#pragma unroll
      for (int i = 0; i < 2 * MH; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          u32x2_t w2;
          w2[0] = pack_bf16x2(acc[vh * 2 * MH + i][j][0] + bj[j], acc[vh * 2 * MH + i][j][1] + bj[j]);
          w2[1] = pack_bf16x2(acc[vh * 2 * MH + i][j][2] + bj[j], acc[vh * 2 * MH + i][j][3] + bj[j]);
          *(u32x2_t*)(reg + (j * 16 + (lane & 15)) * ROWB + (i * 16 + (lane >> 4) * 4) * 2) = w2;
        }
      constexpr int CPR = MH * 4;                // 16-byte chunks (8 rows) per d row
#pragma unroll
      for (int it = 0; it < CPR; ++it) {         // 64 * CPR chunks, 64 per trip
        const int id = it * 64 + lane;
        const int d = id / CPR, c = id - d * CPR;
        const int gm = row0 + vh * MH * 32 + c * 8;
        const u32x4_t val = *(const u32x4_t*)(reg + d * ROWB + c * 16);
        if (gm < p.M) {
          const int b = gm >= bnd ? b0 + 1 : b0;
          const int n = gm - b * p.Ntok;
          __builtin_nontemporal_store(val, (u32x4_t*)(p.vt_out + (((long)b * p.heads + h) * 64 + d) * p.Npad + n));
        }
      }
    }
  }
}

// which specialisation a problem may use (the generic path handles everything)
inline int pick_epilogue(const GemmParams& p) {
  if (p.q_out) return EPI_QKV;
  if (p.mx_out) return EPI_GELU_MX;      // (the launcher checked: bias + GELU-tanh only, N % 32 == 0)
  const bool aligned = ((p.N & 7) == 0) && ((p.ldo & 7) == 0) && !p.out_f32 && !p.mul;
  if (!aligned) return EPI_GENERIC;
  if (p.gate && p.resid && !p.resid_f32 && p.act == 0 && (p.ldr & 7) == 0 && (!p.add2 || (p.ldadd & 7) == 0) &&
      p.rows_per_batch >= 512)
    return EPI_GATE;
  if (p.gate || p.resid || p.add2) return EPI_GENERIC;
  if (p.act == LD_ACT_GELU_TANH) return EPI_GELU;
  if (p.act == 0) return EPI_BIAS;
  return EPI_GENERIC;
}

// Block tile BM x BN, WM x WN waves, each wave (BM/WM) x (BN/WN) = MI x NI MFMA 32x32 tiles.
// M16: the same tiles on v_mfma_f32_16x16x32_bf16 (32-deep k-steps, [2 * MI][4] accumulators of 4 registers): equal FLOPs per
// register and per LDS byte, but the 16x16x32 form draws less power per FLOP on random operands -- under the chip's power
// governor an MFMA-only loop sustains 2105 TFLOP/s on it against 1837 on 32x32x16 (tools/probe/mfma_power.hip,
// profiles/r02_mfma_power_probe.txt) -- and power, not issue slots, is what bounds these kernels.
template <int BM, int BN, int WM, int WN, int NSTAGE, bool CONV, int EPI, bool M16 = false>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN >= 16) ? 4 : 2) void ld_gemm_kernel(GemmParams p) {
  constexpr int NW = WM * WN;
  constexpr int NT = NW * 64;
  constexpr int MI = BM / WM / 32, NI = BN / WN / 32;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int A_LOADS = BM / 8 / NW, B_LOADS = BN / 8 / NW;     // 1 KB LDS-DMA pieces per wave
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile/wave mismatch");
  static_assert(BN / WN == 64, "epilogue staging assumes 64-column wave tiles");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;

  // Tile order: blockIdx -> XCD-contiguous logical id (each XCD has a private 4 MB L2) -> grouped raster: the ~64
  // tiles resident on one XCD form a GROUP_M x (64/GROUP_M) patch, so an A panel and a W panel are each re-read from
  // L2 ~8 times instead of W being re-streamed from MALL/HBM for every row of tiles.
  const int nbm = (p.M - p.m_begin + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, nbm * nbn);
  const int gm_sz = p.group_m;
  const int per_group = gm_sz * nbn;
  const int group = bid / per_group, in_group = bid - group * per_group;
  const int first_m = group * gm_sz;
  const int rows_here = (nbm - first_m) < gm_sz ? (nbm - first_m) : gm_sz;
  const int m0 = p.m_begin + (first_m + in_group % rows_here) * BM, n0 = (in_group / rows_here) * BN;

  // ---- per-thread source row offsets ----
  uint32_t offA[A_LOADS], offW[B_LOADS];    // element offsets (< 2^31 for every shape on the path)
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) {
    const int r = (wave * A_LOADS + i) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((r >> 1) & 7);   // source-side swizzle
    int gm = m0 + r; gm = gm < p.M ? gm : p.M - 1;
    if (CONV) {
      const int hw = p.H * p.W_;
      const int t = gm / hw, rem = gm - t * hw;
      const int h = rem / p.W_, w = rem - h * p.W_;
      offA[i] = (uint32_t)((((long)t * p.Hp + h) * p.Wp + w) * p.Cin + chunk * 8);
    } else {
      offA[i] = (uint32_t)((long)gm * p.lda + chunk * 8);
    }
  }
#pragma unroll
  for (int i = 0; i < B_LOADS; ++i) {
    const int r = (wave * B_LOADS + i) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((r >> 1) & 7);
    int gn = n0 + r; gn = gn < p.N ? gn : p.N - 1;
    offW[i] = (uint32_t)((long)gn * p.K + chunk * 8);
  }

  const int nk = p.K / BK;
  const int cpt = CONV ? p.Cin / BK : 1;   // K-tiles per tap

  auto stage = [&](int buf, int kt) {
    long koffA;
    if (CONV) {
      const int tap = kt / cpt, c0 = (kt - tap * cpt) * BK;
      const int khw = p.kH * p.kW;
      const int dt = tap / khw, r2 = tap - dt * khw;
      const int dh = r2 / p.kW, dw = r2 - dh * p.kW;
      koffA = (((long)dt * p.Hp + dh) * p.Wp + dw) * p.Cin + c0;
    } else {
      koffA = (long)kt * BK;
    }
    const long koffW = (long)kt * BK;
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) glds16(p.A + offA[i] + koffA, base + (wave * A_LOADS + i) * 1024);
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) glds16(p.W + offW[i] + koffW, base + A_BYTES + (wave * B_LOADS + i) * 1024);
  };

  f32x16_t acc[M16 ? 1 : MI][M16 ? 1 : NI];
  f32x4_t acc16[M16 ? 2 * MI : 1][M16 ? 4 : 1];
  if constexpr (M16) {
#pragma unroll
    for (int i = 0; i < 2 * MI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc16[i][j][r] = 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }

  // fragment read offsets: row * 128 B plus the swizzled 16-B chunk of k-step kk (rows of later MFMA tiles are
  // +32 rows = +4096 B (16x16x32: +16 rows = +2048 B) with the same swizzle key, so they fold into the ds_read immediate
  // offset).  16x16x32 operand: lane l holds row l & 15, k = (l >> 4) * 8 .. + 8 of the 32-deep step: the 16-byte chunk
  // ks * 4 + (l >> 4); with the (row >> 1) & 7 XOR the four 16-lane groups of a ds_read_b128 each cover all 64 banks.
  int rdA[4], rdB[4];
  if constexpr (M16) {
    const int ra = wr * (BM / WM) + (lane & 15), rb = wc * (BN / WN) + (lane & 15);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = ks * 4 + (lane >> 4);
      rdA[ks] = ra * 128 + ((c ^ ((ra >> 1) & 7)) << 4);
      rdB[ks] = A_BYTES + rb * 128 + ((c ^ ((rb >> 1) & 7)) << 4);
    }
    rdA[2] = rdA[3] = rdB[2] = rdB[3] = 0;
  } else {
    const int ra = wr * (BM / WM) + (lane & 31), rb = wc * (BN / WN) + (lane & 31);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int c = kk * 2 + (lane >> 5);
      rdA[kk] = ra * 128 + ((c ^ ((ra >> 1) & 7)) << 4);
      rdB[kk] = A_BYTES + rb * 128 + ((c ^ ((rb >> 1) & 7)) << 4);
    }
  }

  // One K-tile of MFMAs.  Fragments are software pipelined by hand (k-step kk+1 is requested before the MFMAs of kk)
  // and a scheduling barrier after every k-step keeps hipcc from hoisting all 4 k-steps' loads at once, which spills
  // the 128-register accumulator tile of the 256x256 configuration.
  // a wave whose 64 output columns lie entirely past N (the half-empty last tile column of the N = 1920 shapes) issues no
  // MFMAs: its accumulators stay zero and are never stored; the tile takes as long, at half the energy
  const bool wave_live = n0 + wc * (BN / WN) < p.N;
  auto compute = [&](auto bufc) {
    constexpr int OFF = decltype(bufc)::value * STAGE;
    if (!wave_live) return;
    if constexpr (M16) {
      // B fragments of both k-steps up front; A fragments single-buffered: block i's k-step-1 fragment is requested right
      // after its four k-step-0 MFMAs (2 * MI - 1 blocks of MFMAs of cover), which keeps the fragment registers at
      // (2 * MI + 8) x 4 next to the accumulators
      bf16x8_t a[2 * MI], b[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[0][j] = *(const bf16x8_t*)(smem + rdB[0] + OFF + j * 2048);
#pragma unroll
      for (int i = 0; i < 2 * MI; ++i) a[i] = *(const bf16x8_t*)(smem + rdA[0] + OFF + i * 2048);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[1][j] = *(const bf16x8_t*)(smem + rdB[1] + OFF + j * 2048);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < 2 * MI; ++i) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[ks][j], acc16[i][j], 0, 0, 0);
          if (ks == 0) {
            __builtin_amdgcn_sched_barrier(0);
            a[i] = *(const bf16x8_t*)(smem + rdA[1] + OFF + i * 2048);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      bf16x8_t a[2][MI], b[2][NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) a[0][i] = *(const bf16x8_t*)(smem + rdA[0] + OFF + i * 4096);
#pragma unroll
      for (int j = 0; j < NI; ++j) b[0][j] = *(const bf16x8_t*)(smem + rdB[0] + OFF + j * 4096);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int cur = kk & 1, nxt = cur ^ 1;
        if (kk < 3) {
#pragma unroll
          for (int i = 0; i < MI; ++i) a[nxt][i] = *(const bf16x8_t*)(smem + rdA[kk + 1] + OFF + i * 4096);
#pragma unroll
          for (int j = 0; j < NI; ++j) b[nxt][j] = *(const bf16x8_t*)(smem + rdB[kk + 1] + OFF + j * 4096);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
        if (MI * NI > 4) __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  if (NSTAGE == 2) {
    // two K-tiles per trip with compile-time buffer indices; no mid-loop exit (a `break` between the two halves makes
    // hipcc keep two copies of the 64 accumulator registers and shuffle them every trip), odd tail peeled
    stage(0, 0);
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
      __syncthreads();   // drains this wave's LDS-DMA (vmcnt(0)) and releases the other buffer
      stage(1, kt + 1);
      compute(std::integral_constant<int, 0>{});
      __syncthreads();
      if (kt + 2 < nk) stage(0, kt + 2);
      compute(std::integral_constant<int, 1>{});
    }
    if (kt < nk) {
      __syncthreads();
      compute(std::integral_constant<int, 0>{});
    }
  } else {
    // 3-deep LDS ring: the K-tile two steps ahead is requested while tile kt is consumed, and the barrier only
    // waits for tile kt (counted vmcnt: the newest tile's DMA stays in flight across the barrier; a raw s_barrier is
    // used because __syncthreads would drain vmcnt to 0).  RAW: own vmcnt + barrier; WAR: buffer (kt+2)%3 was last
    // read by compute(kt-1), which every wave has finished before passing barrier kt.
    constexpr int LPS = A_LOADS + B_LOADS;       // LDS-DMA instructions per wave per stage
    stage(0, 0);
    if (nk > 1) stage(1, 1);
    auto step = [&](auto bufc, int kt) {
      constexpr int B = decltype(bufc)::value;
      if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + 2 < nk) stage((B + 2) % 3, kt + 2);
      compute(bufc);
    };
    int kt = 0;
    for (; kt + 2 < nk; kt += 3) {
      step(std::integral_constant<int, 0>{}, kt);
      step(std::integral_constant<int, 1>{}, kt + 1);
      step(std::integral_constant<int, 2>{}, kt + 2);
    }
    if (kt < nk) step(std::integral_constant<int, 0>{}, kt);
    if (kt + 1 < nk) step(std::integral_constant<int, 1>{}, kt + 1);
  }
  __syncthreads();

  if constexpr (EPI == EPI_QKV) {
    static_assert(M16 || EPI != EPI_QKV, "the fused qkv split exists for the 16x16x32 accumulator layout only");
    if constexpr (M16) qkv_epilogue16<MI>(p, acc16, smem, wave, lane, m0 + wr * (BM / WM), n0 + wc * 64);
  } else if constexpr (M16) gemm_epilogue16<MI, EPI, 4, false, NoHook, CONV && MI % 2 == 0>(p, acc16, 0, smem, wave, lane, m0 + wr * (BM / WM), n0 + wc * 64);
  else gemm_epilogue<MI, NI, EPI, CONV && MI % 2 == 0>(p, acc, smem, wave, lane, m0 + wr * (BM / WM), n0 + wc * 64);
}

// Two 1 KB LDS-DMA pieces of a half-tile through a raw buffer descriptor (rebuilt from its scalars at every use: loop-invariant
// SGPR values for the compiler): per-lane byte offsets o0 / o1, wave-uniform K offset `ko` in an SGPR -- no vector ALU per piece.
#ifndef LD_GEMM_ABL   // timing-only builds (WRONG results): bit 0 = no LDS-DMA in the main loop, bit 1 = fragments read once per tile,
#define LD_GEMM_ABL 0 // bit 2 = every K-tile re-reads K-tiles 0 / 1 (L2 hits), bit 3 = no vmcnt waits, bit 4 = every second LDS-DMA piece only
#endif
template <int OFF>
__device__ __forceinline__ void stage_pieces(const bf16_t* base, int bytes, char* lds, uint32_t o0, uint32_t o1, int ko) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + OFF), 16, o0, ko, 0, 0);
  if (!(LD_GEMM_ABL & 16)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + OFF + 1024), 16, o1, ko, 0, 0);
}

template <int OFF>
__device__ __forceinline__ void stage_piece1(const bf16_t* base, int bytes, char* lds, uint32_t o0, int ko) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + OFF), 16, o0, ko, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// 8-phase main loop (round 3 default for the 256 x 256 tile; LD_GEMM_8P=0 selects the two-stage loop of ld_gemm_kernel): the
// 256x256x64 tile / 8 waves (2 x 4, 128 x 64 per wave) / 16x16x32 MFMAs of ld_gemm_kernel<256,256,2,4,...,M16> with the
// staging PIPELINED through the K loop instead of issued tile by tile:
//   * LDS = 2 K-tile buffers x 4 half-tile slots of 16 KB: A_h (h = 0, 1) holds, for BOTH wave rows wr, the 64 tile rows
//     wr * 128 + h * 64 .. + 64 (local row wr * 64 + r); B_g (g = 0, 1) holds, for ALL FOUR wave columns wc, the 32 tile
//     columns wc * 64 + g * 32 .. + 32 (local row wc * 32 + r).  Which tile row lands in which slot is free -- the LDS-DMA
//     source address is per lane -- and this choice makes every one of a K-tile's four phases read ONE half-tile of A and ONE
//     of W for the whole workgroup, so a slot is dead long before its K-tile is finished and can be re-staged early, while
//     a wave's output stays 128 contiguous rows x 64 contiguous columns (the epilogues, incl. the fused qkv head split, are
//     those of ld_gemm_kernel).
//   * a K-tile = 4 phases of 16 MFMAs (one 64 x 32 quadrant of the wave tile x K = 64):
//       ph0: read B_g0 (4 ds_read_b128) + A_h0 (8)   stage B_1(t+1)   MFMA (h0, g0)
//       ph1: read B_g1 (4)                           stage A_1(t+1)   MFMA (h0, g1)
//       ph2: read A_h1 (8)                           stage A_0(t+2)   MFMA (h1, g1)
//       ph3: --  (B_g0 fragments kept in registers)  stage B_0(t+2)   MFMA (h1, g0)   + the K-tile's only vmcnt wait
//     every slot is re-staged >= 2 phases after its last read (WAR) and its DMA has 1.5-2 K-tiles (~3000 cycles) to land;
//     the counted wait of ph3 leaves the two newest half-tiles (4 LDS-DMA instructions per wave) in flight and retires
//     K-tile t+1, which is read from the next phase on, one barrier later (RAW: own vmcnt + a barrier every wave has
//     passed).  Raw s_barrier throughout: __syncthreads() would drain vmcnt to zero.
//   * each phase is [fragment reads, stage] barrier [lgkmcnt(0), 16 MFMAs] barrier, and the two wave rows run ONE barrier
//     apart (wr = 1 takes an extra barrier up front, wr = 0 one at the end): the two waves that share a SIMD (wave w and
//     w + 4) alternate between the matrix segment and the LDS / DMA segment, so the matrix pipe always has a wave whose
//     operands are already in registers.
//   * staging goes through raw buffer descriptors: per-lane byte offsets fixed for the kernel, the K-tile / filter-tap offset
//     in an SGPR -- two buffer_load ... lds per half-tile and no vector ALU (the flat form cost two 64-bit adds per piece).
//   * PERSISTENT tiles: the grid is at most one workgroup per CU and a workgroup walks tiles blockIdx.x, + gridDim.x, ... of the
//     XCD-grouped raster.  The epilogue's LDS staging lives at the END of the 160 KB, clear of K-tile buffer 0, so the first
//     K-tile of the NEXT tile is requested before the epilogue starts (right after the epilogue's own first loads have been
//     issued: loads and LDS-DMA retire in order) and lands under it: a tile no longer pays workgroup launch, argument loads and
//     the first DMA round trip.  (Round 6: the fused-qkv epilogue too -- its V^T tile goes through LDS in two halves, 72 KB of staging.)
// Measured (tools/gemm_ab.py, profiles/r03_gemm_*): bit-identical outputs; see DESIGN.md section 4.
// ------------------------------------------------------------------------------------------------
constexpr int LD_LDS_TOTAL = 160 * 1024;

#ifdef LD_GEMM_TRACE   // timing builds (tools/gemm_tile_trace.py): per tile of ld_gemm8p_kernel start / end of main loop / end, XCC_ID, HW_ID
__device__ unsigned long long* g_gemm_trace = nullptr;     // [0]: record counter, then 4 words per record
__device__ int g_gemm_trace_cap = 0;
#endif

template <bool CONV, int EPI>
__global__ __launch_bounds__(512, 2) void ld_gemm8p_kernel(GemmParams p) {
  constexpr int BM = 256, BN = 256;
  constexpr int SLOT = 128 * 128, KBUF = 4 * SLOT;        // 16 KB half-tile slot (128 rows x 128 B); A0 A1 B0 B1 per K-tile
  constexpr int EPI_BYTES = (EPI == EPI_QKV) ? 8 * QKV_REGION : 8 * 32 * CW_STRIDE * 4;
  constexpr int EPI_OFF = (LD_LDS_TOTAL - EPI_BYTES) & ~15;      // epilogue staging at the end of the LDS
  // K-tile buffer 0 is free while the epilogue runs.  Not for the fused-qkv epilogue, although its staging has left buffer 0 alone
  // since round 6: measured 0.745 ms with, 0.734 ms without the early request (profiles/r06_gemm_two_phase_ab.txt; -DLD_QKV_PREFETCH: A/B build)
#ifdef LD_QKV_PREFETCH
  constexpr bool PREFETCH = EPI_OFF >= KBUF;
#else
  constexpr bool PREFETCH = EPI_OFF >= KBUF && EPI != EPI_QKV;
#endif
  constexpr bool SWAPACC = EPI != EPI_QKV;                // C^T accumulator blocks: 16-byte epilogue staging stores (gemm_epilogue16<SWAP>)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  const int nbm = (p.M - p.m_begin + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  const int ntiles = nbm * nbn;
  const int gm_sz = p.group_m;
  // virtual block id v -> tile origin (XCD-contiguous logical id -> grouped raster, as ld_gemm_kernel).  gridDim.x is a
  // multiple of 8 whenever a workgroup owns more than one tile, so v % 8 == blockIdx.x % 8: a workgroup's tiles stay on its XCD.
  auto tile_origin = [&](int v, int& m0, int& n0) {
    const int bid = xcd_remap(v, ntiles);
    const int per_group = gm_sz * nbn;
    const int group = bid / per_group, in_group = bid - group * per_group;
    const int first_m = group * gm_sz;
    const int rows_here = (nbm - first_m) < gm_sz ? (nbm - first_m) : gm_sz;
    m0 = p.m_begin + (first_m + in_group % rows_here) * BM;
    n0 = (in_group / rows_here) * BN;
  };

  // ---- LDS-DMA sources: this wave stages pieces 2 * wave + {0, 1} (8 local rows x 128 B each) of every half-tile ----
  // Raw buffer descriptors (A: based at the tile's first row, rows past M read as zeros; convolution: the whole padded input,
  // rows clamped), one 32-bit byte offset per [half][piece] in VGPRs, the K-tile (or filter tap) offset in an SGPR.
  // (The descriptors are rebuilt from their scalars at every use -- loop-invariant SGPR values for the compiler; a
  //  __amdgpu_buffer_rsrc_t object captured by nested generic lambdas does not get through the host pass.)
  const auto clip = [](long v) { return (int)(v < 0x7fffffffL ? v : 0x7fffffffL); };
  struct Src { const bf16_t* a; const bf16_t* w; int a_bytes, w_bytes; };
  auto tile_src = [&](int m0, int n0) {
    Src s;
    s.a = p.A + (CONV ? 0 : (long)m0 * p.lda);
    s.w = p.W + (long)n0 * p.K;
    s.a_bytes = CONV ? 0x7fffffff : clip(((long)(p.M - m0) * p.lda) * 2);
    s.w_bytes = clip(((long)(p.N - n0) * p.K) * 2);
    return s;
  };
  uint32_t offA[2][2], offW[2][2];                        // [half][piece] byte offsets
  auto set_offsets = [&](int m0, bool weights) {          // (A offsets depend on the tile only for a convolution)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int lr = wave * 16 + i * 8 + (lane >> 3);     // local row of the slot, 0 .. 127
      const int chunk = (lane & 7) ^ ((lr >> 1) & 7);     // source-side swizzle (the read applies the same key)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int tm = (lr >> 6) * 128 + h * 64 + (lr & 63);
        if (CONV) {
          int gm = m0 + tm; gm = gm < p.M ? gm : p.M - 1;
          const int hw = p.H * p.W_;
          const int t = gm / hw, rem = gm - t * hw;
          const int hh = rem / p.W_, w = rem - hh * p.W_;
          offA[h][i] = (uint32_t)(((((long)t * p.Hp + hh) * p.Wp + w) * p.Cin + chunk * 8) * 2);
        } else {
          offA[h][i] = (uint32_t)(((long)tm * p.lda + chunk * 8) * 2);
        }
        if (weights) {
          const int tn = (lr >> 5) * 64 + h * 32 + (lr & 31);
          offW[h][i] = (uint32_t)(((long)tn * p.K + chunk * 8) * 2);
        }
      }
    }
  };
  const int nk = p.K / BK;
  const int cpt = CONV ? p.Cin / BK : 1;
  auto koff_a = [&](int kt) -> int {                      // byte offset of K-tile kt within an A row
    if (LD_GEMM_ABL & 4) kt &= 1;                         // (timing build: every K-tile re-reads K-tiles 0 / 1 -- L2 hits only)
    if (CONV) {
      const int tap = kt / cpt, c0 = (kt - tap * cpt) * BK;
      const int khw = p.kH * p.kW;
      const int dt = tap / khw, r2 = tap - dt * khw;
      const int dh = r2 / p.kW, dw = r2 - dh * p.kW;
      return (int)(((((long)dt * p.Hp + dh) * p.Wp + dw) * p.Cin + c0) * 2);
    }
    return kt * (BK * 2);
  };
  char* const my_piece = smem + wave * 2048;              // + buffer * KBUF + slot * SLOT (+ 1024 for the second piece)
  Src src;                                                // the tile being computed
  auto stage_a = [&](const Src& s, auto bufc, auto hc, int kt) {
    constexpr int OFF = decltype(bufc)::value * KBUF + decltype(hc)::value * SLOT;
    stage_pieces<OFF>(s.a, s.a_bytes, my_piece, offA[decltype(hc)::value][0], offA[decltype(hc)::value][1], koff_a(kt));
  };
  auto stage_w = [&](const Src& s, auto bufc, auto gc, int kt) {
    constexpr int OFF = decltype(bufc)::value * KBUF + (2 + decltype(gc)::value) * SLOT;
    stage_pieces<OFF>(s.w, s.w_bytes, my_piece, offW[decltype(gc)::value][0], offW[decltype(gc)::value][1], ((LD_GEMM_ABL & 4) ? (kt & 1) : kt) * (BK * 2));
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  auto stage_ktile0 = [&](const Src& s) {
    stage_a(s, I0{}, I0{}, 0); stage_w(s, I0{}, I0{}, 0); stage_w(s, I0{}, I1{}, 0); stage_a(s, I0{}, I1{}, 0);
  };

  // fragment reads: 16x16x32 operand = row (lane & 15), 16-byte chunk ks * 4 + (lane >> 4) of the 128-byte K row; the swizzle
  // key ((row >> 1) & 7) depends on lane & 15 only (block and wave offsets are multiples of 16 rows), so the blocks of a
  // subtile are immediate offsets (+2048 B) of one address per k-step
  int rdA[2], rdB[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int c = (ks * 4 + (lane >> 4)) ^ (((lane & 15) >> 1) & 7);
    rdA[ks] = (wr * 64 + (lane & 15)) * 128 + (c << 4);
    rdB[ks] = (wc * 32 + (lane & 15)) * 128 + (c << 4);
  }
  f32x4_t acc[8][4];
  bf16x8_t a[4][2], b0[2][2], b1[2][2];
  auto read_a = [&](auto bufc, auto hc) {
    constexpr int OFF = decltype(bufc)::value * KBUF + decltype(hc)::value * SLOT;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) a[i][ks] = *(const bf16x8_t*)(smem + rdA[ks] + OFF + i * 2048);
  };
  auto read_b = [&](auto bufc, auto gc, bf16x8_t (&b)[2][2]) {
    constexpr int OFF = decltype(bufc)::value * KBUF + (2 + decltype(gc)::value) * SLOT;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) b[j][ks] = *(const bf16x8_t*)(smem + rdB[ks] + OFF + j * 2048);
  };
  bool wave_live = true;                                  // (a wave whose 64 columns lie past N issues no MFMAs)
  auto mma = [&](auto hc, auto gc, bf16x8_t (&b)[2][2]) {
    constexpr int H = decltype(hc)::value, G = decltype(gc)::value;
    // lgkmcnt(0) as the BUILTIN (simm16 0xC07F = vmcnt 63, expcnt 7, lgkmcnt 0): hipcc's own wait-count bookkeeping sees it.  As
    // inline asm it is invisible to that pass, which then re-waits before the next phase's fragment reads on the path that
    // skips the MFMAs (a pending ds_read into a register it is about to reuse) -- serialising the B and A reads of ph0.
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
    if (wave_live) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[H * 4 + i][G * 2 + j] = SWAPACC ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][ks], a[i][ks], acc[H * 4 + i][G * 2 + j], 0, 0, 0)
                                                : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][ks], b[j][ks], acc[H * 4 + i][G * 2 + j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto bar = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
#ifndef LD_GEMM_PH4   // round 6: TWO phases of 32 MFMAs per K-tile (-DLD_GEMM_PH4: the four phases of 16 of rounds 3-5, for A/B builds)
  // Half the barriers and half the role switches between the two waves of a SIMD per K-tile; the same MFMAs on the same accumulators
  // in the same order -> the same bits.  Measured -2.6 % on the four DiT GEMMs (profiles/r06_gemm_two_phase_ab.txt).
  auto mma2 = [&](auto hc, auto g0c, bf16x8_t (&bA)[2][2], auto g1c, bf16x8_t (&bB)[2][2]) {
    constexpr int H = decltype(hc)::value, G0 = decltype(g0c)::value, G1 = decltype(g1c)::value;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
    if (wave_live) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[H * 4 + i][G0 * 2 + j] = SWAPACC ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(bA[j][ks], a[i][ks], acc[H * 4 + i][G0 * 2 + j], 0, 0, 0)
                                                 : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][ks], bA[j][ks], acc[H * 4 + i][G0 * 2 + j], 0, 0, 0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[H * 4 + i][G1 * 2 + j] = SWAPACC ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(bB[j][ks], a[i][ks], acc[H * 4 + i][G1 * 2 + j], 0, 0, 0)
                                                 : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][ks], bB[j][ks], acc[H * 4 + i][G1 * 2 + j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // P0: read B_g0, B_g1, A_h0 (16 fragments); stage B_1 / A_1 of K-tile kt + 1; MFMA (h0, g0), (h0, g1)
  // P1: read A_h1 (8);                        stage A_0 / B_0 of K-tile kt + 2; MFMA (h1, g1), (h1, g0)
  // A slot is re-staged as early as ONE phase after its last read, so every wave retires its fragment reads (lgkmcnt 0) BEFORE
  // the first barrier of the reading phase: a wave that has passed the barrier ending that phase knows that every wave of both
  // rows holds its fragments in registers.
  auto ktile = [&](auto bufc, int kt) {
    constexpr int B = decltype(bufc)::value;
    using Bc = std::integral_constant<int, B>;
    using Nc = std::integral_constant<int, B ^ 1>;
    // P0.  LDS-DMA in flight on entry (oldest first): A_1(kt) [2], A_0 / B_0(kt + 1) [4]
    if (!(LD_GEMM_ABL & 2) || kt == 0) {
      read_b(Bc{}, I0{}, b0);
      read_b(Bc{}, I1{}, b1);
      __builtin_amdgcn_sched_barrier(0);
      read_a(Bc{}, I0{});
    }
    if (kt + 1 < nk) {
      if (!(LD_GEMM_ABL & 1)) { stage_w(src, Nc{}, I1{}, kt + 1); stage_a(src, Nc{}, I1{}, kt + 1); }
      if (LD_GEMM_ABL & 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");   // A_1(kt) has landed: read in P1, one barrier later
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    bar(); mma2(I0{}, I0{}, b0, I1{}, b1); bar();
    // P1.  In flight: A_0 / B_0(kt + 1) [4], B_1 / A_1(kt + 1) [4]
    if (!(LD_GEMM_ABL & 2)) read_a(Bc{}, I1{});
    if (kt + 2 < nk) {
      if (!(LD_GEMM_ABL & 1)) { stage_a(src, Bc{}, I0{}, kt + 2); stage_w(src, Bc{}, I0{}, kt + 2); }
      if (LD_GEMM_ABL & 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");   // A_0 / B_0 / B_1 of K-tile kt + 1 have landed
    } else if (kt + 1 < nk) {
      asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");   // nothing new was issued: only A_1(kt + 1) may stay in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    bar(); mma2(I1{}, I1{}, b1, I0{}, b0); bar();
  };
#else
  auto ktile = [&](auto bufc, int kt) {
    constexpr int B = decltype(bufc)::value;
    using Bc = std::integral_constant<int, B>;
    using Nc = std::integral_constant<int, B ^ 1>;
    // ph0
    read_b(Bc{}, I0{}, b0);
    __builtin_amdgcn_sched_barrier(0);
    read_a(Bc{}, I0{});
    if (kt + 1 < nk) stage_w(src, Nc{}, I1{}, kt + 1);
    bar(); mma(I0{}, I0{}, b0); bar();
    // ph1
    read_b(Bc{}, I1{}, b1);
    if (kt + 1 < nk) stage_a(src, Nc{}, I1{}, kt + 1);
    bar(); mma(I0{}, I1{}, b1); bar();
    // ph2
    read_a(Bc{}, I1{});
    if (kt + 2 < nk) stage_a(src, Bc{}, I0{}, kt + 2);
    bar(); mma(I1{}, I1{}, b1); bar();
    // ph3
    if (kt + 2 < nk) {
      stage_w(src, Bc{}, I0{}, kt + 2);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");    // K-tile kt + 1 has landed; A_0 / B_0 of kt + 2 stay in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    bar(); mma(I1{}, I0{}, b0); bar();
  };
#endif

  set_offsets(0, true);
  bool k0_staged = false;                                 // K-tile 0 of the tile about to start is already on its way
  const int v_end = p.tile_end > 0 ? p.tile_end : ntiles; // (the tiles behind it: ld_gemm8p_n128_kernel)
  for (int v = p.tile_begin + blockIdx.x; v < v_end; v += gridDim.x) {
    int m0, n0;
    tile_origin(v, m0, n0);
#ifdef LD_GEMM_TRACE
    const unsigned long long tr0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long tr1 = 0;
#endif
    src = tile_src(m0, n0);
    wave_live = n0 + wc * 64 < p.N;
    if (CONV) set_offsets(m0, false);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    // ---- prologue: K-tile 0 complete, A_0 / B_0 of K-tile 1 in flight ----
    if (!k0_staged) stage_ktile0(src);
    if (nk > 1) {
      stage_a(src, I1{}, I0{}, 1); stage_w(src, I1{}, I0{}, 1);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    bar();
    if (wr == 1) bar();                                   // the second wave row runs one barrier behind the first

    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
      ktile(I0{}, kt);
      ktile(I1{}, kt + 1);
    }
    if (kt < nk) ktile(I0{}, kt);
    if (wr == 0) bar();
    __syncthreads();                                      // every fragment read of this tile has been waited for
#ifdef LD_GEMM_TRACE
    tr1 = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- epilogue, with the next tile's first K-tile requested from inside it ----
    const int vn = v + gridDim.x;
    bool hooked = false;
    Src nsrc = src;
    k0_staged = false;
    if (PREFETCH && !CONV && vn < v_end) {                // (a convolution's next-tile A offsets would need a second register set)
      int m1, n1;
      tile_origin(vn, m1, n1);
      nsrc = tile_src(m1, n1);
      k0_staged = true;
    }
    auto hook = [&]() {
      if (!hooked && k0_staged) stage_ktile0(nsrc);
      hooked = true;
    };
    if constexpr (EPI == EPI_QKV) qkv_epilogue16<4>(p, acc, smem + EPI_OFF, wave, lane, m0 + wr * 128, n0 + wc * 64, hook);
    else gemm_epilogue16<4, EPI, 4, SWAPACC, decltype(hook)&, CONV>(p, acc, 0, smem + EPI_OFF, wave, lane, m0 + wr * 128, n0 + wc * 64, hook);
    hook();
#ifdef LD_GEMM_TRACE
    if (tid == 0 && g_gemm_trace) {
      unsigned hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      const unsigned long long slot = __hip_atomic_fetch_add(g_gemm_trace, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((long long)slot < g_gemm_trace_cap) {
        unsigned long long* rec = g_gemm_trace + 1 + slot * 4;
        rec[0] = tr0; rec[1] = tr1; rec[2] = __builtin_amdgcn_s_memrealtime();
        rec[3] = ((unsigned long long)(xcc & 0xf) << 48) | ((unsigned long long)(hw & 0xffff) << 32) | (unsigned)v;
      }
    }
#endif
    if (vn < v_end) __syncthreads();                      // the staging region is free again before buffer-1 slots are re-staged
  }
}

#ifdef LD_VARIANTS   // measured alternative, not in the shipped library
// ------------------------------------------------------------------------------------------------
// The 8-phase loop on a 512 x 128 tile (round 5): outputs 128 columns wide (the VAE's Cout = 128 level at 480 x 720: 40 % of its
// convolution time).  On the 256 x 256 tile such an output leaves the wave columns 2 and 3 -- two of the four SIMDs -- without
// work; the 128 x 128 two-stage kernel they ran on instead reaches ~1085 TFLOP/s where the 8-phase loop reaches ~1250.  Here the
// eight waves form 4 wave ROWS x 2 wave columns with the SAME wave tile as ld_gemm8p_kernel (128 x 64, [8][4] accumulators, 16
// MFMAs per phase) and the same phase schedule, barriers and one-barrier skew between the two waves of a SIMD; what changes is the
// LDS plan: an A half-tile is 4 x 64 rows (32 KB, four 1 KB LDS-DMA pieces per wave), a W half-tile 2 x 32 columns (8 KB, one
// piece per wave), a K-tile 80 KB, two of them the whole 160 KB -- so the epilogue staging reuses K-tile buffer 0 behind the
// barrier that ends the main loop, and a persistent workgroup does not prefetch its next tile's first K-tile (the convolution
// form of ld_gemm8p_kernel does not either).  The counted wait of ph3 leaves 4 + 1 pieces in flight.  Same dot products in the
// same order as the other two conv routes: bit-identical outputs.
// MEASURED (profiles/r05_vae_conv_route_ab.txt): alone in a loop 2.37 -> 2.16 ms per 8-frame launch (1135 vs 1031 TFLOP/s); inside
// the VAE decode, same box, arms alternated: 337.4 / 336.7 ms per video against 337.4 / 335.7 -- no gain (in context the 128 x 128
// tiles already run at ~1085) -- so it is a measured alternative of the VARIANTS build (LD_GEMM_M512=1 there), not a shipped route.
// (For the DiT GEMMs' half-empty last tile column it would not pay at the headline shape: DESIGN.md section 9.)
// ------------------------------------------------------------------------------------------------
template <bool CONV, int EPI>
__global__ __launch_bounds__(512, 2) void ld_gemm8p_m512_kernel(GemmParams p) {
  static_assert(EPI != EPI_QKV, "the fused qkv split is not built for the 512 x 128 tile");
  constexpr int BM = 512, BN = 128;
  constexpr int ASLOT = 256 * 128, BSLOT = 64 * 128;      // A half-tile: 4 wave rows x 64 rows (32 KB); W half-tile: 2 wave columns x 32 (8 KB)
  constexpr int KBUF = 2 * ASLOT + 2 * BSLOT;             // A0 A1 B0 B1 per K-tile = 80 KB; two buffers = all 160 KB
  static_assert(2 * KBUF == LD_LDS_TOTAL && 8 * 32 * CW_STRIDE * 4 <= KBUF, "LDS plan");
  constexpr int EPI_OFF = 0;                              // the epilogue staging reuses K-tile buffer 0 (entered behind a workgroup barrier)
  constexpr bool SWAPACC = true;                          // C^T accumulator blocks: 16-byte epilogue staging stores (gemm_epilogue16<SWAP>)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wq = wave & 3;                // waves w and w + 4 share a SIMD: they differ in wr only
  const int wrow = (wq >> 1) * 2 + wr, wc = wq & 1;       // wave row 0..3 (128 tile rows each), wave column 0..1 (64 columns each)

  const int nbm = (p.M - p.m_begin + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  const int ntiles = nbm * nbn;
  const int gm_sz = p.group_m;
  // virtual block id v -> tile origin (XCD-contiguous logical id -> grouped raster, as ld_gemm_kernel).  gridDim.x is a
  // multiple of 8 whenever a workgroup owns more than one tile, so v % 8 == blockIdx.x % 8: a workgroup's tiles stay on its XCD.
  auto tile_origin = [&](int v, int& m0, int& n0) {
    const int bid = xcd_remap(v, ntiles);
    const int per_group = gm_sz * nbn;
    const int group = bid / per_group, in_group = bid - group * per_group;
    const int first_m = group * gm_sz;
    const int rows_here = (nbm - first_m) < gm_sz ? (nbm - first_m) : gm_sz;
    m0 = p.m_begin + (first_m + in_group % rows_here) * BM;
    n0 = (in_group / rows_here) * BN;
  };

  // ---- LDS-DMA sources: this wave stages pieces 2 * wave + {0, 1} (8 local rows x 128 B each) of every half-tile ----
  // Raw buffer descriptors (A: based at the tile's first row, rows past M read as zeros; convolution: the whole padded input,
  // rows clamped), one 32-bit byte offset per [half][piece] in VGPRs, the K-tile (or filter tap) offset in an SGPR.
  // (The descriptors are rebuilt from their scalars at every use -- loop-invariant SGPR values for the compiler; a
  //  __amdgpu_buffer_rsrc_t object captured by nested generic lambdas does not get through the host pass.)
  const auto clip = [](long v) { return (int)(v < 0x7fffffffL ? v : 0x7fffffffL); };
  struct Src { const bf16_t* a; const bf16_t* w; int a_bytes, w_bytes; };
  auto tile_src = [&](int m0, int n0) {
    Src s;
    s.a = p.A + (CONV ? 0 : (long)m0 * p.lda);
    s.w = p.W + (long)n0 * p.K;
    s.a_bytes = CONV ? 0x7fffffff : clip(((long)(p.M - m0) * p.lda) * 2);
    s.w_bytes = clip(((long)(p.N - n0) * p.K) * 2);
    return s;
  };
  uint32_t offA[2][4], offW[2];                           // [half][piece] byte offsets: four A pieces and one W piece per half-tile and wave
  auto set_offsets = [&](int m0, bool weights) {          // (A offsets depend on the tile only for a convolution)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int lr = wave * 32 + i * 8 + (lane >> 3);     // local row of the A slot, 0 .. 255
      const int chunk = (lane & 7) ^ ((lr >> 1) & 7);     // source-side swizzle (the read applies the same key)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int tm = (lr >> 6) * 128 + h * 64 + (lr & 63);
        if (CONV) {
          int gm = m0 + tm; gm = gm < p.M ? gm : p.M - 1;
          const int hw = p.H * p.W_;
          const int t = gm / hw, rem = gm - t * hw;
          const int hh = rem / p.W_, w = rem - hh * p.W_;
          offA[h][i] = (uint32_t)(((((long)t * p.Hp + hh) * p.Wp + w) * p.Cin + chunk * 8) * 2);
        } else {
          offA[h][i] = (uint32_t)(((long)tm * p.lda + chunk * 8) * 2);
        }
      }
    }
    if (weights) {
      const int lr = wave * 8 + (lane >> 3);              // local row of the W slot, 0 .. 63
      const int chunk = (lane & 7) ^ ((lr >> 1) & 7);
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int tn = (lr >> 5) * 64 + g * 32 + (lr & 31);
        offW[g] = (uint32_t)(((long)tn * p.K + chunk * 8) * 2);
      }
    }
  };
  const int nk = p.K / BK;
  const int cpt = CONV ? p.Cin / BK : 1;
  auto koff_a = [&](int kt) -> int {                      // byte offset of K-tile kt within an A row
    if (CONV) {
      const int tap = kt / cpt, c0 = (kt - tap * cpt) * BK;
      const int khw = p.kH * p.kW;
      const int dt = tap / khw, r2 = tap - dt * khw;
      const int dh = r2 / p.kW, dw = r2 - dh * p.kW;
      return (int)(((((long)dt * p.Hp + dh) * p.Wp + dw) * p.Cin + c0) * 2);
    }
    return kt * (BK * 2);
  };
  char* const my_a = smem + wave * 4096;                  // + buffer * KBUF + half * ASLOT (+ 1024 per further piece)
  char* const my_w = smem + 2 * ASLOT + wave * 1024;      // + buffer * KBUF + half * BSLOT
  Src src;                                                // the tile being computed
  auto stage_a = [&](const Src& s, auto bufc, auto hc, int kt) {
    constexpr int OFF = decltype(bufc)::value * KBUF + decltype(hc)::value * ASLOT;
    const int ko = koff_a(kt);
    stage_pieces<OFF>(s.a, s.a_bytes, my_a, offA[decltype(hc)::value][0], offA[decltype(hc)::value][1], ko);
    stage_pieces<OFF + 2048>(s.a, s.a_bytes, my_a, offA[decltype(hc)::value][2], offA[decltype(hc)::value][3], ko);
  };
  auto stage_w = [&](const Src& s, auto bufc, auto gc, int kt) {
    constexpr int OFF = decltype(bufc)::value * KBUF + decltype(gc)::value * BSLOT;
    stage_piece1<OFF>(s.w, s.w_bytes, my_w, offW[decltype(gc)::value], kt * (BK * 2));
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  auto stage_ktile0 = [&](const Src& s) {
    stage_a(s, I0{}, I0{}, 0); stage_w(s, I0{}, I0{}, 0); stage_w(s, I0{}, I1{}, 0); stage_a(s, I0{}, I1{}, 0);
  };

  // fragment reads: 16x16x32 operand = row (lane & 15), 16-byte chunk ks * 4 + (lane >> 4) of the 128-byte K row; the swizzle
  // key ((row >> 1) & 7) depends on lane & 15 only (block and wave offsets are multiples of 16 rows), so the blocks of a
  // subtile are immediate offsets (+2048 B) of one address per k-step
  int rdA[2], rdB[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int c = (ks * 4 + (lane >> 4)) ^ (((lane & 15) >> 1) & 7);
    rdA[ks] = (wrow * 64 + (lane & 15)) * 128 + (c << 4);
    rdB[ks] = 2 * ASLOT + (wc * 32 + (lane & 15)) * 128 + (c << 4);
  }
  f32x4_t acc[8][4];
  bf16x8_t a[4][2], b0[2][2], b1[2][2];
  auto read_a = [&](auto bufc, auto hc) {
    constexpr int OFF = decltype(bufc)::value * KBUF + decltype(hc)::value * ASLOT;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) a[i][ks] = *(const bf16x8_t*)(smem + rdA[ks] + OFF + i * 2048);
  };
  auto read_b = [&](auto bufc, auto gc, bf16x8_t (&b)[2][2]) {
    constexpr int OFF = decltype(bufc)::value * KBUF + decltype(gc)::value * BSLOT;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) b[j][ks] = *(const bf16x8_t*)(smem + rdB[ks] + OFF + j * 2048);
  };
  bool wave_live = true;                                  // (a wave whose 64 columns lie past N issues no MFMAs)
  auto mma = [&](auto hc, auto gc, bf16x8_t (&b)[2][2]) {
    constexpr int H = decltype(hc)::value, G = decltype(gc)::value;
    // lgkmcnt(0) as the BUILTIN (simm16 0xC07F = vmcnt 63, expcnt 7, lgkmcnt 0): hipcc's own wait-count bookkeeping sees it.  As
    // inline asm it is invisible to that pass, which then re-waits before the next phase's fragment reads on the path that
    // skips the MFMAs (a pending ds_read into a register it is about to reuse) -- serialising the B and A reads of ph0.
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
    if (wave_live) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[H * 4 + i][G * 2 + j] = SWAPACC ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][ks], a[i][ks], acc[H * 4 + i][G * 2 + j], 0, 0, 0)
                                                : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][ks], b[j][ks], acc[H * 4 + i][G * 2 + j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto bar = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto ktile = [&](auto bufc, int kt) {
    constexpr int B = decltype(bufc)::value;
    using Bc = std::integral_constant<int, B>;
    using Nc = std::integral_constant<int, B ^ 1>;
    // ph0
    read_b(Bc{}, I0{}, b0);
    __builtin_amdgcn_sched_barrier(0);
    read_a(Bc{}, I0{});
    if (kt + 1 < nk) stage_w(src, Nc{}, I1{}, kt + 1);
    bar(); mma(I0{}, I0{}, b0); bar();
    // ph1
    read_b(Bc{}, I1{}, b1);
    if (kt + 1 < nk) stage_a(src, Nc{}, I1{}, kt + 1);
    bar(); mma(I0{}, I1{}, b1); bar();
    // ph2
    read_a(Bc{}, I1{});
    if (kt + 2 < nk) stage_a(src, Bc{}, I0{}, kt + 2);
    bar(); mma(I1{}, I1{}, b1); bar();
    // ph3
    if (kt + 2 < nk) {
      stage_w(src, Bc{}, I0{}, kt + 2);
      asm volatile("s_waitcnt vmcnt(5)" ::: "memory");    // K-tile kt + 1 has landed; A_0 (4 pieces) / B_0 (1) of kt + 2 stay in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    bar(); mma(I1{}, I0{}, b0); bar();
  };

  set_offsets(0, true);
  bool k0_staged = false;                                 // K-tile 0 of the tile about to start is already on its way
  const int v_end = p.tile_end > 0 ? p.tile_end : ntiles; // (the tiles behind it: ld_gemm8p_n128_kernel)
  for (int v = p.tile_begin + blockIdx.x; v < v_end; v += gridDim.x) {
    int m0, n0;
    tile_origin(v, m0, n0);
    src = tile_src(m0, n0);
    wave_live = n0 + wc * 64 < p.N;
    if (CONV) set_offsets(m0, false);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    // ---- prologue: K-tile 0 complete, A_0 / B_0 of K-tile 1 in flight ----
    if (!k0_staged) stage_ktile0(src);
    if (nk > 1) {
      stage_a(src, I1{}, I0{}, 1); stage_w(src, I1{}, I0{}, 1);
      asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    bar();
    if (wr == 1) bar();                                   // the second wave row runs one barrier behind the first

    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
      ktile(I0{}, kt);
      ktile(I1{}, kt + 1);
    }
    if (kt < nk) ktile(I0{}, kt);
    if (wr == 0) bar();
    __syncthreads();                                      // every fragment read of this tile has been waited for

    // ---- epilogue, with the next tile's first K-tile requested from inside it ----
    const int vn = v + gridDim.x;
    bool hooked = false;
    Src nsrc = src;
    k0_staged = false;
    // (no prefetch of the next tile's first K-tile: the epilogue staging lives in buffer 0)
    auto hook = [&]() {
      if (!hooked && k0_staged) stage_ktile0(nsrc);
      hooked = true;
    };
    gemm_epilogue16<4, EPI, 4, SWAPACC, decltype(hook)&, CONV>(p, acc, 0, smem + EPI_OFF, wave, lane, m0 + wrow * 128, n0 + wc * 64, hook);
    hook();
    if (vn < v_end) __syncthreads();                      // the staging region (K-tile buffer 0) is free again before it is re-staged
  }
}


#endif  // LD_VARIANTS

// ------------------------------------------------------------------------------------------------
// The same 8-phase loop on 256 x 128 HALF tiles (round 5): the partial last round of a launch.  A GEMM whose 256 x 256 tiles
// do not fill whole rounds of the chip used to send its last tile ROWS to a second launch of 128 x 128 two-stage tiles: two
// workgroups per CU that share the matrix pipe, 1.4 quarter tiles per CU on average and two on the CUs that set the time -- 9 % of
// a DiT layer-call's GEMM time for 4 % of its tiles (profiles/r04_gemm_tile_trace.txt).  Here the r < 128 tiles behind the whole
// rounds (the tiles [tile_begin, ntiles) of the SAME raster, so the main launch is exactly `rounds` tiles per CU) are cut in two
// along N and run one per CU: 2 r <= 256 workgroups, one round, each half the work of a full tile.
//   * 8 waves as 4 x 2, wave tile 64 x 64 = [4][4] accumulators; per K-tile and wave 8 A + 8 W fragment reads for 32 MFMAs (the
//     2 x 4 layout of the full tile on 128 columns would need 16 + 4) -- 128 KB of LDS reads per K-tile against 1088 MFMA cycles;
//   * LDS: 2 K-tile buffers x (A_0, A_1: 16 KB = for all four wave rows wr the 32 tile rows wr * 64 + h * 32 ..; W_0, W_1: 8 KB =
//     for both wave columns wc the 32 tile columns wc * 64 + g * 32 ..) = 96 KB; a wave stages two 1 KB pieces of every A half
//     and one of every W half: 6 LDS-DMA instructions per K-tile;
//   * phases, staging order, counted vmcnt (3 = A_0 + W_0 of K-tile t + 2), the one-barrier skew between waves 0-3 and 4-7 (the
//     two waves of a SIMD), persistent loop and epilogues: those of ld_gemm8p_kernel; same dot products in the same order ->
//     the same bits as any other tiling of the GEMM.
// ------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(512, 2) void ld_gemm8p_n128_kernel(GemmParams p) {
  constexpr int BM = 256, BNF = 256;                      // the raster is the full tiles'
  constexpr int SLOT_A = 128 * 128, SLOT_B = 64 * 128, KBUF = 2 * SLOT_A + 2 * SLOT_B;     // 48 KB per K-tile: A0 A1 B0 B1
  constexpr int EPI_BYTES = (EPI == EPI_QKV) ? 8 * QKV_REGION : 8 * 32 * CW_STRIDE * 4;
  constexpr int EPI_OFF = (LD_LDS_TOTAL - EPI_BYTES) & ~15;
  constexpr bool PREFETCH = EPI_OFF >= KBUF;
  constexpr bool SWAPACC = EPI != EPI_QKV;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const bool late = wave >= 4;                            // the second wave of each SIMD runs one barrier behind the first

  const int nbm = (p.M - p.m_begin + BM - 1) / BM, nbn = (p.N + BNF - 1) / BNF;
  const int ntiles = nbm * nbn;
  const int gm_sz = p.group_m;
  auto tile_origin = [&](int v, int& m0, int& n0) {       // (ld_gemm8p_kernel's)
    const int bid = xcd_remap(v, ntiles);
    const int per_group = gm_sz * nbn;
    const int group = bid / per_group, in_group = bid - group * per_group;
    const int first_m = group * gm_sz;
    const int rows_here = (nbm - first_m) < gm_sz ? (nbm - first_m) : gm_sz;
    m0 = p.m_begin + (first_m + in_group % rows_here) * BM;
    n0 = (in_group / rows_here) * BNF;
  };
  const int v_end = p.tile_end > 0 ? p.tile_end : ntiles;
  const int nhalf = 2 * (v_end - p.tile_begin);           // work items: half u of tile tile_begin + (u >> 1)
  auto half_origin = [&](int u, int& m0, int& n0) {
    tile_origin(p.tile_begin + (u >> 1), m0, n0);
    n0 += (u & 1) * 128;
  };

  const auto clip = [](long v) { return (int)(v < 0x7fffffffL ? v : 0x7fffffffL); };
  struct Src { const bf16_t* a; const bf16_t* w; int a_bytes, w_bytes; };
  auto tile_src = [&](int m0, int n0) {
    Src s;
    s.a = p.A + (long)m0 * p.lda;
    s.w = p.W + (long)n0 * p.K;
    s.a_bytes = clip(((long)(p.M - m0) * p.lda) * 2);
    s.w_bytes = n0 < p.N ? clip(((long)(p.N - n0) * p.K) * 2) : 0;
    return s;
  };
  uint32_t offA[2][2], offW[2];                           // A: [half][piece], W: [half]
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int lr = wave * 16 + i * 8 + (lane >> 3);       // local row of an A slot, 0 .. 127
    const int chunk = (lane & 7) ^ ((lr >> 1) & 7);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int tm = (lr >> 5) * 64 + h * 32 + (lr & 31);
      offA[h][i] = (uint32_t)(((long)tm * p.lda + chunk * 8) * 2);
    }
  }
  {
    const int lr = wave * 8 + (lane >> 3);                // local row of a W slot, 0 .. 63
    const int chunk = (lane & 7) ^ ((lr >> 1) & 7);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int tn = (lr >> 5) * 64 + g * 32 + (lr & 31);
      offW[g] = (uint32_t)(((long)tn * p.K + chunk * 8) * 2);
    }
  }
  const int nk = p.K / BK;
  char* const a_piece = smem + wave * 2048;               // + buffer * KBUF + h * SLOT_A (+ 1024 for the second piece)
  char* const w_piece = smem + 2 * SLOT_A + wave * 1024;  // + buffer * KBUF + g * SLOT_B
  Src src;
  auto stage_a = [&](const Src& s, auto bufc, auto hc, int kt) {
    constexpr int OFF = decltype(bufc)::value * KBUF + decltype(hc)::value * SLOT_A;
    stage_pieces<OFF>(s.a, s.a_bytes, a_piece, offA[decltype(hc)::value][0], offA[decltype(hc)::value][1], kt * (BK * 2));
  };
  auto stage_w = [&](const Src& s, auto bufc, auto gc, int kt) {
    constexpr int OFF = decltype(bufc)::value * KBUF + decltype(gc)::value * SLOT_B;
    stage_piece1<OFF>(s.w, s.w_bytes, w_piece, offW[decltype(gc)::value], kt * (BK * 2));
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  auto stage_ktile0 = [&](const Src& s) {
    stage_a(s, I0{}, I0{}, 0); stage_w(s, I0{}, I0{}, 0); stage_w(s, I0{}, I1{}, 0); stage_a(s, I0{}, I1{}, 0);
  };

  int rdA[2], rdB[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int c = (ks * 4 + (lane >> 4)) ^ (((lane & 15) >> 1) & 7);
    rdA[ks] = (wr * 32 + (lane & 15)) * 128 + (c << 4);
    rdB[ks] = 2 * SLOT_A + (wc * 32 + (lane & 15)) * 128 + (c << 4);
  }
  f32x4_t acc[4][4];
  bf16x8_t a[2][2], b0[2][2], b1[2][2];
  auto read_a = [&](auto bufc, auto hc) {
    constexpr int OFF = decltype(bufc)::value * KBUF + decltype(hc)::value * SLOT_A;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) a[i][ks] = *(const bf16x8_t*)(smem + rdA[ks] + OFF + i * 2048);
  };
  auto read_b = [&](auto bufc, auto gc, bf16x8_t (&b)[2][2]) {
    constexpr int OFF = decltype(bufc)::value * KBUF + decltype(gc)::value * SLOT_B;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) b[j][ks] = *(const bf16x8_t*)(smem + rdB[ks] + OFF + j * 2048);
  };
  bool wave_live = true;
  auto mma = [&](auto hc, auto gc, bf16x8_t (&b)[2][2]) {
    constexpr int H = decltype(hc)::value, G = decltype(gc)::value;
    __builtin_amdgcn_s_waitcnt(0xC07F);                   // lgkmcnt(0), as the builtin (see ld_gemm8p_kernel)
    __builtin_amdgcn_sched_barrier(0);
    if (wave_live) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[H * 2 + i][G * 2 + j] = SWAPACC ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][ks], a[i][ks], acc[H * 2 + i][G * 2 + j], 0, 0, 0)
                                                : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][ks], b[j][ks], acc[H * 2 + i][G * 2 + j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto bar = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
#ifndef LD_GEMM_PH4   // two phases of 16 MFMAs per K-tile (ld_gemm8p_kernel's round-6 loop on the half tile)
  auto mma2 = [&](auto hc, auto g0c, bf16x8_t (&bA)[2][2], auto g1c, bf16x8_t (&bB)[2][2]) {
    constexpr int H = decltype(hc)::value, G0 = decltype(g0c)::value, G1 = decltype(g1c)::value;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
    if (wave_live) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[H * 2 + i][G0 * 2 + j] = SWAPACC ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(bA[j][ks], a[i][ks], acc[H * 2 + i][G0 * 2 + j], 0, 0, 0)
                                                 : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][ks], bA[j][ks], acc[H * 2 + i][G0 * 2 + j], 0, 0, 0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[H * 2 + i][G1 * 2 + j] = SWAPACC ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(bB[j][ks], a[i][ks], acc[H * 2 + i][G1 * 2 + j], 0, 0, 0)
                                                 : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][ks], bB[j][ks], acc[H * 2 + i][G1 * 2 + j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto ktile = [&](auto bufc, int kt) {      // (LDS-DMA instructions per wave: an A half = 2, a W half = 1)
    constexpr int B = decltype(bufc)::value;
    using Bc = std::integral_constant<int, B>;
    using Nc = std::integral_constant<int, B ^ 1>;
    // P0.  In flight on entry (oldest first): A_1(kt) [2], A_0 / W_0(kt + 1) [3]
    read_b(Bc{}, I0{}, b0);
    read_b(Bc{}, I1{}, b1);
    __builtin_amdgcn_sched_barrier(0);
    read_a(Bc{}, I0{});
    if (kt + 1 < nk) {
      stage_w(src, Nc{}, I1{}, kt + 1); stage_a(src, Nc{}, I1{}, kt + 1);
      asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");   // A_1(kt) has landed
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    bar(); mma2(I0{}, I0{}, b0, I1{}, b1); bar();
    // P1.  In flight: A_0 / W_0(kt + 1) [3], W_1 / A_1(kt + 1) [3]
    read_a(Bc{}, I1{});
    if (kt + 2 < nk) {
      stage_a(src, Bc{}, I0{}, kt + 2); stage_w(src, Bc{}, I0{}, kt + 2);
      asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");   // A_0 / W_0 / W_1 of K-tile kt + 1 have landed
    } else if (kt + 1 < nk) {
      asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    bar(); mma2(I1{}, I1{}, b1, I0{}, b0); bar();
  };
#else
  auto ktile = [&](auto bufc, int kt) {
    constexpr int B = decltype(bufc)::value;
    using Bc = std::integral_constant<int, B>;
    using Nc = std::integral_constant<int, B ^ 1>;
    // ph0
    read_b(Bc{}, I0{}, b0);
    __builtin_amdgcn_sched_barrier(0);
    read_a(Bc{}, I0{});
    if (kt + 1 < nk) stage_w(src, Nc{}, I1{}, kt + 1);
    bar(); mma(I0{}, I0{}, b0); bar();
    // ph1
    read_b(Bc{}, I1{}, b1);
    if (kt + 1 < nk) stage_a(src, Nc{}, I1{}, kt + 1);
    bar(); mma(I0{}, I1{}, b1); bar();
    // ph2
    read_a(Bc{}, I1{});
    if (kt + 2 < nk) stage_a(src, Bc{}, I0{}, kt + 2);
    bar(); mma(I1{}, I1{}, b1); bar();
    // ph3
    if (kt + 2 < nk) {
      stage_w(src, Bc{}, I0{}, kt + 2);
      asm volatile("s_waitcnt vmcnt(3)" ::: "memory");    // K-tile kt + 1 has landed; A_0 (2 pieces) / W_0 (1) of kt + 2 stay in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    bar(); mma(I1{}, I0{}, b0); bar();
  };
#endif

  bool k0_staged = false;
  for (int u = blockIdx.x; u < nhalf; u += gridDim.x) {
    int m0, n0;
    half_origin(u, m0, n0);
    const int un = u + gridDim.x;
    if (n0 >= p.N) continue;                              // the empty half of a tile in a half-wide last column (never prefetched for)
    src = tile_src(m0, n0);
    wave_live = n0 + wc * 64 < p.N;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    if (!k0_staged) stage_ktile0(src);
    if (nk > 1) {
      stage_a(src, I1{}, I0{}, 1); stage_w(src, I1{}, I0{}, 1);
      asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    bar();
    if (late) bar();

    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
      ktile(I0{}, kt);
      ktile(I1{}, kt + 1);
    }
    if (kt < nk) ktile(I0{}, kt);
    if (!late) bar();
    __syncthreads();

    bool hooked = false;
    Src nsrc = src;
    k0_staged = false;
    if (PREFETCH && un < nhalf) {
      int m1, n1;
      half_origin(un, m1, n1);
      if (n1 < p.N) { nsrc = tile_src(m1, n1); k0_staged = true; }
    }
    auto hook = [&]() {
      if (!hooked && k0_staged) stage_ktile0(nsrc);
      hooked = true;
    };
    if constexpr (EPI == EPI_QKV) qkv_epilogue16<2>(p, acc, smem + EPI_OFF, wave, lane, m0 + wr * 64, n0 + wc * 64, hook);
    else gemm_epilogue16<2, EPI, 4, SWAPACC>(p, acc, 0, smem + EPI_OFF, wave, lane, m0 + wr * 64, n0 + wc * 64, hook);
    hook();
    if (un < nhalf) __syncthreads();
  }
}


// ------------------------------------------------------------------------------------------------
// MXFP8 operands on the persistent loop of ld_gemm8p_kernel (round 6; BASELINE configs[4]).  A K-tile of 128 e4m3 elements is a
// 128-byte row, exactly the bf16 kernel's 64-element row: the same eight 16 KB half-tile slots, the same per-lane LDS-DMA offsets
// (in bytes), the same XOR swizzle, the same two phases of the K-tile with the two wave rows one barrier apart, the same persistent
// tile walk, next-tile request from inside the epilogue and the same epilogues (incl. the fused qkv head split and the MXFP8-writing
// GELU epilogue), because v_mfma_scale_f32_16x16x128_f8f6f4 leaves its 16 x 16 block in the registers of v_mfma_f32_16x16x32_bf16.
// What changes:
//   * one MFMA per accumulator block and K-tile (32 per wave, ~32 cycles each) instead of two; its 32-byte operand is the PAIR of
//     16-byte fragments the bf16 loop reads for its two k-steps -- lane (r, g) holds k = 16 g .. + 16 and 64 + 16 g .. + 16 of row r
//     (tools/probe/fp8_mfma16_layout.hip: layout D1) -- so the fragment reads are the bf16 kernel's, address for address;
//   * block scales: one E8M0 byte per row and 32 K elements, [K / 128][rows][4] in memory (ld_quantize_mxfp8).  A K-tile's 256 + 256
//     row dwords travel by 4-byte LDS-DMA next to A_0 / B_0 (waves 0-3: A rows, 4-7: W rows) into a 2 KB strip per K-tile buffer; in
//     P0 a lane reads the dwords of its 8 + 4 block rows and keeps byte g (the hardware takes block g's scale from lane group g:
//     scale layout S0) of each, packed four to a register -- the MFMA's op_sel picks the byte.
// LDS: [K-tile buffer 0: 64 KB][scale strips: 2 x 2 KB][K-tile buffer 1: 64 KB]; the epilogue staging at the end of the 160 KB stays
// clear of buffer 0 and the strips.
// ------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(512, 2) void ld_gemm8p_mx_kernel(GemmParams p) {
  constexpr int BM = 256, BN = 256, KB = 128;             // KB: bytes (= e4m3 elements) of K per tile
  constexpr int SLOT = 128 * 128, KBUF = 4 * SLOT;
  constexpr int SC_OFF = KBUF, SC_BYTES = 4096;           // [buffer][A rows 1 KB | W rows 1 KB]
  constexpr int BUF1 = KBUF + SC_BYTES;                   // byte offset of K-tile buffer 1
  constexpr int EPI_BYTES = (EPI == EPI_QKV) ? 8 * QKV_REGION : 8 * 32 * CW_STRIDE * 4;
  constexpr int EPI_OFF = (LD_LDS_TOTAL - EPI_BYTES) & ~15;
  constexpr bool PREFETCH = EPI_OFF >= BUF1 && EPI != EPI_QKV;
  constexpr bool SWAPACC = EPI != EPI_QKV;
  static_assert(BUF1 + KBUF <= LD_LDS_TOTAL, "LDS layout");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  const int ntiles = nbm * nbn;
  const int gm_sz = p.group_m;
  auto tile_origin = [&](int v, int& m0, int& n0) {       // (ld_gemm8p_kernel's)
    const int bid = xcd_remap(v, ntiles);
    const int per_group = gm_sz * nbn;
    const int group = bid / per_group, in_group = bid - group * per_group;
    const int first_m = group * gm_sz;
    const int rows_here = (nbm - first_m) < gm_sz ? (nbm - first_m) : gm_sz;
    m0 = (first_m + in_group % rows_here) * BM;
    n0 = (in_group / rows_here) * BN;
  };
  const auto clip = [](long v) { return (int)(v < 0x7fffffffL ? v : 0x7fffffffL); };
  const long ldab = p.lda, ldwb = p.K;                    // row strides in bytes
  struct Src { const unsigned char* a; const unsigned char* w; const unsigned char* s; int a_bytes, w_bytes, s_bytes; };
  const bool a_wave = wave < 4;                           // which operand's scale dwords this wave fetches
  const long srows = a_wave ? p.M : p.N;
  auto tile_src = [&](int m0, int n0) {
    Src s;
    s.a = (const unsigned char*)p.A + (long)m0 * ldab;
    s.w = (const unsigned char*)p.W + (long)n0 * ldwb;
    s.a_bytes = clip((long)(p.M - m0) * ldab);
    s.w_bytes = clip((long)(p.N - n0) * ldwb);
    const long so = a_wave ? m0 : n0;
    s.s = (a_wave ? p.mx_a : p.mx_w) + so * 4;            // rows past M / N read as zero scale bytes (their products are never stored)
    s.s_bytes = clip(((long)(p.K >> 7) * srows - so) * 4);
    return s;
  };
  // [piece] byte offsets of half 0 (the bf16 kernel's rows and swizzle); half 1 = + 64 rows of A / + 32 rows of W, added per use by
  // an asm statement the compiler cannot hoist: with four more offset registers live through the K loop the gated-residual and
  // GELU instantiations spilled one of them, and the reload's s_waitcnt vmcnt(0) drained the LDS-DMA queue once per K-tile
  uint32_t offA[2], offW[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int lr = wave * 16 + i * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((lr >> 1) & 7);
    const int tm = (lr >> 6) * 128 + (lr & 63);
    const int tn = (lr >> 5) * 64 + (lr & 31);
    offA[i] = (uint32_t)((long)tm * ldab + chunk * 16);
    offW[i] = (uint32_t)((long)tn * ldwb + chunk * 16);
  }
  const int dA1 = (int)(64 * ldab), dW1 = (int)(32 * ldwb);
  auto half_off = [](uint32_t o, int d, auto hc) -> uint32_t {
    if constexpr (decltype(hc)::value == 0) return o;
    uint32_t r;
    asm volatile("v_add_u32 %0, %1, %2" : "=v"(r) : "v"(o), "s"(d));
    return r;
  };
  const uint32_t offS = (uint32_t)(((wave & 3) * 64 + lane) * 4);     // this lane's row dword of the strip
  const int sslab = (int)(srows * 4);                     // bytes between consecutive K-tiles' scale slabs
  const int nk = p.K / KB;
  char* const my_piece = smem + wave * 2048;
  Src src;
  auto stage_a = [&](const Src& s, auto bufc, auto hc, int kt) {
    constexpr int OFF = (decltype(bufc)::value ? BUF1 : 0) + decltype(hc)::value * SLOT;
    stage_pieces<OFF>((const bf16_t*)s.a, s.a_bytes, my_piece, half_off(offA[0], dA1, hc), half_off(offA[1], dA1, hc), kt * KB);
  };
  auto stage_w = [&](const Src& s, auto bufc, auto gc, int kt) {
    constexpr int OFF = (decltype(bufc)::value ? BUF1 : 0) + (2 + decltype(gc)::value) * SLOT;
    stage_pieces<OFF>((const bf16_t*)s.w, s.w_bytes, my_piece, half_off(offW[0], dW1, gc), half_off(offW[1], dW1, gc), kt * KB);
  };
  auto stage_s = [&](const Src& s, auto bufc, int kt) {     // 256 B per wave: the 64 row dwords (waves 0-3: A rows, 4-7: W rows)
    constexpr int OFF = SC_OFF + decltype(bufc)::value * 2048;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)s.s, 0, s.s_bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + OFF + wave * 256), 4, offS, kt * sslab, 0, 0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  auto stage_ktile0 = [&](const Src& s) {                 // 9 LDS-DMA instructions per wave
    stage_a(s, I0{}, I0{}, 0); stage_w(s, I0{}, I0{}, 0); stage_s(s, I0{}, 0); stage_w(s, I0{}, I1{}, 0); stage_a(s, I0{}, I1{}, 0);
  };

  int rdA[2], rdB[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {                        // 16-byte chunk g and chunk 4 + g of the row: the operand's two halves
    const int c = (ks * 4 + (lane >> 4)) ^ (((lane & 15) >> 1) & 7);
    rdA[ks] = (wr * 64 + (lane & 15)) * 128 + (c << 4);
    rdB[ks] = (wc * 32 + (lane & 15)) * 128 + (c << 4);
  }
  const int sh8 = (lane >> 4) * 8;                        // this lane group's scale byte inside a row dword
  f32x4_t acc[8][4];
  u32x4_t a[4][2], b0[2][2], b1[2][2];
  uint32_t sA[2] = {0u, 0u}, sB = 0u;                     // packed scale bytes: sA[h] byte i = block row i of half h; sB byte 2 g + j
  auto read_a = [&](auto bufc, auto hc) {
    constexpr int OFF = (decltype(bufc)::value ? BUF1 : 0) + decltype(hc)::value * SLOT;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) a[i][ks] = *(const u32x4_t*)(smem + rdA[ks] + OFF + i * 2048);
  };
  auto read_b = [&](auto bufc, auto gc, u32x4_t (&b)[2][2]) {
    constexpr int OFF = (decltype(bufc)::value ? BUF1 : 0) + (2 + decltype(gc)::value) * SLOT;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) b[j][ks] = *(const u32x4_t*)(smem + rdB[ks] + OFF + j * 2048);
  };
#ifndef LD_MX_SCALE_BYTES
#define LD_MX_SCALE_BYTES 1      // 1: the lane group's byte of a row dword by a byte load (12 ds_read_u8 + 9 shift-ors per K-tile); 0: dword loads + extract
#endif
  const int sbyteA = (wr * 128 + (lane & 15)) * 4 + (lane >> 4);       // byte g of this lane's row dword, block row 0 of half 0
  const int sbyteB = 1024 + (wc * 64 + (lane & 15)) * 4 + (lane >> 4);
  auto read_scales = [&](auto bufc) {
    const char* sc = smem + SC_OFF + decltype(bufc)::value * 2048;
#if LD_MX_SCALE_BYTES
    const unsigned char* sa = (const unsigned char*)sc + sbyteA;
    const unsigned char* sb = (const unsigned char*)sc + sbyteB;
    sA[0] = (uint32_t)sa[0] | ((uint32_t)sa[64] << 8) | ((uint32_t)sa[128] << 16) | ((uint32_t)sa[192] << 24);
    sB = (uint32_t)sb[0] | ((uint32_t)sb[64] << 8) | ((uint32_t)sb[128] << 16) | ((uint32_t)sb[192] << 24);     // byte 2 g + j: row g * 32 + j * 16
    // (all three in P0: the strip is restaged for K-tile kt + 2 by the OTHER wave row's P1, which runs while this row is in P1 too --
    //  reading the second half's scales only where they are first used, in P1, raced with that DMA and bought nothing)
    sA[1] = (uint32_t)sa[256] | ((uint32_t)sa[320] << 8) | ((uint32_t)sa[384] << 16) | ((uint32_t)sa[448] << 24);
#else
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      uint32_t pk = 0u;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t d = *(const uint32_t*)(sc + (wr * 128 + h * 64 + i * 16 + (lane & 15)) * 4);
        pk |= ((d >> sh8) & 0xffu) << (8 * i);
      }
      sA[h] = pk;
    }
    uint32_t pk = 0u;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uint32_t d = *(const uint32_t*)(sc + 1024 + (wc * 64 + g * 32 + j * 16 + (lane & 15)) * 4);
        pk |= ((d >> sh8) & 0xffu) << (8 * (2 * g + j));
      }
    sB = pk;
#endif
  };
  bool wave_live = true;
  auto frag = [](const u32x4_t (&f)[2]) {
    return (i32x8_t){(int)f[0][0], (int)f[0][1], (int)f[0][2], (int)f[0][3], (int)f[1][0], (int)f[1][1], (int)f[1][2], (int)f[1][3]};
  };
  auto mma1 = [&](auto hc, auto gc, u32x4_t (&b)[2][2]) {   // 8 MFMAs: the 64 x 32 quadrant (h, g)
    constexpr int H = decltype(hc)::value, G = decltype(gc)::value;
    auto one = [&](auto ic, auto jc) {                      // (op_sel is an immediate: block indices as types)
      constexpr int i = decltype(ic)::value, j = decltype(jc)::value;
      if constexpr (SWAPACC)
        acc[H * 4 + i][G * 2 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(frag(b[j]), frag(a[i]), acc[H * 4 + i][G * 2 + j], 0, 0, 2 * G + j, sB, i, sA[H]);
      else
        acc[H * 4 + i][G * 2 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(frag(a[i]), frag(b[j]), acc[H * 4 + i][G * 2 + j], 0, 0, i, sA[H], 2 * G + j, sB);
    };
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    one(I0{}, I0{}); one(I0{}, I1{}); one(I1{}, I0{}); one(I1{}, I1{});
    one(I2{}, I0{}); one(I2{}, I1{}); one(I3{}, I0{}); one(I3{}, I1{});
  };
  auto mma2 = [&](auto hc, auto g0c, u32x4_t (&bA)[2][2], auto g1c, u32x4_t (&bB)[2][2]) {
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
    if (wave_live) {
      __builtin_amdgcn_s_setprio(1);
      mma1(hc, g0c, bA);
      mma1(hc, g1c, bB);
      __builtin_amdgcn_s_setprio(0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto bar = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // LDS-DMA instructions per wave: a half-tile = 2, a scale strip = 1.  In flight on entry of P0(kt), oldest first:
  // A_1(kt) [2], then A_0 / B_0 / S(kt + 1) [5]
  auto ktile = [&](auto bufc, int kt) {
    constexpr int B = decltype(bufc)::value;
    using Bc = std::integral_constant<int, B>;
    using Nc = std::integral_constant<int, B ^ 1>;
    // P0
    read_b(Bc{}, I0{}, b0);
    read_b(Bc{}, I1{}, b1);
    __builtin_amdgcn_sched_barrier(0);
    read_a(Bc{}, I0{});
    read_scales(Bc{});
    if (kt + 1 < nk) {
      stage_w(src, Nc{}, I1{}, kt + 1); stage_a(src, Nc{}, I1{}, kt + 1);
      asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory");   // A_1(kt) has landed
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    bar(); mma2(I0{}, I0{}, b0, I1{}, b1); bar();
    // P1.  In flight: A_0 / B_0 / S(kt + 1) [5], B_1 / A_1(kt + 1) [4]
    read_a(Bc{}, I1{});
    if (kt + 2 < nk) {
      stage_a(src, Bc{}, I0{}, kt + 2); stage_w(src, Bc{}, I0{}, kt + 2); stage_s(src, Bc{}, kt + 2);
      asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");   // A_0 / B_0 / S / B_1 of K-tile kt + 1 have landed
    } else if (kt + 1 < nk) {
      asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    bar(); mma2(I1{}, I1{}, b1, I0{}, b0); bar();
  };

  bool k0_staged = false;
  for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
    int m0, n0;
    tile_origin(v, m0, n0);
    src = tile_src(m0, n0);
    wave_live = n0 + wc * 64 < p.N;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    if (!k0_staged) stage_ktile0(src);
    if (nk > 1) {
      stage_a(src, I1{}, I0{}, 1); stage_w(src, I1{}, I0{}, 1); stage_s(src, I1{}, 1);
      asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    bar();
    if (wr == 1) bar();
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
      ktile(I0{}, kt);
      ktile(I1{}, kt + 1);
    }
    if (kt < nk) ktile(I0{}, kt);
    if (wr == 0) bar();
    __syncthreads();

    const int vn = v + gridDim.x;
    bool hooked = false;
    Src nsrc = src;
    k0_staged = false;
    if (PREFETCH && vn < ntiles) {
      int m1, n1;
      tile_origin(vn, m1, n1);
      nsrc = tile_src(m1, n1);
      k0_staged = true;
    }
    auto hook = [&]() {
      if (!hooked && k0_staged) stage_ktile0(nsrc);
      hooked = true;
    };
    if constexpr (EPI == EPI_QKV) qkv_epilogue16<4>(p, acc, smem + EPI_OFF, wave, lane, m0 + wr * 128, n0 + wc * 64, hook);
    else gemm_epilogue16<4, EPI, 4, SWAPACC, decltype(hook)&>(p, acc, 0, smem + EPI_OFF, wave, lane, m0 + wr * 128, n0 + wc * 64, hook);
    hook();
    if (vn < ntiles) __syncthreads();
  }
}

#ifdef LD_VARIANTS   // measured alternatives, not in the shipped library (build.sh: LD_BUILD_VARIANTS=1)
// ------------------------------------------------------------------------------------------------
// Software-pipelined main loop (round 4; LD_GEMM_SP=1): the 256 x 256 x 64 tile / 8 waves (2 x 4, 128 x 64 per wave) /
// 16x16x32 MFMAs of ld_gemm8p_kernel, but every wave pipelines ITS OWN fragment reads under its own MFMAs and the workgroup
// meets ONCE per K-tile instead of eight times:
//   * a K-tile = 4 stages of 16 MFMAs (k-step ks x row half of the wave tile); while a stage's MFMAs issue, the wave's
//     ds_read_b128 for the NEXT stage go out in between them into a second fragment register set (two A sets + two W sets of
//     4 fragments = 64 registers next to the 128 accumulators);
//       S0 (ks 0, rows 0-63):   reads A rows 64-127 ks 0
//       S1 (ks 0, rows 64-127): reads A rows 0-63 ks 1 and W ks 1
//       S2 (ks 1, rows 0-63):   reads A rows 64-127 ks 1               -- the wave's last reads of this K-tile
//       [lgkmcnt(0), vmcnt(0): K-tile t+1 has landed; s_barrier: every wave is done reading K-tile t]
//       S3 (ks 1, rows 64-127): issues the 8 LDS-DMA pieces of K-tile t+2 into the buffer just freed and reads A rows 0-63 / W
//                               ks 0 of K-tile t+1 from the other buffer
//   * LDS: two K-tile buffers of 64 KB (A tile 256 rows x 128 B | W tile 256 rows x 128 B, chunk index XOR ((row >> 1) & 7) on
//     the DMA source and on the read); a wave stages pieces 4 w .. 4 w + 3 (8 rows each) of both tiles: one per-lane byte
//     offset per piece parity, the rest of the address in the SGPR offset.
//   * persistent tiles, epilogues and the next tile's first K-tile requested from inside the epilogue: as ld_gemm8p_kernel.
// The matrix pipe no longer waits for a partner wave to get through a load segment and seven of the eight barriers per K-tile
// are gone; what a DMA has to land in is one K-tile (~2300 cycles) instead of 1.5-2.
// ------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(512, 2) void ld_gemm_sp_kernel(GemmParams p) {
  constexpr int BM = 256, BN = 256;
  constexpr int TILE = 256 * 128, KBUF = 2 * TILE;        // A tile | W tile per K-tile buffer
  constexpr int EPI_BYTES = (EPI == EPI_QKV) ? 8 * QKV_REGION : 8 * 32 * CW_STRIDE * 4;
  constexpr int EPI_OFF = (LD_LDS_TOTAL - EPI_BYTES) & ~15;
  constexpr bool PREFETCH = EPI_OFF >= KBUF;
  constexpr bool SWAPACC = EPI != EPI_QKV;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int nbm = (p.M - p.m_begin + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  const int ntiles = nbm * nbn;
  const int gm_sz = p.group_m;
  auto tile_origin = [&](int v, int& m0, int& n0) {
    const int bid = xcd_remap(v, ntiles);
    const int per_group = gm_sz * nbn;
    const int group = bid / per_group, in_group = bid - group * per_group;
    const int first_m = group * gm_sz;
    const int rows_here = (nbm - first_m) < gm_sz ? (nbm - first_m) : gm_sz;
    m0 = p.m_begin + (first_m + in_group % rows_here) * BM;
    n0 = (in_group / rows_here) * BN;
  };
  const auto clip = [](long v) { return (int)(v < 0x7fffffffL ? v : 0x7fffffffL); };
  struct Src { const bf16_t* a; const bf16_t* w; int a_bytes, w_bytes; };
  auto tile_src = [&](int m0, int n0) {
    Src s;
    s.a = p.A + (long)m0 * p.lda;
    s.w = p.W + (long)n0 * p.K;
    s.a_bytes = clip(((long)(p.M - m0) * p.lda) * 2);
    s.w_bytes = clip(((long)(p.N - n0) * p.K) * 2);
    return s;
  };
  // staging: piece 4 * wave + i (i = 0..3) of each tile = local rows 32 * wave + 8 * i + (lane >> 3); the swizzle key
  // ((row >> 1) & 7) = (4 * i + (lane >> 4)) & 7 depends on the parity of i only, the 16-row step of i >> 1 goes into the SGPR offset
  uint32_t offA[2], offW[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int lr = wave * 32 + par * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((lr >> 1) & 7);
    offA[par] = (uint32_t)(((long)lr * p.lda + chunk * 8) * 2);
    offW[par] = (uint32_t)(((long)lr * p.K + chunk * 8) * 2);
  }
  const int stepA = (int)(16 * p.lda * 2), stepW = 16 * p.K * 2;       // bytes per 16 rows
  const int nk = p.K / BK;
  char* const my_piece = smem + wave * 4096;
  auto stage_piece = [&](const Src& s, auto bufc, auto ic, int kt, bool weights) {
    constexpr int B = decltype(bufc)::value, i = decltype(ic)::value;
    constexpr int OFF = B * KBUF + i * 1024;
    if (!weights) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)s.a, 0, s.a_bytes, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(my_piece + OFF), 16, offA[i & 1],
                                               kt * (BK * 2) + (i >> 1) * stepA, 0, 0);
    } else {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)s.w, 0, s.w_bytes, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(my_piece + TILE + OFF), 16, offW[i & 1],
                                               kt * (BK * 2) + (i >> 1) * stepW, 0, 0);
    }
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  auto stage_ktile = [&](const Src& s, auto bufc, int kt) {
    stage_piece(s, bufc, I0{}, kt, false); stage_piece(s, bufc, I0{}, kt, true);
    stage_piece(s, bufc, I1{}, kt, false); stage_piece(s, bufc, I1{}, kt, true);
    stage_piece(s, bufc, I2{}, kt, false); stage_piece(s, bufc, I2{}, kt, true);
    stage_piece(s, bufc, I3{}, kt, false); stage_piece(s, bufc, I3{}, kt, true);
  };
  // fragment reads
  int rdA[2], rdW[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int c = (ks * 4 + (lane >> 4)) ^ (((lane & 15) >> 1) & 7);
    rdA[ks] = (wr * 128 + (lane & 15)) * 128 + (c << 4);
    rdW[ks] = TILE + (wc * 64 + (lane & 15)) * 128 + (c << 4);
  }
  f32x4_t acc[8][4];
  bf16x8_t aP[4], aR[4], wQ[4], wS[4];
  auto ld_a = [&](bf16x8_t& d, auto bufc, int ks, int blk) { d = *(const bf16x8_t*)(smem + rdA[ks] + decltype(bufc)::value * KBUF + blk * 2048); };
  auto ld_w = [&](bf16x8_t& d, auto bufc, int ks, int blk) { d = *(const bf16x8_t*)(smem + rdW[ks] + decltype(bufc)::value * KBUF + blk * 2048); };
  bool wave_live = true;
  auto mm = [&](int ib, int j, const bf16x8_t& a, const bf16x8_t& w) {
    acc[ib][j] = SWAPACC ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, a, acc[ib][j], 0, 0, 0)
                         : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, w, acc[ib][j], 0, 0, 0);
  };
#define SPF() __builtin_amdgcn_sched_barrier(0)
  // one stage: 16 MFMAs on (a[0..3] -> row blocks r0 .. r0 + 3) x (w[0..3]); `side(g)` issues the stage's loads / DMA in gap g
  // (LIVE = false: a wave whose 64 columns lie past N takes part in the staging and the barriers but issues no MFMAs)
  auto stage16 = [&](auto livec, int r0, const bf16x8_t (&a)[4], const bf16x8_t (&w)[4], auto&& side) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      if constexpr (decltype(livec)::value) mm(r0 + (g >> 2), g & 3, a[g >> 2], w[g & 3]);
      side(g);
      SPF();
    }
  };
  // stage2c / morec (compile time): K-tile kt + 2 / kt + 1 exists -- the steady-state loop carries no tests
  auto ktile = [&](auto bufc, const Src& src, int kt, auto livec, auto stage2c, auto morec) {
    constexpr int B = decltype(bufc)::value;
    using Bc = std::integral_constant<int, B>;
    using Nc = std::integral_constant<int, B ^ 1>;
    stage16(livec, 0, aP, wQ, [&](int g) { if ((g & 3) == 1) ld_a(aR[g >> 2], Bc{}, 0, 4 + (g >> 2)); });
    stage16(livec, 4, aR, wQ, [&](int g) {
      if ((g & 3) == 0) ld_a(aP[g >> 2], Bc{}, 1, g >> 2);
      if ((g & 3) == 2) ld_w(wS[g >> 2], Bc{}, 1, g >> 2);
    });
    stage16(livec, 0, aP, wS, [&](int g) { if ((g & 3) == 1) ld_a(aR[g >> 2], Bc{}, 1, 4 + (g >> 2)); });
    __builtin_amdgcn_s_waitcnt(0xC07F);                              // lgkmcnt(0): this wave's last reads of K-tile kt are in registers
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // its pieces of K-tile kt + 1 have landed
    SPF(); __builtin_amdgcn_s_barrier(); SPF();
    constexpr bool more = decltype(morec)::value, stage2 = decltype(stage2c)::value;
    stage16(livec, 4, aR, wS, [&](int g) {
      if constexpr (stage2) {
        if (g == 0) stage_piece(src, Bc{}, I0{}, kt + 2, false);
        if (g == 2) stage_piece(src, Bc{}, I0{}, kt + 2, true);
        if (g == 4) stage_piece(src, Bc{}, I1{}, kt + 2, false);
        if (g == 6) stage_piece(src, Bc{}, I1{}, kt + 2, true);
        if (g == 8) stage_piece(src, Bc{}, I2{}, kt + 2, false);
        if (g == 10) stage_piece(src, Bc{}, I2{}, kt + 2, true);
        if (g == 12) stage_piece(src, Bc{}, I3{}, kt + 2, false);
        if (g == 14) stage_piece(src, Bc{}, I3{}, kt + 2, true);
      }
      if constexpr (more) {
        if ((g & 3) == 1) ld_a(aP[g >> 2], Nc{}, 0, g >> 2);
        if ((g & 3) == 3) ld_w(wQ[g >> 2], Nc{}, 0, g >> 2);
      }
    });
  };

  bool k0_staged = false;
  for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
    int m0, n0;
    tile_origin(v, m0, n0);
    const Src src = tile_src(m0, n0);
    wave_live = n0 + wc * 64 < p.N;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // ---- prologue: K-tile 0 landed, K-tile 1 in flight; first fragments in registers ----
    if (!k0_staged) stage_ktile(src, I0{}, 0);
    if (nk > 1) {
      stage_ktile(src, I1{}, 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    SPF(); __builtin_amdgcn_s_barrier(); SPF();
#pragma unroll
    for (int b = 0; b < 4; ++b) { ld_a(aP[b], I0{}, 0, b); ld_w(wQ[b], I0{}, 0, b); }
    using T = std::true_type; using F = std::false_type;
    auto kloop = [&](auto livec) {                                  // nk is even and >= 4 (launcher)
      int kt = 0;
      for (; kt + 2 < nk; kt += 2) {
        ktile(I0{}, src, kt, livec, T{}, T{});
        ktile(I1{}, src, kt + 1, livec, T{}, T{});
      }
      ktile(I0{}, src, kt, livec, F{}, T{});
      ktile(I1{}, src, kt + 1, livec, F{}, F{});
    };
    if (wave_live) kloop(T{}); else kloop(F{});
    // (every LDS read of this tile completed before its last barrier: the epilogue may overwrite the buffers)
    const int vn = v + gridDim.x;
    bool hooked = false;
    Src nsrc = src;
    k0_staged = false;
    if (PREFETCH && vn < ntiles) {
      int m1, n1;
      tile_origin(vn, m1, n1);
      nsrc = tile_src(m1, n1);
      k0_staged = true;
    }
    auto hook = [&]() {
      if (!hooked && k0_staged) stage_ktile(nsrc, I0{}, 0);
      hooked = true;
    };
    if constexpr (EPI == EPI_QKV) qkv_epilogue16<4>(p, acc, smem + EPI_OFF, wave, lane, m0 + wr * 128, n0 + wc * 64, hook);
    else gemm_epilogue16<4, EPI, 4, SWAPACC>(p, acc, 0, smem + EPI_OFF, wave, lane, m0 + wr * 128, n0 + wc * 64, hook);
    hook();
    if (vn < ntiles) __syncthreads();
  }
#undef SPF
}
#endif  // LD_VARIANTS

// ------------------------------------------------------------------------------------------------
// fp8 (OCP e4m3) x fp8 -> fp32 GEMM for the DiT's four large linear layers (BASELINE config 5; never the headline
// metric, which is bf16).  Same 256x256 tile / 8 waves (2 x 4, 128x64 per wave) / two-stage LDS-DMA structure as
// ld_gemm_kernel: a K-tile is again 128 BYTES per row -- now 128 elements -- so the DMA pieces, the XOR swizzle and the
// LDS footprint are unchanged while every tile carries twice the K.  v_mfma_scale_f32_32x32x64_f8f6f4 (unit scales)
// takes 32 bytes per lane per operand: row = lane % 32; lanes 0-31 hold k 0-15 and 32-47 of the 64-deep step, lanes 32-63
// hold k 16-31 and 48-63 (tools/probe/fp8_mfma_layout.hip, fp8_mfma_scale.hip), i.e. two 16-byte chunks of the tile row.
// The accumulator is dequantised in registers -- acc * scale_a[row] * scale_w[col] -- and then takes the ordinary
// epilogues (bias / GELU / gated residual).
// ------------------------------------------------------------------------------------------------

// MX = true: MXFP8 operands -- the per-32-element E8M0 scales go into the MFMA itself (one byte per lane and operand: the
// lane's row and its 32-deep half of the 64-deep step), fetched as one dword per row and 128-deep K-tile straight into
// registers one tile ahead; no dequantisation in the epilogue.
template <int EPI, bool MX>
__global__ __launch_bounds__(512, 2) void ld_gemm_f8_kernel(GemmParams p) {
  constexpr int BM = 256, BN = 256, WN = 4, NW = 8, MI = 4, NI = 2;
  constexpr int KB = 128;                                    // bytes (= elements) of K per tile
  constexpr int A_BYTES = BM * KB, B_BYTES = BN * KB, STAGE = A_BYTES + B_BYTES;
  constexpr int A_LOADS = BM / 8 / NW, B_LOADS = BN / 8 / NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, nbm * nbn);
  const int gm_sz = p.group_m;
  const int per_group = gm_sz * nbn;
  const int group = bid / per_group, in_group = bid - group * per_group;
  const int first_m = group * gm_sz;
  const int rows_here = (nbm - first_m) < gm_sz ? (nbm - first_m) : gm_sz;
  const int m0 = (first_m + in_group % rows_here) * BM, n0 = (in_group / rows_here) * BN;
  // LDS-DMA through raw buffer descriptors based at the tile origin (rows past M / N read as zeros, no clamping): the
  // wave-uniform part of every address -- K-tile, 16-row step between a wave's pieces -- is the scalar offset, the per-lane
  // part is ONE 32-bit offset per piece parity (the source-side swizzle key (row >> 1) & 7 repeats every 16 rows).
  const long ldab = p.lda, ldwb = p.K;
  const auto clip = [](long v) { return (int)(v < 0x7fffffffL ? v : 0x7fffffffL); };
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.A + (long)m0 * ldab), 0, clip((long)(p.M - m0) * ldab), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.W + (long)n0 * ldwb), 0, clip((long)(p.N - n0) * ldwb), 0x00020000);
  uint32_t voA[2], voW[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int ra = (wave * A_LOADS + par) * 8 + (lane >> 3), rb = (wave * B_LOADS + par) * 8 + (lane >> 3);
    voA[par] = (uint32_t)(ra * ldab + (((lane & 7) ^ ((ra >> 1) & 7)) << 4));
    voW[par] = (uint32_t)(rb * ldwb + (((lane & 7) ^ ((rb >> 1) & 7)) << 4));
  }
  const int sa16 = (int)(16 * ldab), sw16 = (int)(16 * ldwb);
  const int nk = p.K / KB;
  // MX scales: one dword (4 blocks = one K-tile) per tile row, staged through LDS next to the operands -- waves 0-3 fetch
  // the 256 A rows' dwords, waves 4-7 the 256 W rows' (one 4-byte LDS-DMA each) -- and read back at use (no registers held)
  constexpr int SC_OFF = 2 * STAGE;                          // [2 stages][A 1 KB | W 1 KB]
  const long srows = wave < 4 ? p.M : p.N;                    // rows per K-tile slab of the scale array
  const long sorig = wave < 4 ? m0 : n0;
  const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(MX ? (wave < 4 ? p.mx_a : p.mx_w) + sorig * 4 : (const unsigned char*)p.A), 0,
      MX ? clip(((long)(p.K >> 7) * srows - sorig) * 4) : 0, 0x00020000);
  const uint32_t soff = (uint32_t)(((wave & 3) * 64 + lane) * 4);
  const int sslab = (int)(srows * 4);                         // bytes between consecutive K-tiles' slabs
  auto stage = [&](auto bufc, int kt) {
    constexpr int buf = decltype(bufc)::value;
    if (MX) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsS, (__attribute__((address_space(3))) void*)(smem + SC_OFF + buf * 2048 + wave * 256),
                                               4, soff, kt * sslab, 0, 0);
    }
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(base + (wave * A_LOADS + i) * 1024), 16,
                                               voA[i & 1], kt * KB + (i >> 1) * sa16, 0, 0);
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(base + A_BYTES + (wave * B_LOADS + i) * 1024), 16,
                                               voW[i & 1], kt * KB + (i >> 1) * sw16, 0, 0);
  };

  f32x16_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int rdA[2][2], rdB[2][2];                                  // [64-deep step][16-byte half]
  {
    const int ra = wr * 128 + (lane & 31), rb = wc * 64 + (lane & 31);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // register half h of lane-half g holds k = 64 kk + 32 h + 16 g .. +16: that is the hardware's K order (it matters
        // once the two 32-element blocks of a step carry different scales; tools/probe/fp8_mfma_scale.hip)
        const int c = kk * 4 + h * 2 + (lane >> 5);
        rdA[kk][h] = ra * 128 + ((c ^ ((ra >> 1) & 7)) << 4);
        rdB[kk][h] = A_BYTES + rb * 128 + ((c ^ ((rb >> 1) & 7)) << 4);
      }
  }
  auto ldfrag = [&](int off) {
    const u32x4_t lo = *(const u32x4_t*)(smem + off);
    return lo;
  };
  auto frag32 = [&](int off0, int off1) {
    const u32x4_t lo = ldfrag(off0), hi = ldfrag(off1);
    return (i32x8_t){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
  };
  // One K-tile: per 64-deep step the two W fragments stay live, the A fragments stream through a two-deep register
  // pipeline (fragment i+1 is requested before the MFMAs of fragment i) -- 32 fragment registers instead of 48.
  auto compute = [&](auto bufc) {
    constexpr int OFF = decltype(bufc)::value * STAGE;
    uint32_t sb[NI];
    const char* sc = smem + SC_OFF + decltype(bufc)::value * 2048;
    if (MX) {
#pragma unroll
      for (int j = 0; j < NI; ++j) sb[j] = *(const uint32_t*)(sc + 1024 + (wc * 64 + j * 32 + (lane & 31)) * 4) >> ((lane >> 5) * 8);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      i32x8_t b[NI], a[2];
#pragma unroll
      for (int j = 0; j < NI; ++j) b[j] = frag32(rdB[kk][0] + OFF + j * 4096, rdB[kk][1] + OFF + j * 4096);
      a[0] = frag32(rdA[kk][0] + OFF, rdA[kk][1] + OFF);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        if (i + 1 < MI) a[(i + 1) & 1] = frag32(rdA[kk][0] + OFF + (i + 1) * 4096, rdA[kk][1] + OFF + (i + 1) * 4096);
        uint32_t sa = 0;
        if (MX) sa = *(const uint32_t*)(sc + (wr * 128 + i * 32 + (lane & 31)) * 4) >> ((lane >> 5) * 8);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          if constexpr (MX) {
            // after the >> (8 * half), byte 0 / byte 2 of the register is this lane's block of step 0 / step 1
            if (kk == 0) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i & 1], b[j], acc[i][j], 0, 0, 0, sa, 0, sb[j]);
            else acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i & 1], b[j], acc[i][j], 0, 0, 2, sa, 2, sb[j]);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i & 1], b[j], acc[i][j], 0, 0, 0, 127, 0, 127);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  stage(B0{}, 0);
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    __syncthreads();
    stage(B1{}, kt + 1);
    compute(std::integral_constant<int, 0>{});
    __syncthreads();
    if (kt + 2 < nk) stage(B0{}, kt + 2);
    compute(std::integral_constant<int, 1>{});
  }
  if (kt < nk) {
    __syncthreads();
    compute(std::integral_constant<int, 0>{});
  }
  __syncthreads();

  // dequantise: acc[i][j][r] is C[row0 + 32 i + 8 (r / 4) + 4 (lane / 32) + r % 4][col0 + 32 j + lane % 32]
  const int row0 = m0 + wr * 128, col0 = n0 + wc * 64;
  if constexpr (!MX) {
    float sw[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) { const int gn = col0 + j * 32 + (lane & 31); sw[j] = p.scale_w[gn < p.N ? gn : p.N - 1]; }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gm = row0 + i * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
        const float sa = p.scale_a[gm < p.M ? gm : p.M - 1];
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j][r] *= sa * sw[j];
      }
  }
  gemm_epilogue<MI, NI, EPI>(p, acc, smem, wave, lane, row0, col0);
}

#ifdef LD_VARIANTS
// ------------------------------------------------------------------------------------------------
// One-wave-per-SIMD, register-staged main loop on v_mfma_f32_32x32x16_bf16 (the round-1 default, LD_GEMM_TILE=11; see
// profiles/r01d_gemm_vs_vendor_library.txt for the measurements that shaped it): 256x256 tile, 4 waves x 128x128 = 4x4
// accumulators of 16 registers (256 of the wave's 512 registers), K-tiles 64 deep on full 128-byte lines.  Global memory -> VGPRs by buffer_load_dwordx4 (row offsets in SGPRs, out-of-range rows read as zero
// through the buffer descriptor's bounds check), two register sets = prefetch three K-tiles ahead; VGPRs -> LDS by
// ds_write_b128 one tile ahead into a two-slot ring (2 x 64 KB, XOR-swizzled 16-byte chunks as in ld_gemm_kernel).
//   tile t:  k-steps 0,1: 16 MFMA each + ds_write of K-tile t+1 (8 per k-step)     [its slot was last read in tile t-1]
//            k-step  2  : 16 MFMA + first half of the loads of K-tile t+3
//            lgkmcnt(0) + barrier: K-tile t+1 visible everywhere, and nobody reads slot t&1 past k-step 3's registers
//            k-step  3  : 16 MFMA + second half of the loads; fragment prefetch of (t+1, 0)
template <int EPI>
__global__ __launch_bounds__(256, 1) void ld_gemm_w4r_kernel(GemmParams p) {
  constexpr int BM = 256, BN = 256, KT = 64;
  constexpr int A_BYTES = BM * KT * 2;              // 32 KB
  constexpr int SLOT = (BM + BN) * KT * 2;          // 64 KB
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  const int nbm = (p.M - p.m_begin + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, nbm * nbn);
  const int gm_sz = p.group_m;
  const int per_group = gm_sz * nbn;
  const int group = bid / per_group, in_group = bid - group * per_group;
  const int first_m = group * gm_sz;
  const int rows_here = (nbm - first_m) < gm_sz ? (nbm - first_m) : gm_sz;
  const int m0 = p.m_begin + (first_m + in_group % rows_here) * BM, n0 = (in_group / rows_here) * BN;

  // this wave stages rows [wave*64, wave*64+64) of the A tile and of the W tile: 8 + 8 loads of 8 rows x 128 B per K-tile
  const long ldab = p.lda * 2, ldwb = (long)p.K * 2;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.A + (long)m0 * ldab), 0,
      (int)(((long)(p.M - m0) * ldab) < 0x7fffffffL ? ((long)(p.M - m0) * ldab) : 0x7fffffffL), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.W + (long)n0 * ldwb), 0,
      (int)(((long)(p.N - n0) * ldwb) < 0x7fffffffL ? ((long)(p.N - n0) * ldwb) : 0x7fffffffL), 0x00020000);
  const int rl = lane >> 3, cl = lane & 7;                         // row within the 8-row group, 16-byte chunk
  const uint32_t voA = (uint32_t)((wave * 64 + rl) * ldab + cl * 16);
  const uint32_t voW = (uint32_t)((wave * 64 + rl) * ldwb + cl * 16);
  const int sa8 = (int)(8 * ldab), sw8 = (int)(8 * ldwb);          // SGPR step between a wave's 8-row groups
  // LDS write addresses: row = wave*64 + i*8 + rl, chunk cl ^ ((row >> 1) & 7); (row>>1)&7 alternates with i's parity
  int wrofs[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int row = wave * 64 + par * 8 + rl;
    wrofs[par] = row * 128 + ((cl ^ ((row >> 1) & 7)) << 4);
  }
  const int nk = p.K / KT;

  u32x4_t st[2][16];           // two staging sets: K-tile tau lives in set tau & 1 (loads 0-7: A groups, 8-15: W groups)
  auto LOAD = [&](auto setc, int q, int kt) {
    constexpr int U = decltype(setc)::value;
    if (q < 8) st[U][q] = __builtin_amdgcn_raw_buffer_load_b128(rsA, voA, kt * (KT * 2) + q * sa8, 0);
    else st[U][q] = __builtin_amdgcn_raw_buffer_load_b128(rsW, voW, kt * (KT * 2) + (q - 8) * sw8, 0);
  };
  auto WRITE = [&](auto setc, int slot, int q) {
    constexpr int U = decltype(setc)::value;
    const int i = q & 7;
    char* dst = smem + slot * SLOT + (q < 8 ? 0 : A_BYTES) + wrofs[i & 1] + (i >> 1) * 2048;
    *(u32x4_t*)dst = st[U][q];
  };

  f32x16_t acc[2][4][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[h][i][j][r] = 0.f;

  int rdA[4], rdB[4];          // fragment read offsets inside a slot for the four k-steps (+4096 B per further 32 rows)
  {
    const int ra = wr * 128 + (lane & 31), rb = wc * 128 + (lane & 31);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int c = kk * 2 + (lane >> 5);
      rdA[kk] = ra * 128 + ((c ^ ((ra >> 1) & 7)) << 4);
      rdB[kk] = A_BYTES + rb * 128 + ((c ^ ((rb >> 1) & 7)) << 4);
    }
  }
  bf16x8_t fa[2][4], fb[2][4];
  auto FRAG = [&](auto bufc, int slot, int kk, int g) {
    constexpr int B = decltype(bufc)::value;
    if (g < 4) fa[B][g] = *(const bf16x8_t*)(smem + rdA[kk] + slot * SLOT + g * 4096);
    else fb[B][g - 4] = *(const bf16x8_t*)(smem + rdB[kk] + slot * SLOT + (g - 4) * 4096);
  };
#define FENCE() __builtin_amdgcn_sched_barrier(0)
  // one k-step: 16 MFMAs on fragment set B, the 8 fragment reads of the next k-step behind the first four pairs, and one
  // staging operation per pair: MODE 1 = ds_write of set U pieces q0..q0+7, MODE 2 = loads of K-tile lkt into set U
  auto kstep = [&](auto bufc, int nslot, int nkk, auto modec, auto setc, int q0, int wslot, int lkt) {
    constexpr int B = decltype(bufc)::value;
    constexpr int MODE = decltype(modec)::value;
    using NB = std::integral_constant<int, 1 - B>;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const int i = g >> 1, j0 = (g & 1) * 2;
      acc[j0 >> 1][i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[B][i], fb[B][j0], acc[j0 >> 1][i][0], 0, 0, 0);
      if (g < 4) FRAG(NB{}, nslot, nkk, 2 * g);
      acc[j0 >> 1][i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[B][i], fb[B][j0 + 1], acc[j0 >> 1][i][1], 0, 0, 0);
      if (g < 4) FRAG(NB{}, nslot, nkk, 2 * g + 1);
      if (MODE == 1) WRITE(setc, wslot, q0 + g);
      if (MODE == 2) LOAD(setc, q0 + g, lkt);
      FENCE();
    }
  };
  using B0 = std::integral_constant<int, 0>; using B1 = std::integral_constant<int, 1>;
  using M0 = std::integral_constant<int, 0>; using M1 = std::integral_constant<int, 1>; using M2 = std::integral_constant<int, 2>;

  // ---- prologue: K-tiles 0, 1, 2 requested; K-tile 0 -> slot 0; fragments of (0,0) ----
#pragma unroll
  for (int q = 0; q < 16; ++q) LOAD(B0{}, q, 0);
#pragma unroll
  for (int q = 0; q < 16; ++q) LOAD(B1{}, q, 1 < nk ? 1 : 0);
#pragma unroll
  for (int q = 0; q < 16; ++q) WRITE(B0{}, 0, q);
#pragma unroll
  for (int q = 0; q < 16; ++q) LOAD(B0{}, q, 2 < nk ? 2 : 0);
  __builtin_amdgcn_s_waitcnt(0xc07f);         // lgkmcnt(0) only (vmcnt / expcnt fields at "no wait")
  __builtin_amdgcn_s_barrier();
  FENCE();
#pragma unroll
  for (int g = 0; g < 8; ++g) FRAG(B0{}, 0, 0, g);

  // tile t in slot S = t & 1; set U = (t + 1) & 1 holds K-tile t+1 on entry and receives K-tile t+3
  auto tile = [&](auto slotc, int t) {
    constexpr int S = decltype(slotc)::value;
    using U = std::integral_constant<int, 1 - S>;
    const int lkt = t + 3 < nk ? t + 3 : nk - 1;      // past the end: a re-fetch that is never multiplied
    kstep(B0{}, S, 1, M1{}, U{}, 0, 1 - S, 0);        // k-step 0: fragments of (t,1); ds_write pieces 0-7 of K-tile t+1
    kstep(B1{}, S, 2, M1{}, U{}, 8, 1 - S, 0);        // k-step 1: fragments of (t,2); ds_write pieces 8-15
    kstep(B0{}, S, 3, M2{}, U{}, 0, 0, lkt);          // k-step 2: fragments of (t,3); loads 0-7 of K-tile t+3
    __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): this wave's ds_writes and fragment reads retired
    __builtin_amdgcn_s_barrier();
    FENCE();
    kstep(B1{}, 1 - S, 0, M2{}, U{}, 8, 0, lkt);      // k-step 3: fragments of (t+1,0); loads 8-15
  };
  for (int t = 0; t < nk; t += 2) {
    tile(std::integral_constant<int, 0>{}, t);
    tile(std::integral_constant<int, 1>{}, t + 1);
  }
#undef FENCE
  __syncthreads();

  gemm_epilogue<4, 2, EPI>(p, acc[0], smem, wave, lane, m0 + wr * 128, n0 + wc * 128);
  gemm_epilogue<4, 2, EPI>(p, acc[1], smem, wave, lane, m0 + wr * 128, n0 + wc * 128 + 64);
}
#endif  // LD_VARIANTS

template <auto Kernel>
int launch_kernel(const char* what, dim3 grid, dim3 block, int smem, hipStream_t stream, const GemmParams& p) {
  static thread_local LdSmemCache cache{};      // per kernel instantiation (and per host thread, per device inside)
  if (int rc = ld_ensure_dyn_smem((const void*)Kernel, (size_t)smem, &cache)) return rc;
  hipLaunchKernelGGL(Kernel, grid, block, smem, stream, p);
  return ld_check_launch(what);
}

template <int BM, int BN, int WM, int WN, int NSTAGE>
int launch_cfg(const GemmParams& p, bool conv, hipStream_t stream) {
  constexpr int NW = WM * WN;
  constexpr int STAGE = (BM + BN) * BK * 2;
  constexpr int EPIB = NW * 32 * CW_STRIDE * 4;
  constexpr int SMEM = (NSTAGE * STAGE > EPIB) ? NSTAGE * STAGE : EPIB;
  constexpr int SMEM_QKV = (SMEM > NW * QKV_REGION) ? SMEM : NW * QKV_REGION;
  const int nbm = (p.M - p.m_begin + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  dim3 grid(nbm * nbn), block(NW * 64);
  const int epi = pick_epilogue(p);
  // GroupNorm partials are summed per 64-row unit = two 32-row blocks of a wave tile (gemm_epilogue_core<GN>): a tile whose wave
  // rows are an odd number of blocks compiles the sums out, and the caller's buffer would stay unwritten
  LD_REQUIRE(!p.gn_part || (conv && (BM / WM / 32) % 2 == 0), "ld_conv_cl_bf16_gn: the %d x %d tile route cannot write GroupNorm partials", BM, BN);
  if (epi == EPI_QKV) {
    LD_REQUIRE(!conv, "ld_gemm_qkv_heads: not a convolution epilogue");
    return launch_kernel<ld_gemm_kernel<BM, BN, WM, WN, NSTAGE, false, EPI_QKV, true>>("ld_gemm_qkv_heads", grid, block, SMEM_QKV, stream, p);
  }
  // 16x16x32 MFMAs by default (LD_GEMM_M16=0: the 32x32x16 form, kept for A/B measurements): +7...11 % on the DiT shapes
  static int m16 = -1;
  if (m16 < 0) { const char* e = getenv("LD_GEMM_M16"); m16 = e ? atoi(e) : 1; }
#define LD_GEMM_LAUNCH16(CONV_, EPI_) \
  return launch_kernel<ld_gemm_kernel<BM, BN, WM, WN, NSTAGE, CONV_, EPI_, true>>("ld_gemm16", grid, block, SMEM, stream, p)
  if (m16) {
    if (conv) {
      if (epi == EPI_BIAS) LD_GEMM_LAUNCH16(true, EPI_BIAS);
      LD_GEMM_LAUNCH16(true, EPI_GENERIC);
    }
    switch (epi) {
      case EPI_BIAS: LD_GEMM_LAUNCH16(false, EPI_BIAS);
      case EPI_GELU: LD_GEMM_LAUNCH16(false, EPI_GELU);
      case EPI_GATE: LD_GEMM_LAUNCH16(false, EPI_GATE);
      default: LD_GEMM_LAUNCH16(false, EPI_GENERIC);
    }
  }
#undef LD_GEMM_LAUNCH16
#define LD_GEMM_LAUNCH(CONV_, EPI_) \
  return launch_kernel<ld_gemm_kernel<BM, BN, WM, WN, NSTAGE, CONV_, EPI_>>("ld_gemm", grid, block, SMEM, stream, p)
  if (conv) {
    if (epi == EPI_BIAS) LD_GEMM_LAUNCH(true, EPI_BIAS);
    LD_GEMM_LAUNCH(true, EPI_GENERIC);
  }
  switch (epi) {
    case EPI_BIAS: LD_GEMM_LAUNCH(false, EPI_BIAS);
    case EPI_GELU: LD_GEMM_LAUNCH(false, EPI_GELU);
    case EPI_GATE: LD_GEMM_LAUNCH(false, EPI_GATE);
    default: LD_GEMM_LAUNCH(false, EPI_GENERIC);
  }
#undef LD_GEMM_LAUNCH
}

int launch_8p(const GemmParams& p, bool conv, hipStream_t stream) {
  constexpr int SMEM = LD_LDS_TOTAL;                       // 2 x 64 KB K-tile buffers; epilogue staging at the end of the 160 KB
  const int nbm = (p.M - p.m_begin + 255) / 256, nbn = (p.N + 255) / 256;
  const long ntiles = (p.tile_end > 0 ? p.tile_end : (long)nbm * nbn) - p.tile_begin;     // tiles of THIS launch
  // persistent tiles: one workgroup per CU (LD_GEMM_PERSIST=0: one workgroup per tile)
  static int persist = -1, ncu = 0;
  if (persist < 0) {
    const char* e = getenv("LD_GEMM_PERSIST"); persist = e ? atoi(e) : 1;
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    if (ncu <= 0 || (ncu & 7)) ncu = 256;
  }
  dim3 grid((unsigned)((persist && ntiles > ncu) ? ncu : ntiles)), block(512);
  const int epi = pick_epilogue(p);
#ifdef LD_VARIANTS
  // LD_GEMM_SP=1: the software-pipelined loop (with LD_TUNING=1 re-read per call: tools/gemm_ab.py times both loops alternately)
  static int k_sp = LD_KNOB_UNSET;
  if (!conv && ld_knob("LD_GEMM_SP", 0, &k_sp) == 1 && (p.K / BK) % 2 == 0 && p.K / BK >= 4) {
    switch (epi) {
      case EPI_QKV: return launch_kernel<ld_gemm_sp_kernel<EPI_QKV>>("ld_gemm_qkv_heads(sp)", grid, block, SMEM, stream, p);
      case EPI_BIAS: return launch_kernel<ld_gemm_sp_kernel<EPI_BIAS>>("ld_gemm_sp", grid, block, SMEM, stream, p);
      case EPI_GELU: return launch_kernel<ld_gemm_sp_kernel<EPI_GELU>>("ld_gemm_sp", grid, block, SMEM, stream, p);
      case EPI_GATE: return launch_kernel<ld_gemm_sp_kernel<EPI_GATE>>("ld_gemm_sp", grid, block, SMEM, stream, p);
      default: return launch_kernel<ld_gemm_sp_kernel<EPI_GENERIC>>("ld_gemm_sp", grid, block, SMEM, stream, p);
    }
  }
#endif
  if (epi == EPI_QKV) {
    LD_REQUIRE(!conv, "ld_gemm_qkv_heads: not a convolution epilogue");
    return launch_kernel<ld_gemm8p_kernel<false, EPI_QKV>>("ld_gemm_qkv_heads", grid, block, SMEM, stream, p);
  }
  if (conv) {
    if (epi == EPI_BIAS) return launch_kernel<ld_gemm8p_kernel<true, EPI_BIAS>>("ld_gemm8p", grid, block, SMEM, stream, p);
    return launch_kernel<ld_gemm8p_kernel<true, EPI_GENERIC>>("ld_gemm8p", grid, block, SMEM, stream, p);
  }
  switch (epi) {
    case EPI_BIAS: return launch_kernel<ld_gemm8p_kernel<false, EPI_BIAS>>("ld_gemm8p", grid, block, SMEM, stream, p);
    case EPI_GELU: return launch_kernel<ld_gemm8p_kernel<false, EPI_GELU>>("ld_gemm8p", grid, block, SMEM, stream, p);
    case EPI_GATE: return launch_kernel<ld_gemm8p_kernel<false, EPI_GATE>>("ld_gemm8p", grid, block, SMEM, stream, p);
    default: return launch_kernel<ld_gemm8p_kernel<false, EPI_GENERIC>>("ld_gemm8p", grid, block, SMEM, stream, p);
  }
}

#ifdef LD_VARIANTS
// 512 x 128 tiles for convolutions with a 128-column output (ld_gemm8p_m512_kernel), persistent like launch_8p
int launch_8p_m512(const GemmParams& p, hipStream_t stream) {
  const long ntiles = (long)((p.M + 511) / 512) * ((p.N + 127) / 128);
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    if (ncu <= 0 || (ncu & 7)) ncu = 256;
  }
  dim3 grid((unsigned)(ntiles > ncu ? ncu : ntiles)), block(512);
  if (pick_epilogue(p) == EPI_BIAS) return launch_kernel<ld_gemm8p_m512_kernel<true, EPI_BIAS>>("ld_gemm8p_m512", grid, block, LD_LDS_TOTAL, stream, p);
  return launch_kernel<ld_gemm8p_m512_kernel<true, EPI_GENERIC>>("ld_gemm8p_m512", grid, block, LD_LDS_TOTAL, stream, p);
}
#endif  // LD_VARIANTS

// the partial last round of a launch as 256 x 128 half tiles, one per workgroup (ld_gemm8p_n128_kernel)
int launch_8p_n128(const GemmParams& p, hipStream_t stream) {
  const int nbm = (p.M - p.m_begin + 255) / 256, nbn = (p.N + 255) / 256;
  const int v_end = p.tile_end > 0 ? p.tile_end : nbm * nbn;
  dim3 grid((unsigned)(2 * (v_end - p.tile_begin))), block(512);
  constexpr int SMEM = LD_LDS_TOTAL;
  switch (pick_epilogue(p)) {
    case EPI_QKV: return launch_kernel<ld_gemm8p_n128_kernel<EPI_QKV>>("ld_gemm_qkv_heads(half tiles)", grid, block, SMEM, stream, p);
    case EPI_BIAS: return launch_kernel<ld_gemm8p_n128_kernel<EPI_BIAS>>("ld_gemm8p_n128", grid, block, SMEM, stream, p);
    case EPI_GELU: return launch_kernel<ld_gemm8p_n128_kernel<EPI_GELU>>("ld_gemm8p_n128", grid, block, SMEM, stream, p);
    case EPI_GATE: return launch_kernel<ld_gemm8p_n128_kernel<EPI_GATE>>("ld_gemm8p_n128", grid, block, SMEM, stream, p);
    default: return launch_kernel<ld_gemm8p_n128_kernel<EPI_GENERIC>>("ld_gemm8p_n128", grid, block, SMEM, stream, p);
  }
}

#ifdef LD_VARIANTS
int launch_w4r(const GemmParams& p, hipStream_t stream) {
  constexpr int SMEM = 2 * (256 + 256) * 64 * 2;   // two 64 KB K-tile slots (the epilogue staging reuses them)
  const int nbm = (p.M - p.m_begin + 255) / 256, nbn = (p.N + 255) / 256;
  dim3 grid(nbm * nbn), block(256);
  switch (pick_epilogue(p)) {
    case EPI_BIAS: return launch_kernel<ld_gemm_w4r_kernel<EPI_BIAS>>("ld_gemm_w4r", grid, block, SMEM, stream, p);
    case EPI_GELU: return launch_kernel<ld_gemm_w4r_kernel<EPI_GELU>>("ld_gemm_w4r", grid, block, SMEM, stream, p);
    case EPI_GATE: return launch_kernel<ld_gemm_w4r_kernel<EPI_GATE>>("ld_gemm_w4r", grid, block, SMEM, stream, p);
    default: return launch_kernel<ld_gemm_w4r_kernel<EPI_GENERIC>>("ld_gemm_w4r", grid, block, SMEM, stream, p);
  }
}
#endif  // LD_VARIANTS

// Bytes of the zero-bordered channels-last input a convolution's A-address generator walks: [(T + kT - 1)][Hp][Wp][Cin] bf16.
long conv_input_bytes(const GemmParams& p) {
  const long kT = p.K / ((long)p.kH * p.kW * p.Cin);
  const long T = p.M / ((long)p.H * p.W_);
  return (T + kT - 1) * p.Hp * p.Wp * p.Cin * 2;
}
// The 8-phase kernel addresses a convolution input through ONE raw buffer descriptor based at the tensor (num_records 2^31 - 1,
// 32-bit per-lane BYTE offsets): inputs of 2 GiB or more are out of its range and take the two-stage kernel, whose 32-bit
// ELEMENT offsets + 64-bit tap offsets reach 8 GiB (ld_conv_cl_bf16 refuses anything larger).
constexpr long CONV_8P_MAX_BYTES = 0x7fffffffL;
constexpr long CONV_MAX_BYTES = 1L << 33;

enum { ROUTE_128_2STAGE = 0, ROUTE_256_2STAGE = 1, ROUTE_256_8PHASE = 2, ROUTE_256_W4R = 3, ROUTE_512_8PHASE = 4, ROUTE_NARROW = 5 };   // (documented at ld_conv_route in landiff_hip.h)
thread_local int g_last_route = -1;   // what launch() picked last ON THIS HOST THREAD (ld_conv_route's dry run reads it; the
                                      // pipeline runs launches from a helper thread too)

int launch(const GemmParams& p, bool conv, hipStream_t stream, bool dry_run = false) {
  // LD_GEMM_TILE (tuning knob): 1 = 128x128 / 4 waves, 3 = 256x256 / 8 waves (both 2-stage, barrier-drained, 16x16x32 MFMAs
  // unless LD_GEMM_M16=0; 3 is the default for large problems when the knob is unset), 11 = 4 waves x 128x128 register-staged
  // on 32x32x16 MFMAs (the round-1 default, kept as the measured alternative; its two siblings -- an 8-wave load/compute
  // ping-pong and a 4-wave LDS-DMA pipeline, within 2 % of it -- were removed in round 2)
  static int forced = -1, group_m = 8;
  if (forced < 0) {
    const char* e = getenv("LD_GEMM_TILE"); forced = e ? atoi(e) : 0;
    const char* g = getenv("LD_GEMM_GROUP_M"); if (g && atoi(g) > 0) group_m = atoi(g);
  }
  // raster group height: 8 x 4 tile patches per XCD (32 resident tiles) for wide outputs; narrow ones (the DiT's N = 1920 GEMMs:
  // 8 tile columns) do better with 4 rows x all 8 columns -- the whole W panel set stays in the XCD's L2 (ff2 1311 -> 1344 TFLOP/s)
  static bool group_forced = getenv("LD_GEMM_GROUP_M") != nullptr;
  const_cast<GemmParams&>(p).group_m = (!group_forced && (p.N + 255) / 256 <= 8) ? 4 : group_m;
  if (p.gn_part) {      // (every conv route's epilogue is EPI_BIAS or EPI_GENERIC with 64-row units: the forms that sum partials)
    const int e = pick_epilogue(p);
    LD_REQUIRE(conv && (e == EPI_BIAS || e == EPI_GENERIC), "ld_conv_cl_bf16_gn: epilogue %d does not sum GroupNorm partials", e);
  }
  // measured on MI355X (tools/gemm_dit_shapes.py): the 256x256 tile wins once the grid fills the chip twice over
  // (L2->LDS traffic halves); the 128x128 tile (2 workgroups/CU) is for small problems.  A 128x256-tile, two-workgroups-
  // per-CU form of the pipelined loop (epilogue of one workgroup under the main loop of the other) measured 10 % slower
  // than the 256x256 tile on all four DiT shapes and was dropped.
  int cfg = forced;
  if (cfg == 0) {
    const long tiles256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
    // (the DiT's 1920x1920 GEMMs: 256x256 tiles 0.26 ms vs 0.29 ms on 128x128; the VAE's narrow convolutions, Cout <= 512,
    //  stay on 128x128 tiles unless K is long: 0.50 vs 0.58 s per video.  Round 5, per shape (tools/conv_shape_time.py,
    //  profiles/r05_vae_conv_route_ab.txt): the 256-wide tile must be at least 3/4 used -- Cin 256 -> Cout 128 at 480 x 720 ran
    //  with half of its waves dead, 5.82 ms against 4.41 on 128 x 128 tiles -- and K >= 2048 is long enough: the 1 x 3 x 3
    //  upsampler convs, K = 2304, 3.15 -> 2.69 ms)
    const bool wide_enough = 4L * p.N >= 3L * 256 * ((p.N + 255) / 256);
    cfg = (tiles256 >= 512 && (conv ? (wide_enough && p.K >= 2048) : p.K >= 1024)) ? 3 : 1;
  }
  const bool pp_ok = (p.K % 128 == 0) && (!conv || p.Cin % 32 == 0);
#ifdef LD_VARIANTS
  // LD_GEMM_M512=1 (variants build): a convolution with one 128-wide column of output and 2048 <= K <= 4096 (the VAE's 480 x 720
  // level) on 512 x 128 tiles of the 8-phase loop, when they fill the chip at least once.  Bit-identical; faster alone (2.37 ->
  // 2.16 ms), no gain inside the VAE decode: see ld_gemm8p_m512_kernel.  At K = 6912 (Cin 256) it measured slower: 4.49 -> 4.92 ms.
  static int k_m512 = LD_KNOB_UNSET;
  if (conv && forced == 0 && p.N > 64 && p.N <= 128 && p.K >= 2048 && p.K <= 4096 && p.K % BK == 0 && (p.M + 511) / 512 >= 256 &&
      conv_input_bytes(p) < CONV_8P_MAX_BYTES && ld_knob("LD_GEMM_M512", 0, &k_m512) != 0) {
    g_last_route = ROUTE_512_8PHASE;
    return dry_run ? 0 : launch_8p_m512(p, stream);
  }
#endif
  if (cfg != 3 && cfg != 11 && cfg != 8) {
    g_last_route = ROUTE_128_2STAGE;
    return dry_run ? 0 : launch_cfg<128, 128, 2, 2, 2>(p, conv, stream);
  }
  // (round 1 default for the large linear layers: the register-staged 4-wave loop on 32x32x16 MFMAs, LD_GEMM_TILE=11, now only
  //  in the variants build; the 8-wave LDS-DMA kernel on 16x16x32 MFMAs is 3-8 % faster than it on all four DiT shapes: both are
  //  bound by the power governor, and the 16x16x32 form costs less energy per FLOP)
  static int use8p = -1;
  if (use8p < 0) { const char* e = getenv("LD_GEMM_8P"); use8p = e ? atoi(e) : 1; }
  auto big = [&](const GemmParams& q) {
    const bool conv_in_8p_range = !conv || conv_input_bytes(q) < CONV_8P_MAX_BYTES;
    if ((cfg == 8 || use8p) && cfg != 11 && conv_in_8p_range) {
      g_last_route = ROUTE_256_8PHASE;
      return dry_run ? 0 : launch_8p(q, conv, stream);
    }
#ifdef LD_VARIANTS
    if (!q.q_out && cfg == 11 && pp_ok && !conv) {
      g_last_route = ROUTE_256_W4R;
      return dry_run ? 0 : launch_w4r(q, stream);
    }
#endif
    g_last_route = ROUTE_256_2STAGE;      // (also the fused qkv split with LD_GEMM_8P=0: it lives in the 16x16x32 kernels only)
    return dry_run ? 0 : launch_cfg<256, 256, 2, 4, 2>(q, conv, stream);
  };
  // Wave quantisation: one 256x256 tile per CU at a time, so a grid of 4.3 "rounds" of 256 tiles costs 5.  When the last
  // round would be less than ~60 % full, the bottom rows are cut off and run as 128x128 tiles (two per CU, four times as
  // many) in a second launch: DiT proj / 4h->h GEMMs (N = 1920: 1112 tiles = 4.34 rounds) gain ~12 %.
  static int split = -1;
  if (split < 0) { const char* e = getenv("LD_GEMM_MSPLIT"); split = e ? atoi(e) : 1; }
  const int nbm = (p.M + 255) / 256, nbn = (p.N + 255) / 256;
  const long tiles = (long)nbm * nbn;
  const int ncu = 256;
  const long full = tiles / ncu, rem = tiles % ncu;
  // Round 5: at most half a round left over and a short K -> the whole rounds (tiles [0, full * ncu) of the raster, exactly `full`
  // per CU) on the 8-phase kernel, the rest cut in two along N: 2 * rem <= ncu half tiles, one per CU (ld_gemm8p_n128_kernel).
  // Measured (profiles/r05_gemm_half_tile_tail_ab.txt): a half tile takes 0.86 of a full tile's time -- its phases hold 8 MFMAs
  // between two barriers instead of 16 and the loop's fixed cost per phase no longer hides behind the partner wave -- so the form
  // only wins where the tail launch's own fixed costs matter: K <= 2048 (DiT qkv -12 us of 765, dense / 4h +-3 us); at K = 7680
  // (4h->h) it loses 24 us of 831 to the two-per-CU 128 x 128 tiles and is not used.  LD_GEMM_MSPLIT=2: the round-1..4 form below
  // for every shape, =3: half tiles for every K (A/B timing).
  const bool main_is_8p = (cfg == 8 || use8p) && cfg != 11;
  if ((split == 1 || split == 3) && !conv && main_is_8p && p.m_begin == 0 && full >= 2 && rem > 0 && 2 * rem <= ncu && p.K % BK == 0 &&
      (p.K <= 2048 || split == 3)) {
    GemmParams a = p, b = p;
    a.tile_begin = 0; a.tile_end = (int)(full * ncu);
    b.tile_begin = (int)(full * ncu); b.tile_end = 0;
    const int rc = big(a);
    if (rc || dry_run) return rc;
    return launch_8p_n128(b, stream);
  }
  // (rounds 1-4, and today for remainders between 50 and 60 % of a round) the bottom tile ROWS cut off and run as 128 x 128 tiles
  if (split && !conv && p.m_begin == 0 && full >= 2 && rem > 0 && rem * 100 <= 60 * ncu) {
    const int rows_main = (int)((full * ncu) / nbn);           // whole tile rows that fit in `full` rounds
    if (rows_main > 0 && rows_main < nbm) {
      GemmParams a = p, b = p;
      a.M = rows_main * 256;
      b.m_begin = rows_main * 256;
      const int rc = big(a);
      if (rc || dry_run) return rc;
      return launch_cfg<128, 128, 2, 2, 2>(b, conv, stream);
    }
  }
  return big(p);
}

int fill_epilogue(GemmParams& p, const ld_epilogue_t* e) {
  p.bias = nullptr; p.mul = nullptr; p.resid = nullptr; p.gate = nullptr; p.add2 = nullptr;
  p.act = 0; p.out_f32 = 0; p.resid_f32 = 0; p.rows_per_batch = 1 << 30; p.text_len = 0;
  p.gate_bstride = 0; p.gate_off_img = 0; p.gate_off_txt = 0;
  p.ldr = p.ldmul = p.ldadd = 0;
  if (!e) return LD_OK;
  LD_REQUIRE(e->act >= 0 && e->act <= LD_ACT_TANH, "ld_gemm: bad activation %d", e->act);
  p.bias = (const bf16_t*)e->bias; p.act = e->act;
  p.mul = (const bf16_t*)e->mul; p.ldmul = e->ldmul;
  p.resid = e->resid; p.ldr = e->ldr; p.resid_f32 = e->resid_f32;
  p.gate = (const bf16_t*)e->gate;
  p.add2 = (const bf16_t*)e->add2; p.ldadd = e->ldadd;
  p.out_f32 = e->out_f32;
  if (e->rows_per_batch > 0) p.rows_per_batch = e->rows_per_batch;
  p.text_len = e->text_len;
  p.gate_bstride = e->gate_bstride; p.gate_off_img = e->gate_off_img; p.gate_off_txt = e->gate_off_txt;
  return LD_OK;
}

// MXFP8 on the persistent two-phase loop (ld_gemm8p_mx_kernel): every tile of the raster, one workgroup per CU walking it
int launch_8p_mx(const GemmParams& p, hipStream_t stream) {
  const long ntiles = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    if (ncu <= 0 || (ncu & 7)) ncu = 256;
  }
  const_cast<GemmParams&>(p).group_m = (p.N + 255) / 256 <= 8 ? 4 : 8;      // (launch()'s raster rule)
  dim3 grid((unsigned)(ntiles > ncu ? ncu : ntiles)), block(512);
  constexpr int SMEM = LD_LDS_TOTAL;
  switch (pick_epilogue(p)) {
    case EPI_QKV: return launch_kernel<ld_gemm8p_mx_kernel<EPI_QKV>>("ld_gemm_qkv_heads_mxfp8", grid, block, SMEM, stream, p);
    case EPI_BIAS: return launch_kernel<ld_gemm8p_mx_kernel<EPI_BIAS>>("ld_gemm_mxfp8(8p)", grid, block, SMEM, stream, p);
    case EPI_GELU: return launch_kernel<ld_gemm8p_mx_kernel<EPI_GELU>>("ld_gemm_mxfp8(8p)", grid, block, SMEM, stream, p);
    case EPI_GATE: return launch_kernel<ld_gemm8p_mx_kernel<EPI_GATE>>("ld_gemm_mxfp8(8p)", grid, block, SMEM, stream, p);
    case EPI_GELU_MX: return launch_kernel<ld_gemm8p_mx_kernel<EPI_GELU_MX>>("ld_gemm_mxfp8(8p)", grid, block, SMEM, stream, p);
    default: return launch_kernel<ld_gemm8p_mx_kernel<EPI_GENERIC>>("ld_gemm_mxfp8(8p)", grid, block, SMEM, stream, p);
  }
}

int launch_f8(const GemmParams& p, hipStream_t stream) {
  // LD_GEMM_MX8P=0: the round-1 two-stage MXFP8 kernel (A/B timing; the fused qkv form exists in the persistent kernel only)
  static int k_mx8p = LD_KNOB_UNSET;
  if (p.mx_a && (p.q_out || ld_knob("LD_GEMM_MX8P", 1, &k_mx8p) != 0)) return launch_8p_mx(p, stream);
  constexpr int STAGE = (256 + 256) * 128;
  constexpr int EPIB = 8 * 32 * CW_STRIDE * 4;
  constexpr int SMEM = ((2 * STAGE > EPIB) ? 2 * STAGE : EPIB) + 4096;      // + the MX scale strips of both stages
  const int nbm = (p.M + 255) / 256, nbn = (p.N + 255) / 256;
  dim3 grid(nbm * nbn), block(512);
  if (p.mx_a) {
    switch (pick_epilogue(p)) {
      case EPI_BIAS: return launch_kernel<ld_gemm_f8_kernel<EPI_BIAS, true>>("ld_gemm_mxfp8", grid, block, SMEM, stream, p);
      case EPI_GELU: return launch_kernel<ld_gemm_f8_kernel<EPI_GELU, true>>("ld_gemm_mxfp8", grid, block, SMEM, stream, p);
      case EPI_GATE: return launch_kernel<ld_gemm_f8_kernel<EPI_GATE, true>>("ld_gemm_mxfp8", grid, block, SMEM, stream, p);
      case EPI_GELU_MX: return launch_kernel<ld_gemm_f8_kernel<EPI_GELU_MX, true>>("ld_gemm_mxfp8", grid, block, SMEM, stream, p);
      default: return launch_kernel<ld_gemm_f8_kernel<EPI_GENERIC, true>>("ld_gemm_mxfp8", grid, block, SMEM, stream, p);
    }
  }
  switch (pick_epilogue(p)) {
    case EPI_BIAS: return launch_kernel<ld_gemm_f8_kernel<EPI_BIAS, false>>("ld_gemm_fp8", grid, block, SMEM, stream, p);
    case EPI_GELU: return launch_kernel<ld_gemm_f8_kernel<EPI_GELU, false>>("ld_gemm_fp8", grid, block, SMEM, stream, p);
    case EPI_GATE: return launch_kernel<ld_gemm_f8_kernel<EPI_GATE, false>>("ld_gemm_fp8", grid, block, SMEM, stream, p);
    default: return launch_kernel<ld_gemm_f8_kernel<EPI_GENERIC, false>>("ld_gemm_fp8", grid, block, SMEM, stream, p);
  }
}

// MXFP8 quantiser (OCP Microscaling v1.0 container: e4m3 elements + one E8M0 scale per 32 consecutive K elements, byte =
// exponent + 127), scale = smallest power of two >= amax / 448, elements = e4m3 cast of x / scale.  One wave per row; a block is the four
// 8-element chunks of four adjacent lanes.
__global__ __launch_bounds__(256) void ld_quant_mxfp8_kernel(const bf16_t* x, long ldx, unsigned char* q, long ldq,
                                                             unsigned char* sc, long lds, int rows, int K) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int nchunk = K >> 3;
  for (int c = lane; c < ((nchunk + 63) & ~63); c += 64) {
    u32x4_t v = (u32x4_t){0u, 0u, 0u, 0u};
    if (c < nchunk) v = *(const u32x4_t*)(x + (long)r * ldx + c * 8);
    float f[8];
    float amax = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[2 * e] = bf_lo(v[e]); f[2 * e + 1] = bf_hi(v[e]); amax = fmaxf(amax, fmaxf(fabsf(f[2 * e]), fabsf(f[2 * e + 1]))); }
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    // E8M0 byte sb: the smallest power of two 2^(sb - 127) >= amax / 448, so that no element saturates (the floor rule of
    // the MX paper, 2^(floor(log2 amax) - 8), clips block maxima in [448, 512) x scale and measured 20-45 % more error)
    const uint32_t tb = __float_as_uint(amax * (1.0f / 448.0f));
    int sb = (int)((tb >> 23) & 0xffu) + ((tb & 0x7fffffu) != 0u ? 1 : 0);
    sb = amax > 0.f ? (sb < 1 ? 1 : (sb > 254 ? 254 : sb)) : 0;
    const float inv = __uint_as_float((uint32_t)(254 - sb) << 23);     // 2^(127 - sb)
    if (c < nchunk) {
      u32x2_t o;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        unsigned w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(f[4 * h] * inv, f[4 * h + 1] * inv, w, false);       // |f| * inv <= 448 by construction
        w = __builtin_amdgcn_cvt_pk_fp8_f32(f[4 * h + 2] * inv, f[4 * h + 3] * inv, w, true);
        o[h] = w;
      }
      *(u32x2_t*)(q + (long)r * ldq + c * 8) = o;
      if ((c & 3) == 0) sc[((long)(c >> 4) * lds + r) * 4 + ((c >> 2) & 3)] = (unsigned char)sb;   // [K/128][lds rows][4]
    }
  }
}

// Row-wise dynamic quantisation to OCP e4m3: scale[r] = amax(row) / 448 (1 for an all-zero row), q = cvt(x / scale).
// One wave per row, the row stays in registers between the two passes (K <= 8192: 16 chunks of 8 per lane).
__global__ __launch_bounds__(256) void ld_quant_fp8_kernel(const bf16_t* x, long ldx, unsigned char* q, long ldq, float* scale,
                                                           int rows, int K) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int nchunk = K >> 3;
  u32x4_t v[16];
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = lane + 64 * i;
    v[i] = (u32x4_t){0u, 0u, 0u, 0u};
    if (c < nchunk) v[i] = *(const u32x4_t*)(x + (long)r * ldx + c * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(bf_lo(v[i][e])), fabsf(bf_hi(v[i][e]))));
  }
  amax = wave_max(amax);
  const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
  const float inv = 1.0f / sc;
  if (lane == 0) scale[r] = sc;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = lane + 64 * i;
    if (c >= nchunk) continue;
    u32x2_t o;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float f0 = fminf(fmaxf(bf_lo(v[i][2 * h]) * inv, -448.f), 448.f), f1 = fminf(fmaxf(bf_hi(v[i][2 * h]) * inv, -448.f), 448.f);
      const float f2 = fminf(fmaxf(bf_lo(v[i][2 * h + 1]) * inv, -448.f), 448.f), f3 = fminf(fmaxf(bf_hi(v[i][2 * h + 1]) * inv, -448.f), 448.f);
      unsigned w = 0;
      w = __builtin_amdgcn_cvt_pk_fp8_f32(f0, f1, w, false);
      w = __builtin_amdgcn_cvt_pk_fp8_f32(f2, f3, w, true);
      o[h] = w;
    }
    *(u32x2_t*)(q + (long)r * ldq + c * 8) = o;
  }
}

}  // namespace

LD_API int ld_gemm_bf16(const void* A, int64_t lda, const void* W, void* out, int64_t ldo,
                        int64_t M, int64_t N, int64_t K, const ld_epilogue_t* epi, void* stream) {
  LD_REQUIRE(A && W && out, "ld_gemm_bf16: null pointer");
  LD_REQUIRE(M > 0 && N > 0 && K > 0, "ld_gemm_bf16: empty problem M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
  LD_REQUIRE(K % BK == 0, "ld_gemm_bf16: K=%ld must be a multiple of %d", (long)K, BK);
  LD_REQUIRE(lda % 8 == 0, "ld_gemm_bf16: lda=%ld must be a multiple of 8 elements", (long)lda);
  LD_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)out & 15) == 0,
             "ld_gemm_bf16: pointers must be 16-byte aligned");
  LD_REQUIRE(M * (int64_t)N < (1LL << 40) && M < (1LL << 31) && N < (1LL << 31), "ld_gemm_bf16: problem too large");
  GemmParams p{};
  p.A = (const bf16_t*)A; p.W = (const bf16_t*)W; p.out = out;
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.lda = lda; p.ldo = ldo;
  int rc = fill_epilogue(p, epi);
  if (rc) return rc;
  return launch(p, false, (hipStream_t)stream);
}

LD_API int ld_gemm_qkv_heads(const void* A, int64_t lda, const void* W, const void* bias, int64_t M, int64_t K,
                             void* Q, void* Kh, void* Vt, int64_t B, int64_t Ntok, int64_t heads, int64_t Npad,
                             const void* q_w, const void* q_b, const void* k_w, const void* k_b, float eps, void* stream) {
  LD_REQUIRE(A && W && bias && Q && Kh && Vt && q_w && q_b && k_w && k_b, "ld_gemm_qkv_heads: null pointer");
  LD_REQUIRE(M == B * Ntok && B > 0 && heads > 0 && K > 0, "ld_gemm_qkv_heads: M=%ld must be B*Ntok=%ld", (long)M, (long)(B * Ntok));
  LD_REQUIRE(K % BK == 0 && lda % 8 == 0, "ld_gemm_qkv_heads: K=%ld must be a multiple of %d, lda of 8", (long)K, BK);
  LD_REQUIRE(Ntok % 8 == 0 && Ntok >= 256 && Npad % 8 == 0 && Npad >= Ntok, "ld_gemm_qkv_heads: Ntok=%ld (multiple of 8, >= 256), Npad=%ld", (long)Ntok, (long)Npad);
  LD_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)bias & 15) == 0 && ((uintptr_t)Q & 15) == 0 &&
             ((uintptr_t)Kh & 15) == 0 && ((uintptr_t)Vt & 15) == 0 && ((uintptr_t)q_w & 15) == 0 && ((uintptr_t)q_b & 15) == 0 &&
             ((uintptr_t)k_w & 15) == 0 && ((uintptr_t)k_b & 15) == 0, "ld_gemm_qkv_heads: pointers must be 16-byte aligned");
  LD_REQUIRE(M < (1LL << 31), "ld_gemm_qkv_heads: problem too large");
  GemmParams p{};
  p.A = (const bf16_t*)A; p.W = (const bf16_t*)W; p.out = nullptr;
  p.M = (int)M; p.N = (int)(3 * heads * 64); p.K = (int)K; p.lda = lda; p.ldo = 0;
  int rc = fill_epilogue(p, nullptr);
  if (rc) return rc;
  p.bias = (const bf16_t*)bias;
  p.q_out = (bf16_t*)Q; p.k_out = (bf16_t*)Kh; p.vt_out = (bf16_t*)Vt;
  p.qn_w = (const bf16_t*)q_w; p.qn_b = (const bf16_t*)q_b; p.kn_w = (const bf16_t*)k_w; p.kn_b = (const bf16_t*)k_b;
  p.heads = (int)heads; p.Ntok = (int)Ntok; p.Npad = (int)Npad; p.qk_eps = eps;
  return launch(p, false, (hipStream_t)stream);
}

static int conv_cl(const void* in_padded, const void* Wt, void* out, int64_t ldo, int64_t T, int64_t H, int64_t W, int64_t Cin,
                   int64_t Cout, int64_t kT, int64_t kH, int64_t kW, const ld_epilogue_t* epi, float* gn_partials, void* stream) {
  LD_REQUIRE(in_padded && Wt && out, "ld_conv_cl_bf16: null pointer");
  LD_REQUIRE(T > 0 && H > 0 && W > 0 && Cout > 0, "ld_conv_cl_bf16: empty problem");
  LD_REQUIRE(Cin % BK == 0, "ld_conv_cl_bf16: Cin=%ld must be a multiple of %d (zero-pad channels)", (long)Cin, BK);
  LD_REQUIRE(kT >= 1 && kH >= 1 && kW >= 1 && (kH & 1) && (kW & 1), "ld_conv_cl_bf16: bad kernel size");
  LD_REQUIRE(T * H * W < (1LL << 31), "ld_conv_cl_bf16: too many output positions");
  GemmParams p{};
  p.A = (const bf16_t*)in_padded; p.W = (const bf16_t*)Wt; p.out = out;
  p.M = (int)(T * H * W); p.N = (int)Cout; p.K = (int)(kT * kH * kW * Cin); p.lda = 0; p.ldo = ldo;
  p.H = (int)H; p.W_ = (int)W; p.Hp = (int)(H + kH - 1); p.Wp = (int)(W + kW - 1);
  p.Cin = (int)Cin; p.kH = (int)kH; p.kW = (int)kW;
  LD_REQUIRE(conv_input_bytes(p) < CONV_MAX_BYTES, "ld_conv_cl_bf16: padded input of %ld bytes is beyond the 8 GiB the kernels address "
             "(split the chunk in time)", conv_input_bytes(p));
  int rc = fill_epilogue(p, epi);
  if (rc) return rc;
  // a handful of output channels (the VAE's conv_out): not a GEMM worth a 128-wide tile -- ld_conv_narrow.hip reads the input once
  const bool plain = !p.act && !p.mul && !p.resid && !p.gate && !p.add2 && !p.out_f32;
  if (gn_partials) {
    // the epilogue sums the values it stores: 16-byte rows of 8 channels only (the vector path of both conv epilogues)
    LD_REQUIRE(Cout % 8 == 0 && ldo % 8 == 0 && !p.out_f32 && (!p.resid || p.ldr % 8 == 0) && (!p.mul || p.ldmul % 8 == 0) &&
               (!p.add2 || p.ldadd % 8 == 0), "ld_conv_cl_bf16_gn: Cout, ldo and the epilogue operands' leading dimensions must be multiples of 8, bf16 output");
    LD_REQUIRE(((uintptr_t)gn_partials & 15) == 0, "ld_conv_cl_bf16_gn: gn_partials must be 16-byte aligned");
    p.gn_part = gn_partials;
  } else {
    rc = ld_conv_narrow_try(in_padded, Wt, p.bias, out, ldo, T, H, W, Cin, Cout, kT, kH, kW, plain, (hipStream_t)stream, false);
    if (rc <= 0) return rc;
  }
  return launch(p, true, (hipStream_t)stream);
}

LD_API int ld_conv_cl_bf16(const void* in_padded, const void* Wt, void* out, int64_t ldo,
                           int64_t T, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                           int64_t kT, int64_t kH, int64_t kW, const ld_epilogue_t* epi, void* stream) {
  return conv_cl(in_padded, Wt, out, ldo, T, H, W, Cin, Cout, kT, kH, kW, epi, nullptr, stream);
}

LD_API int64_t ld_conv_gn_partials_size(int64_t M, int64_t Cout) { return ((M + 63) / 64) * (Cout / 4) * 2; }

LD_API int ld_conv_cl_bf16_gn(const void* in_padded, const void* Wt, void* out, int64_t ldo,
                              int64_t T, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                              int64_t kT, int64_t kH, int64_t kW, const ld_epilogue_t* epi, float* gn_partials, void* stream) {
  LD_REQUIRE(gn_partials, "ld_conv_cl_bf16_gn: null gn_partials");
  return conv_cl(in_padded, Wt, out, ldo, T, H, W, Cin, Cout, kT, kH, kW, epi, gn_partials, stream);
}

LD_API int ld_conv_route(int64_t T, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t kT, int64_t kH, int64_t kW) {
  if (T <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cin <= 0 || Cin % BK || kT < 1 || kH < 1 || kW < 1 || T * H * W >= (1LL << 31))
    return ld_set_error(LD_ERR_INVALID, "ld_conv_route: bad shape");
  GemmParams p{};
  p.M = (int)(T * H * W); p.N = (int)Cout; p.K = (int)(kT * kH * kW * Cin);
  p.H = (int)H; p.W_ = (int)W; p.Hp = (int)(H + kH - 1); p.Wp = (int)(W + kW - 1);
  p.Cin = (int)Cin; p.kH = (int)kH; p.kW = (int)kW;
  if (conv_input_bytes(p) >= CONV_MAX_BYTES) return ld_set_error(LD_ERR_INVALID, "ld_conv_route: padded input beyond 8 GiB");
  if (ld_conv_narrow_try(nullptr, nullptr, nullptr, nullptr, Cout, T, H, W, Cin, Cout, kT, kH, kW, true, nullptr, true) == 0) return ROUTE_NARROW;
  g_last_route = -1;
  const int rc = launch(p, true, nullptr, /*dry_run=*/true);
  return rc ? rc : g_last_route;
}

LD_API int ld_quantize_fp8(const void* x, int64_t ldx, void* q, int64_t ldq, float* scale, int64_t rows, int64_t K,
                           void* stream) {
  LD_REQUIRE(x && q && scale && rows > 0, "ld_quantize_fp8: bad args");
  LD_REQUIRE(K % 8 == 0 && K <= 8192 && ldx % 8 == 0 && ldq % 8 == 0, "ld_quantize_fp8: K=%ld must be a multiple of 8 and <= 8192", (long)K);
  hipLaunchKernelGGL(ld_quant_fp8_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (long)ldx, (unsigned char*)q, (long)ldq, scale, (int)rows, (int)K);
  return ld_check_launch("ld_quantize_fp8");
}

LD_API int ld_gemm_fp8(const void* A8, int64_t lda, const float* scale_a, const void* W8, const float* scale_w, void* out,
                       int64_t ldo, int64_t M, int64_t N, int64_t K, const ld_epilogue_t* epi, void* stream) {
  LD_REQUIRE(A8 && W8 && out && scale_a && scale_w, "ld_gemm_fp8: null pointer");
  LD_REQUIRE(M > 0 && N > 0 && K > 0 && K % 128 == 0, "ld_gemm_fp8: K=%ld must be a positive multiple of 128", (long)K);
  LD_REQUIRE(lda % 16 == 0 && ((uintptr_t)A8 & 15) == 0 && ((uintptr_t)W8 & 15) == 0 && ((uintptr_t)out & 15) == 0,
             "ld_gemm_fp8: lda and pointers must be 16-byte aligned");
  LD_REQUIRE(M * lda < (1LL << 32) && N * K < (1LL << 32), "ld_gemm_fp8: operand larger than 4 GiB");
  GemmParams p{};
  p.A = (const bf16_t*)A8; p.W = (const bf16_t*)W8; p.out = out;
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.lda = lda; p.ldo = ldo;
  p.scale_a = scale_a; p.scale_w = scale_w;
  p.group_m = 8;
  int rc = fill_epilogue(p, epi);
  if (rc) return rc;
  return launch_f8(p, (hipStream_t)stream);
}

LD_API int ld_quantize_mxfp8(const void* x, int64_t ldx, void* q, int64_t ldq, void* scales, int64_t lds, int64_t rows,
                             int64_t K, void* stream) {
  LD_REQUIRE(x && q && scales && rows > 0, "ld_quantize_mxfp8: bad args");
  LD_REQUIRE(K % 128 == 0 && ldx % 8 == 0 && ldq % 8 == 0 && lds >= rows, "ld_quantize_mxfp8: K=%ld must be a multiple of 128, lds >= rows", (long)K);
  hipLaunchKernelGGL(ld_quant_mxfp8_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (long)ldx, (unsigned char*)q, (long)ldq, (unsigned char*)scales, (long)lds, (int)rows, (int)K);
  return ld_check_launch("ld_quantize_mxfp8");
}

LD_API int ld_gemm_mxfp8(const void* A8, int64_t lda, const void* scales_a, const void* W8, const void* scales_w, void* out,
                         int64_t ldo, void* out_scales, int64_t ldos, int64_t M, int64_t N, int64_t K,
                         const ld_epilogue_t* epi, void* stream) {
  LD_REQUIRE(A8 && W8 && out && scales_a && scales_w, "ld_gemm_mxfp8: null pointer");
  LD_REQUIRE(M > 0 && N > 0 && K > 0 && K % 128 == 0, "ld_gemm_mxfp8: K=%ld must be a positive multiple of 128", (long)K);
  LD_REQUIRE(lda % 16 == 0 && ((uintptr_t)A8 & 15) == 0 && ((uintptr_t)W8 & 15) == 0 && ((uintptr_t)out & 15) == 0 &&
             ((uintptr_t)scales_a & 3) == 0 && ((uintptr_t)scales_w & 3) == 0, "ld_gemm_mxfp8: alignment (operands 16 B, scales 4 B)");
  LD_REQUIRE(M * lda < (1LL << 32) && N * K < (1LL << 32), "ld_gemm_mxfp8: operand larger than 4 GiB");
  GemmParams p{};
  p.A = (const bf16_t*)A8; p.W = (const bf16_t*)W8; p.out = out;
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.lda = lda; p.ldo = ldo;
  p.mx_a = (const unsigned char*)scales_a; p.mx_w = (const unsigned char*)scales_w;      // [K/128][M][4], [K/128][N][4] contiguous
  p.group_m = 8;
  int rc = fill_epilogue(p, epi);
  if (rc) return rc;
  if (out_scales) {     // MXFP8 output (the 4h activation handed to the next MXFP8 GEMM): bias + GELU-tanh only
    LD_REQUIRE(N % 128 == 0 && ldo % 8 == 0 && ldos >= M, "ld_gemm_mxfp8: MXFP8 output needs N %% 128 == 0, ldo %% 8 == 0, ldos >= M");
    LD_REQUIRE(p.act == LD_ACT_GELU_TANH && !p.resid && !p.gate && !p.add2 && !p.mul && !p.out_f32,
               "ld_gemm_mxfp8: MXFP8 output is the bias + GELU-tanh epilogue only");
    p.mx_out = (unsigned char*)out_scales; p.ld_mx_out = ldos;
  }
  return launch_f8(p, (hipStream_t)stream);
}

/* The fused qkv head split on MXFP8 operands (BASELINE configs[4]): ld_gemm_qkv_heads with A / W as e4m3 codes + block scales. */
LD_API int ld_gemm_qkv_heads_mxfp8(const void* A8, int64_t lda, const void* scales_a, const void* W8, const void* scales_w,
                                   const void* bias, int64_t M, int64_t K, void* Q, void* Kh, void* Vt, int64_t B, int64_t Ntok,
                                   int64_t heads, int64_t Npad, const void* q_w, const void* q_b, const void* k_w, const void* k_b,
                                   float eps, void* stream) {
  LD_REQUIRE(A8 && W8 && scales_a && scales_w && bias && Q && Kh && Vt && q_w && q_b && k_w && k_b, "ld_gemm_qkv_heads_mxfp8: null pointer");
  LD_REQUIRE(M == B * Ntok && B > 0 && heads > 0 && K > 0 && K % 128 == 0, "ld_gemm_qkv_heads_mxfp8: M=%ld must be B*Ntok=%ld, K=%ld a multiple of 128",
             (long)M, (long)(B * Ntok), (long)K);
  LD_REQUIRE(Ntok % 8 == 0 && Ntok >= 256 && Npad % 8 == 0 && Npad >= Ntok, "ld_gemm_qkv_heads_mxfp8: Ntok=%ld (multiple of 8, >= 256), Npad=%ld", (long)Ntok, (long)Npad);
  LD_REQUIRE(lda % 16 == 0 && ((uintptr_t)A8 & 15) == 0 && ((uintptr_t)W8 & 15) == 0 && ((uintptr_t)scales_a & 3) == 0 && ((uintptr_t)scales_w & 3) == 0 &&
             ((uintptr_t)bias & 15) == 0 && ((uintptr_t)Q & 15) == 0 && ((uintptr_t)Kh & 15) == 0 && ((uintptr_t)Vt & 15) == 0 &&
             ((uintptr_t)q_w & 15) == 0 && ((uintptr_t)q_b & 15) == 0 && ((uintptr_t)k_w & 15) == 0 && ((uintptr_t)k_b & 15) == 0,
             "ld_gemm_qkv_heads_mxfp8: alignment (operands 16 B, scales 4 B)");
  LD_REQUIRE(M * lda < (1LL << 32) && 3 * heads * 64 * K < (1LL << 32) && M < (1LL << 31), "ld_gemm_qkv_heads_mxfp8: operand larger than 4 GiB");
  GemmParams p{};
  p.A = (const bf16_t*)A8; p.W = (const bf16_t*)W8; p.out = nullptr;
  p.M = (int)M; p.N = (int)(3 * heads * 64); p.K = (int)K; p.lda = lda; p.ldo = 0;
  p.mx_a = (const unsigned char*)scales_a; p.mx_w = (const unsigned char*)scales_w;
  int rc = fill_epilogue(p, nullptr);
  if (rc) return rc;
  p.bias = (const bf16_t*)bias;
  p.q_out = (bf16_t*)Q; p.k_out = (bf16_t*)Kh; p.vt_out = (bf16_t*)Vt;
  p.qn_w = (const bf16_t*)q_w; p.qn_b = (const bf16_t*)q_b; p.kn_w = (const bf16_t*)k_w; p.kn_b = (const bf16_t*)k_b;
  p.heads = (int)heads; p.Ntok = (int)Ntok; p.Npad = (int)Npad; p.qk_eps = eps;
  return launch_8p_mx(p, (hipStream_t)stream);
}

#ifdef LD_GEMM_TRACE
LD_API int ld_gemm_trace_set(void* buf, int64_t capacity_records) {
  unsigned long long* b = (unsigned long long*)buf; int cap = (int)capacity_records;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_trace), &b, sizeof(b)) != hipSuccess) return 1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_trace_cap), &cap, sizeof(cap)) != hipSuccess) return 1;
  return 0;
}
#endif

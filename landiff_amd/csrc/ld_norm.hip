// HBM-bound normalisation kernels: LayerNorm (+AdaLN modulate, text/image regions), qkv head split with
// QK-LayerNorm or 3D RoPE and V transposition, GroupNorm statistics, fused GroupNorm/SpatialNorm + swish.
//
// Replaces: layer.input_layernorm/post_attention_layernorm + modulate
// (landiff/diffusion/dit_video_concat.py:388,577-586,601-611), query/key_layernorm (:649-653), sat's
// _transpose_for_scores; MultiheadAttention's head split + apply_rope (landiff/tokenizer/modules/blocks.py:172-180);
// Normalize/GroupNorm + nonlinearity (vq_gan_blocks.py:29-38), SpatialNorm3D.forward + nonlinearity
// (landiff/diffusion/vae_modules/cp_enc_dec.py:67-69,546-569).
//
// All of these are pure streaming passes: 16-byte coalesced loads, one wave64 per row (no LDS, no barriers)
// for the LayerNorms, fp32 statistics, bf16 outputs rounded at the same points the reference's bf16 ops round.
#include "ld_common.h"
#include <stdlib.h>
#include "../../include/landiff_hip.h"

namespace {

// ---------------------------------------------------------------------------------------------
// LayerNorm over D (D % 8 == 0, D <= 4096), optional AdaLN modulation selected per row region.
// ---------------------------------------------------------------------------------------------
struct LnParams {
  const void* x; void* out;
  const bf16_t* w; const bf16_t* b;
  const bf16_t* mod;            // modulation vectors, or null
  long ldx, ldo;
  int rows, D;
  float eps;
  int x_f32, out_f32;
  int rows_per_batch, text_len;
  long mod_bstride, shift_img, scale_img, shift_txt, scale_txt;
};

template <int NC>   // chunks (of 8 elements) per lane
__global__ __launch_bounds__(256) void ld_layernorm_kernel(LnParams p) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= p.rows) return;
  const int nchunk = p.D >> 3;
  float v[NC][8];
  float s = 0.f, s_odd = 0.f;       // even / odd elements summed apart: the order of ld_layernorm_mod_kernel's packed sums
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      if (p.x_f32) {
        const float* xr = (const float*)p.x + (long)r * p.ldx + c * 8;
        const f32x4_t a = *(const f32x4_t*)xr, b4 = *(const f32x4_t*)(xr + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[i][e] = a[e]; v[i][4 + e] = b4[e]; }
      } else {
        const u32x4_t a = *(const u32x4_t*)((const bf16_t*)p.x + (long)r * p.ldx + c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[i][2 * e] = bf_lo(a[e]); v[i][2 * e + 1] = bf_hi(a[e]); }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { s += v[i][2 * e]; s_odd += v[i][2 * e + 1]; }
    }
  }
  const float mean = wave_sum(s + s_odd) / (float)p.D;
  float ss = 0.f, ss_odd = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    if (lane + 64 * i < nchunk) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d0 = v[i][2 * e] - mean, d1 = v[i][2 * e + 1] - mean;
        ss += d0 * d0; ss_odd += d1 * d1;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(ss + ss_odd) / (float)p.D + p.eps);
  const bf16_t* shift = nullptr; const bf16_t* scale = nullptr;
  if (p.mod) {
    const int bb = r / p.rows_per_batch;
    const bool txt = (r - bb * p.rows_per_batch) < p.text_len;
    shift = p.mod + bb * p.mod_bstride + (txt ? p.shift_txt : p.shift_img);
    scale = p.mod + bb * p.mod_bstride + (txt ? p.scale_txt : p.scale_img);
  }
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    if (c >= nchunk) continue;
    float y[8];
    float wv[8], bv[8];
    if (p.w) {
      const u32x4_t ww = *(const u32x4_t*)(p.w + c * 8), bw = *(const u32x4_t*)(p.b + c * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) { wv[2 * e] = bf_lo(ww[e]); wv[2 * e + 1] = bf_hi(ww[e]); bv[2 * e] = bf_lo(bw[e]); bv[2 * e + 1] = bf_hi(bw[e]); }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = (v[i][e] - mean) * rstd;
      if (p.w) t = t * wv[e] + bv[e];
      y[e] = t;
    }
    if (shift) {
      const u32x4_t sh = *(const u32x4_t*)(shift + c * 8), sc = *(const u32x4_t*)(scale + c * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // modulate(): x * (1 + scale) + shift, every op in bf16 as in the reference
        const float s0 = rbf(1.0f + bf_lo(sc[e])), s1 = rbf(1.0f + bf_hi(sc[e]));
        // (the last rounding, of the sum, is the one the bf16 pack / the out_f32 branch below performs)
        y[2 * e] = rbf(rbf(y[2 * e]) * s0) + bf_lo(sh[e]);
        y[2 * e + 1] = rbf(rbf(y[2 * e + 1]) * s1) + bf_hi(sh[e]);
      }
    }
    if (p.out_f32) {
      if (shift) {
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = rbf(y[e]);
      }
      float* o = (float*)p.out + (long)r * p.ldo + c * 8;
      *(f32x4_t*)o = (f32x4_t){y[0], y[1], y[2], y[3]};
      *(f32x4_t*)(o + 4) = (f32x4_t){y[4], y[5], y[6], y[7]};
    } else {
      u32x4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(y[2 * e], y[2 * e + 1]);
      *(u32x4_t*)((bf16_t*)p.out + (long)r * p.ldo + c * 8) = o;
    }
  }
}

// The DiT's hot form of the kernel above (bf16 in / out, affine weights, AdaLN modulation): the same operations at the same
// roundings, arranged for VALU issue count -- the general kernel spends ~25 VALU operations per element, more issue time
// than the row's HBM time; this one ~11:
//  * element PAIRS: one v_cvt_pk_bf16_f32 per bf16 rounding of two elements, v_pk_{add,mul,fma}_f32 for the arithmetic (a
//    bf16 pair unpacks into an adjacent register pair);
//  * TWO rows per wave: the four per-column vectors (weight, bias, shift, scale) are loaded and unpacked once for both.
// The statistics are summed per pair lane (even / odd elements) and combined at the end.
typedef ld_f32x2_t f32x2_t;

template <int NC, bool SAME>
__device__ __forceinline__ void ln_mod2_apply(const LnParams& p, f32x2_t (&v)[2][NC][4], const float (&rstd)[2], const bf16_t* const (&shift)[2],
                                              const bf16_t* const (&scale)[2], int r0, bool has1, int lane, int nchunk) {
  const f32x2_t one2 = {1.0f, 1.0f};
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    if (c >= nchunk) continue;
    const u32x4_t ww = *(const u32x4_t*)(p.w + c * 8), bw = *(const u32x4_t*)(p.b + c * 8);
    u32x4_t sh[2], sc[2];
    sh[0] = *(const u32x4_t*)(shift[0] + c * 8); sc[0] = *(const u32x4_t*)(scale[0] + c * 8);
    if (!SAME) { sh[1] = *(const u32x4_t*)(shift[1] + c * 8); sc[1] = *(const u32x4_t*)(scale[1] + c * 8); }
    u32x4_t o[2];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const f32x2_t w2 = unpack_bf16x2(ww[e]), b2 = unpack_bf16x2(bw[e]);
      f32x2_t sp[2], st[2];
#pragma unroll
      for (int j = 0; j < (SAME ? 1 : 2); ++j) {
        sp[j] = rbf2(unpack_bf16x2(sc[j][e]) + one2);
        st[j] = unpack_bf16x2(sh[j][e]);
      }
      if (SAME) { sp[1] = sp[0]; st[1] = st[0]; }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f32x2_t rs = {rstd[j], rstd[j]};
        f32x2_t y = rbf2(v[j][i][e] * rs * w2 + b2);              // LayerNorm output in bf16
        // modulate(): x * (1 + scale) + shift, every op in bf16 as in the reference (dit_video_concat.py:388)
        y = rbf2(y * sp[j]) + st[j];
        o[j][e] = pack_bf16x2(y[0], y[1]);
      }
    }
    *(u32x4_t*)((bf16_t*)p.out + (long)r0 * p.ldo + c * 8) = o[0];
    if (has1) *(u32x4_t*)((bf16_t*)p.out + (long)(r0 + 1) * p.ldo + c * 8) = o[1];
  }
}

template <int NC>
__global__ __launch_bounds__(256) void ld_layernorm_mod_kernel(LnParams p) {
  const int lane = threadIdx.x & 63;
  const int r0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
  if (r0 >= p.rows) return;
  const bool has1 = r0 + 1 < p.rows;
  const int nchunk = p.D >> 3;
  u32x4_t raw[2][NC];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const bf16_t* xr = (const bf16_t*)p.x + (long)(r0 + (has1 ? j : 0)) * p.ldx;      // odd row count: the last wave does its row twice
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + 64 * i;
      raw[j][i] = (u32x4_t){0u, 0u, 0u, 0u};
      if (c < nchunk) raw[j][i] = *(const u32x4_t*)(xr + c * 8);
    }
  }
  const bf16_t* shift[2]; const bf16_t* scale[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = r0 + (has1 ? j : 0);
    const int bb = r / p.rows_per_batch;
    const bool txt = (r - bb * p.rows_per_batch) < p.text_len;
    shift[j] = p.mod + bb * p.mod_bstride + (txt ? p.shift_txt : p.shift_img);
    scale[j] = p.mod + bb * p.mod_bstride + (txt ? p.scale_txt : p.scale_img);
  }
  f32x2_t v[2][NC][4];
  float rstd[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    f32x2_t s2 = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[j][i][e] = unpack_bf16x2(raw[j][i][e]); s2 += v[j][i][e]; }      // chunks past the row are zeros
    const float mean = wave_sum(s2[0] + s2[1]) / (float)p.D;
    const f32x2_t mean2 = {mean, mean};
    f32x2_t ss2 = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      if (lane + 64 * i < nchunk) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[j][i][e] -= mean2; ss2 += v[j][i][e] * v[j][i][e]; }
      }
    }
    rstd[j] = rsqrtf(wave_sum(ss2[0] + ss2[1]) / (float)p.D + p.eps);
  }
  const bf16_t* const shc[2] = {shift[0], shift[1]};
  const bf16_t* const scc[2] = {scale[0], scale[1]};
  if (shift[0] == shift[1] && scale[0] == scale[1]) ln_mod2_apply<NC, true>(p, v, rstd, shc, scc, r0, has1, lane, nchunk);
  else ln_mod2_apply<NC, false>(p, v, rstd, shc, scc, r0, has1, lane, nchunk);
}

// The DiT's block LayerNorms with MXFP8 output in the two-rows-per-wave, packed-math form of ld_layernorm_mod_kernel (round 6: the
// one-row kernel below spends ~20 scalar VALU operations per element on bf16 roundings and ran at 2.7 TB/s; same bits).
template <int NC, bool SAME>
__device__ __forceinline__ void ln_mod2_apply_mx(const LnParams& p, f32x2_t (&v)[2][NC][4], const float (&rstd)[2], const bf16_t* const (&shift)[2],
                                                 const bf16_t* const (&scale)[2], int r0, bool has1, int lane, int nchunk,
                                                 unsigned char* mxs, long lds) {
  const f32x2_t one2 = {1.0f, 1.0f};
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    const bool live = c < nchunk;
    const int cc = live ? c : 0;                      // (lanes past the row's last chunk read chunk 0 and store nothing)
    const u32x4_t ww = *(const u32x4_t*)(p.w + cc * 8), bw = *(const u32x4_t*)(p.b + cc * 8);
    u32x4_t sh[2], sc[2];
    sh[0] = *(const u32x4_t*)(shift[0] + cc * 8); sc[0] = *(const u32x4_t*)(scale[0] + cc * 8);
    if (!SAME) { sh[1] = *(const u32x4_t*)(shift[1] + cc * 8); sc[1] = *(const u32x4_t*)(scale[1] + cc * 8); }
    f32x2_t y[2][4];
    float amax[2] = {0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const f32x2_t w2 = unpack_bf16x2(ww[e]), b2 = unpack_bf16x2(bw[e]);
      f32x2_t sp[2], st[2];
#pragma unroll
      for (int j = 0; j < (SAME ? 1 : 2); ++j) {
        sp[j] = rbf2(unpack_bf16x2(sc[j][e]) + one2);
        st[j] = unpack_bf16x2(sh[j][e]);
      }
      if (SAME) { sp[1] = sp[0]; st[1] = st[0]; }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f32x2_t rs = {rstd[j], rstd[j]};
        f32x2_t t = rbf2(v[j][i][e] * rs * w2 + b2);              // LayerNorm output in bf16
        t = rbf2(rbf2(t * sp[j]) + st[j]);                        // modulate(), every op in bf16: the value the plain kernel stores
        y[j][e] = t;
        amax[j] = fmaxf(amax[j], fmaxf(fabsf(t[0]), fabsf(t[1])));
      }
    }
    // a 32-element MX block = the chunks of four adjacent lanes; scale = smallest power of two >= amax / 448 (ld_quant_mxfp8_kernel)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float am = live ? amax[j] : 0.f;
      am = fmaxf(am, __shfl_xor(am, 1, 64));
      am = fmaxf(am, __shfl_xor(am, 2, 64));
      if (!live || (j == 1 && !has1)) continue;
      const uint32_t tb = __float_as_uint(am * (1.0f / 448.0f));
      int sb = (int)((tb >> 23) & 0xffu) + ((tb & 0x7fffffu) != 0u ? 1 : 0);
      sb = am > 0.f ? (sb < 1 ? 1 : (sb > 254 ? 254 : sb)) : 0;
      const float inv = __uint_as_float((uint32_t)(254 - sb) << 23);
      u32x2_t o;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        unsigned w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(y[j][2 * h][0] * inv, y[j][2 * h][1] * inv, w, false);     // <= 448 by construction
        w = __builtin_amdgcn_cvt_pk_fp8_f32(y[j][2 * h + 1][0] * inv, y[j][2 * h + 1][1] * inv, w, true);
        o[h] = w;
      }
      const long r = r0 + j;
      *(u32x2_t*)((unsigned char*)p.out + r * p.ldo + c * 8) = o;
      if ((c & 3) == 0) mxs[((long)(c >> 4) * lds + r) * 4 + ((c >> 2) & 3)] = (unsigned char)sb;   // [D/128][lds rows][4]
    }
  }
}

template <int NC>
__global__ __launch_bounds__(256) void ld_layernorm_mx2_kernel(LnParams p, unsigned char* mxs, long lds) {
  const int lane = threadIdx.x & 63;
  const int r0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
  if (r0 >= p.rows) return;
  const bool has1 = r0 + 1 < p.rows;
  const int nchunk = p.D >> 3;
  u32x4_t raw[2][NC];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const bf16_t* xr = (const bf16_t*)p.x + (long)(r0 + (has1 ? j : 0)) * p.ldx;      // odd row count: the last wave does its row twice
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + 64 * i;
      raw[j][i] = (u32x4_t){0u, 0u, 0u, 0u};
      if (c < nchunk) raw[j][i] = *(const u32x4_t*)(xr + c * 8);
    }
  }
  const bf16_t* shift[2]; const bf16_t* scale[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = r0 + (has1 ? j : 0);
    const int bb = r / p.rows_per_batch;
    const bool txt = (r - bb * p.rows_per_batch) < p.text_len;
    shift[j] = p.mod + bb * p.mod_bstride + (txt ? p.shift_txt : p.shift_img);
    scale[j] = p.mod + bb * p.mod_bstride + (txt ? p.scale_txt : p.scale_img);
  }
  f32x2_t v[2][NC][4];
  float rstd[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {                      // (statistics exactly as ld_layernorm_mod_kernel takes them)
    f32x2_t s2 = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[j][i][e] = unpack_bf16x2(raw[j][i][e]); s2 += v[j][i][e]; }
    const float mean = wave_sum(s2[0] + s2[1]) / (float)p.D;
    const f32x2_t mean2 = {mean, mean};
    f32x2_t ss2 = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      if (lane + 64 * i < nchunk) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[j][i][e] -= mean2; ss2 += v[j][i][e] * v[j][i][e]; }
      }
    }
    rstd[j] = rsqrtf(wave_sum(ss2[0] + ss2[1]) / (float)p.D + p.eps);
  }
  const bf16_t* const shc[2] = {shift[0], shift[1]};
  const bf16_t* const scc[2] = {scale[0], scale[1]};
  if (shift[0] == shift[1] && scale[0] == scale[1]) ln_mod2_apply_mx<NC, true>(p, v, rstd, shc, scc, r0, has1, lane, nchunk, mxs, lds);
  else ln_mod2_apply_mx<NC, false>(p, v, rstd, shc, scc, r0, has1, lane, nchunk, mxs, lds);
}

// LayerNorm (+ modulate) with MXFP8 output: the activation of the next MXFP8 GEMM is quantised where it is produced.
template <int NC>   // chunks (of 8 elements) per lane
__global__ __launch_bounds__(256) void ld_layernorm_mx_kernel(LnParams p, unsigned char* mxs, long lds) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= p.rows) return;
  const int nchunk = p.D >> 3;
  float v[NC][8];
  float yv[NC][8], amaxv[NC];
  float s = 0.f, s_odd = 0.f;       // even / odd elements summed apart: the order of ld_layernorm_mod_kernel's packed sums
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      if (p.x_f32) {
        const float* xr = (const float*)p.x + (long)r * p.ldx + c * 8;
        const f32x4_t a = *(const f32x4_t*)xr, b4 = *(const f32x4_t*)(xr + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[i][e] = a[e]; v[i][4 + e] = b4[e]; }
      } else {
        const u32x4_t a = *(const u32x4_t*)((const bf16_t*)p.x + (long)r * p.ldx + c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[i][2 * e] = bf_lo(a[e]); v[i][2 * e + 1] = bf_hi(a[e]); }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { s += v[i][2 * e]; s_odd += v[i][2 * e + 1]; }
    }
  }
  const float mean = wave_sum(s + s_odd) / (float)p.D;
  float ss = 0.f, ss_odd = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    if (lane + 64 * i < nchunk) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d0 = v[i][2 * e] - mean, d1 = v[i][2 * e + 1] - mean;
        ss += d0 * d0; ss_odd += d1 * d1;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(ss + ss_odd) / (float)p.D + p.eps);
  const bf16_t* shift = nullptr; const bf16_t* scale = nullptr;
  if (p.mod) {
    const int bb = r / p.rows_per_batch;
    const bool txt = (r - bb * p.rows_per_batch) < p.text_len;
    shift = p.mod + bb * p.mod_bstride + (txt ? p.shift_txt : p.shift_img);
    scale = p.mod + bb * p.mod_bstride + (txt ? p.scale_txt : p.scale_img);
  }
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    amaxv[i] = 0.f;
    if (c >= nchunk) continue;
    float y[8];
    float wv[8], bv[8];
    if (p.w) {
      const u32x4_t ww = *(const u32x4_t*)(p.w + c * 8), bw = *(const u32x4_t*)(p.b + c * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) { wv[2 * e] = bf_lo(ww[e]); wv[2 * e + 1] = bf_hi(ww[e]); bv[2 * e] = bf_lo(bw[e]); bv[2 * e + 1] = bf_hi(bw[e]); }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = (v[i][e] - mean) * rstd;
      if (p.w) t = t * wv[e] + bv[e];
      y[e] = t;
    }
    if (shift) {
      const u32x4_t sh = *(const u32x4_t*)(shift + c * 8), sc = *(const u32x4_t*)(scale + c * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // modulate(): x * (1 + scale) + shift, every op in bf16 as in the reference
        const float s0 = rbf(1.0f + bf_lo(sc[e])), s1 = rbf(1.0f + bf_hi(sc[e]));
        y[2 * e] = rbf(rbf(rbf(y[2 * e]) * s0) + bf_lo(sh[e]));
        y[2 * e + 1] = rbf(rbf(rbf(y[2 * e + 1]) * s1) + bf_hi(sh[e]));
      }
    }
    // MXFP8 output: the bf16 value the plain kernel would store, quantised in place.  A 32-element block is the chunks of
    // four adjacent lanes (c = lane + 64 i); scale = smallest power of two >= amax / 448, as in ld_quant_mxfp8_kernel.
    float amax = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { y[e] = rbf(y[e]); amax = fmaxf(amax, fabsf(y[e])); }
    amaxv[i] = amax;
#pragma unroll
    for (int e = 0; e < 8; ++e) yv[i][e] = y[e];
  }
  // (second pass: the shuffles must be executed by all 64 lanes, including those past the row's last chunk)
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    float amax = c < nchunk ? amaxv[i] : 0.f;
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    if (c >= nchunk) continue;
    const uint32_t tb = __float_as_uint(amax * (1.0f / 448.0f));
    int sb = (int)((tb >> 23) & 0xffu) + ((tb & 0x7fffffu) != 0u ? 1 : 0);
    sb = amax > 0.f ? (sb < 1 ? 1 : (sb > 254 ? 254 : sb)) : 0;
    const float inv = __uint_as_float((uint32_t)(254 - sb) << 23);
    u32x2_t o;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      unsigned w = 0;
      w = __builtin_amdgcn_cvt_pk_fp8_f32(yv[i][4 * h] * inv, yv[i][4 * h + 1] * inv, w, false);     // <= 448 by construction
      w = __builtin_amdgcn_cvt_pk_fp8_f32(yv[i][4 * h + 2] * inv, yv[i][4 * h + 3] * inv, w, true);
      o[h] = w;
    }
    *(u32x2_t*)((unsigned char*)p.out + (long)r * p.ldo + c * 8) = o;
    if ((c & 3) == 0) mxs[((long)(c >> 4) * lds + r) * 4 + ((c >> 2) & 3)] = (unsigned char)sb;   // [D/128][lds rows][4]
  }
}


// ---------------------------------------------------------------------------------------------
// qkv [B*N][3*H*64] (thirds q|k|v) -> Q,K [B][H][Npad][64] and V^T [B][H][64][Npad].
// mode 0: per-head LayerNorm(64) on q and k (DiT);  mode 1: interleaved-pair RoPE with a [N][32] table (TiTok).
// One workgroup = 64 tokens x 1 head.
// ---------------------------------------------------------------------------------------------
struct SplitParams {
  const bf16_t* qkv; bf16_t* Q; bf16_t* K; bf16_t* Vt;
  const bf16_t* qw; const bf16_t* qb; const bf16_t* kw; const bf16_t* kb;
  const float* cos_t; const float* sin_t;
  int B, N, H, Npad, mode;
  float eps;
};

__global__ __launch_bounds__(256) void ld_qkv_split_kernel(SplitParams p) {
  __shared__ bf16_t vt[64][72];     // V tile [token][d] (+pad) for the transposed write
  const int tid = threadIdx.x;
  const int n0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  const long row_stride = 3L * p.H * 64;
  // ---- q and k: 8 lanes per (token, tensor), 8 elements per lane ----
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int task = it * 256 + tid;        // 0..1023 = 64 tokens x 2 tensors x 8 lanes
    const int sub = task & 7;
    const int which = (task >> 3) & 1;      // 0 = q, 1 = k
    const int tok = task >> 4;
    const int n = n0 + tok;
    float v[8];
    const bool valid = n < p.N;
    if (valid) {
      const u32x4_t a = *(const u32x4_t*)(p.qkv + ((long)b * p.N + n) * row_stride + (long)which * p.H * 64 + h * 64 + sub * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[2 * e] = bf_lo(a[e]); v[2 * e + 1] = bf_hi(a[e]); }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
    if (p.mode == 0) {
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[e];
      s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
      const float mean = s * (1.0f / 64.0f);
      float ss = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[e] - mean; ss += d * d; }
      ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 4, 64);
      const float rstd = rsqrtf(ss * (1.0f / 64.0f) + p.eps);
      const bf16_t* w = which ? p.kw : p.qw;
      const bf16_t* bb = which ? p.kb : p.qb;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (v[e] - mean) * rstd * bf2f(w[sub * 8 + e]) + bf2f(bb[sub * 8 + e]);
    } else if (valid) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float c = p.cos_t[(long)n * 32 + sub * 4 + e], s = p.sin_t[(long)n * 32 + sub * 4 + e];
        const float a0 = v[2 * e], a1 = v[2 * e + 1];
        v[2 * e] = a0 * c - a1 * s;
        v[2 * e + 1] = a0 * s + a1 * c;
      }
    }
    if (!valid) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;     // zero rows in the padding region
    }
    bf16_t* dst = (which ? p.K : p.Q) + (((long)b * p.H + h) * p.Npad + n) * 64 + sub * 8;
    u32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
    if (n < p.Npad) *(u32x4_t*)dst = o;
  }
  // ---- v: stage [64 tokens][64 d] in LDS, write transposed rows [d][64 tokens] ----
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int task = it * 256 + tid;        // 512 = 64 tokens x 8 chunks
    const int sub = task & 7, tok = task >> 3;
    const int n = n0 + tok;
    u32x4_t a = {0u, 0u, 0u, 0u};
    if (n < p.N) a = *(const u32x4_t*)(p.qkv + ((long)b * p.N + n) * row_stride + 2L * p.H * 64 + h * 64 + sub * 8);
    *(u32x4_t*)(&vt[tok][sub * 8]) = a;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int task = it * 256 + tid;        // 512 = 64 d x 8 token-chunks
    const int tc = task & 7, d = task >> 3;
    bf16_t tmp[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) tmp[e] = vt[tc * 8 + e][d];
    u32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (uint32_t)tmp[2 * e] | ((uint32_t)tmp[2 * e + 1] << 16);
    if (n0 + tc * 8 < p.Npad)
      *(u32x4_t*)(p.Vt + (((long)b * p.H + h) * 64 + d) * p.Npad + n0 + tc * 8) = o;
  }
}

// ---------------------------------------------------------------------------------------------
// GroupNorm statistics over a channels-last tensor x [F][P][C]: per (frame-group f, group g) sum and sum of
// squares in double.  No floating-point atomics anywhere: every workgroup writes its partial pair to
// partials [F][nblk][G][2] and ld_gn_stats_reduce_kernel sums them in a fixed order, so the statistics -- and with them the
// whole VAE / upsampler decode -- are bit-identical from run to run.
// ---------------------------------------------------------------------------------------------
constexpr int GN_ROWS_PER_BLOCK = 512;
constexpr int LD_GN_FOLD_BLOCKS = 256;          // ld_groupnorm_stats_from_conv: workgroups of the fold = double pairs per group in `partials`

__global__ __launch_bounds__(256) void ld_gn_stats_kernel(const bf16_t* x, double* partials, long P, int C, int G, int rows_per_block) {
  // Per-thread partials are combined in a FIXED order: the fp32 LDS atomics this replaced made the statistics differ by
  // ~6e-7 from run to run, which a 30-layer decoder amplifies into visible last-bit noise.
  __shared__ float part[2][256][4];   // [sum | sumsq][thread][sub-group of the thread's 8 channels]
  const int f = blockIdx.y;
  const long p0 = (long)blockIdx.x * rows_per_block;
  const int cpg = C / G;
  const int chunks_per_row = C >> 3;
  const int tid = threadIdx.x;
  // a thread owns one 8-channel chunk column and strides over rows
  const int chunk = tid % chunks_per_row;
  const int rlane = tid / chunks_per_row;
  const int rstep = 256 / chunks_per_row;
  const int nsub = cpg >= 8 ? 1 : 8 / cpg;       // groups inside one 8-channel chunk (cpg in {2,4} -> 4, 2)
  float gs[4] = {0.f, 0.f, 0.f, 0.f}, gss[4] = {0.f, 0.f, 0.f, 0.f};
  if (rlane < rstep) {
    // Round 5: GN_U rows are requested before the first is consumed (the loop used to hold ONE 16-byte load in flight per thread:
    // 2.4 TB/s); the rows are still accumulated one after the other in row order -- the same sums, bit for bit.
    constexpr int GN_U = 8;
    const long pend = min(p0 + rows_per_block, P);
    const bf16_t* base = x + (long)f * P * C + chunk * 8;
    auto accumulate = [&](const u32x4_t a) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[2 * e] = bf_lo(a[e]); v[2 * e + 1] = bf_hi(a[e]); }
      if (cpg >= 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { gs[0] += v[e]; gss[0] += v[e] * v[e]; }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) { gs[e / cpg % 4] += v[e]; gss[e / cpg % 4] += v[e] * v[e]; }
      }
    };
    long r = p0 + rlane;
    for (; r + (long)(GN_U - 1) * rstep < pend; r += (long)GN_U * rstep) {
      u32x4_t a[GN_U];
#pragma unroll
      for (int u = 0; u < GN_U; ++u) a[u] = __builtin_nontemporal_load((const u32x4_t*)(base + (r + (long)u * rstep) * C));
#pragma unroll
      for (int u = 0; u < GN_U; ++u) accumulate(a[u]);
    }
    for (; r < pend; r += rstep) accumulate(*(const u32x4_t*)(base + r * C));
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) { part[0][tid][q] = gs[q]; part[1][tid][q] = gss[q]; }
  __syncthreads();
  if (tid < G) {
    // group tid <- chunks [c0, c1), sub-slot q, all row lanes, in index order
    double s = 0.0, ss = 0.0;
    int c0, c1, q;
    if (cpg >= 8) { c0 = tid * (cpg >> 3); c1 = c0 + (cpg >> 3); q = 0; }
    else { c0 = tid / nsub; c1 = c0 + 1; q = tid % nsub; }
    for (int rl = 0; rl < rstep; ++rl)
      for (int c = c0; c < c1; ++c) {
        const int t = rl * chunks_per_row + c;
        s += (double)part[0][t][q]; ss += (double)part[1][t][q];
      }
    double* o = partials + (((long)f * gridDim.x + blockIdx.x) * G + tid) * 2;
    o[0] = s; o[1] = ss;
  }
}

// One wave per (f, g): lane l sums blocks l, l + 64, ... in index order, then a fixed butterfly.
__global__ __launch_bounds__(64) void ld_gn_stats_reduce_kernel(const double* partials, double* stats, int nblk, int G) {
  const int fg = blockIdx.x, f = fg / G, g = fg - f * G, lane = threadIdx.x;
  double s = 0.0, ss = 0.0;
  for (int b = lane; b < nblk; b += 64) {
    const double* q = partials + (((long)f * nblk + b) * G + g) * 2;
    s += q[0]; ss += q[1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); ss += __shfl_xor(ss, o, 64); }
  if (lane == 0) { stats[(long)fg * 2] = s; stats[(long)fg * 2 + 1] = ss; }
}

// The same statistics from what the producing convolution's epilogue left behind (ld_conv_cl_bf16_gn, ld_gemm.hip): fp32
// (sum, sum of squares) of every 64-row x 4-channel patch, part [U][C / 4][2].  Workgroup b folds units [b * upb, (b + 1) * upb)
// into double partials [nblk][G][2] -- thread = (row lane, quad), coalesced rows of C / 4 pairs, row lanes and quads summed in
// index order -- and ld_gn_stats_reduce_kernel finishes as above.  11 MB instead of the 708 MB activation at 8 x 480 x 720 x 128.
__global__ __launch_bounds__(256) void ld_gn_fold_partials_kernel(const float* part, double* partials, int U, int C4, int G, int upb) {
  __shared__ double sh[2][256];
  const int tid = threadIdx.x;
  const int quad = tid % C4, ul = tid / C4, ustep = 256 / C4;
  const int u1 = min((int)(blockIdx.x + 1) * upb, U);
  double s = 0.0, ss = 0.0;
  for (int u = blockIdx.x * upb + ul; u < u1; u += ustep) {
    const ld_f32x2_t v = *(const ld_f32x2_t*)(part + ((long)u * C4 + quad) * 2);
    s += (double)v[0]; ss += (double)v[1];
  }
  sh[0][tid] = s; sh[1][tid] = ss;
  __syncthreads();
  if (tid < G) {
    const int qpg = C4 / G;                                 // quads per group
    double a = 0.0, b = 0.0;
    for (int l = 0; l < ustep; ++l)
      for (int q = tid * qpg; q < (tid + 1) * qpg; ++q) { a += sh[0][l * C4 + q]; b += sh[1][l * C4 + q]; }
    double* o = partials + ((long)blockIdx.x * G + tid) * 2;
    o[0] = a; o[1] = b;
  }
}

// ---------------------------------------------------------------------------------------------
// Apply GroupNorm (+ optional SpatialNorm scale/shift from low-res zq convs) + optional swish and write the
// result into the interior of a zero-bordered channels-last buffer (the conv kernel's input layout).
//   x [F*T][H][W][C]  ->  out [F][T + tpad][H + 2*hpad][W + 2*wpad][Cout_stride]  at (t + tpad, h + hpad, w + wpad)
// ---------------------------------------------------------------------------------------------
struct GnApplyParams {
  const bf16_t* x; bf16_t* out;
  const double* stats; const bf16_t* gamma; const bf16_t* beta;
  const bf16_t* zy; const bf16_t* zb;     // [Tz][Hz][Wz][C] or null
  int F, T, H, W, C, G;
  int Tz, Hz, Wz;
  int tpad, hpad, wpad;
  int swish;
  float eps;
  double inv_count;
  int fast_rcp;          // sigmoid's reciprocal as v_rcp_f32 (1 ulp, before the bf16 rounding) instead of the IEEE division sequence
  int wshift;            // W == Wz << wshift: the nearest-neighbour zq column is a shift (-1: the general rule, a division per position)
};

__global__ __launch_bounds__(256) void ld_gn_apply_kernel(GnApplyParams p) {
  __shared__ float s_mean[512], s_rstd[512];      // per (frame-group, group)
  for (int i = threadIdx.x; i < p.F * p.G; i += 256) {
    const double mean = p.stats[2 * i] * p.inv_count;
    const double var = p.stats[2 * i + 1] * p.inv_count - mean * mean;
    s_mean[i] = (float)mean;
    s_rstd[i] = rsqrtf(fmaxf((float)var, 0.f) + p.eps);
  }
  __syncthreads();
  const int chunks = p.C >> 3;
  const long total = (long)p.F * p.T * p.H * p.W * chunks;
  const int cpg = p.C / p.G;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int chunk = (int)(i % chunks);
    long pos = i / chunks;
    const int w = (int)(pos % p.W); pos /= p.W;
    const int h = (int)(pos % p.H); pos /= p.H;
    const int t = (int)(pos % p.T);
    const int f = (int)(pos / p.T);
    const u32x4_t a = *(const u32x4_t*)(p.x + (i / chunks) * p.C + chunk * 8);
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = bf_lo(a[e]); v[2 * e + 1] = bf_hi(a[e]); }
    const u32x4_t gw = *(const u32x4_t*)(p.gamma + chunk * 8), bw = *(const u32x4_t*)(p.beta + chunk * 8);
    float sy[8], sb[8];
    if (p.zy) {
      int tz;
      if (p.T > 1 && (p.T & 1)) tz = (t == 0) ? 0 : 1 + (int)(((long)(t - 1) * (p.Tz - 1)) / (p.T - 1));
      else tz = (int)(((long)t * p.Tz) / p.T);
      const int hz = (int)(((long)h * p.Hz) / p.H), wz = (int)(((long)w * p.Wz) / p.W);
      const long zo = (((long)tz * p.Hz + hz) * p.Wz + wz) * p.C + chunk * 8;
      const u32x4_t yw = *(const u32x4_t*)(p.zy + zo), zw = *(const u32x4_t*)(p.zb + zo);
#pragma unroll
      for (int e = 0; e < 4; ++e) { sy[2 * e] = bf_lo(yw[e]); sy[2 * e + 1] = bf_hi(yw[e]); sb[2 * e] = bf_lo(zw[e]); sb[2 * e + 1] = bf_hi(zw[e]); }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int g = (chunk * 8 + e) / cpg;
      const float mean = s_mean[f * p.G + g], rstd = s_rstd[f * p.G + g];
      const float gm = (e & 1) ? bf_hi(gw[e >> 1]) : bf_lo(gw[e >> 1]);
      const float bt = (e & 1) ? bf_hi(bw[e >> 1]) : bf_lo(bw[e >> 1]);
      float y = rbf((v[e] - mean) * rstd * gm + bt);      // GroupNorm output (bf16)
      if (p.zy) y = rbf(rbf(y * sy[e]) + sb[e]);                  // norm_f * conv_y(zq) + conv_b(zq)
      if (p.swish) y = rbf(y * rbf(p.fast_rcp ? __builtin_amdgcn_rcpf(1.0f + __expf(-y)) : 1.0f / (1.0f + __expf(-y))));  // x * sigmoid(x)
      v[e] = y;
    }
    const long Tp = p.T + p.tpad, Hp = p.H + 2 * p.hpad, Wp = p.W + 2 * p.wpad;
    const long o = ((((long)f * Tp + t + p.tpad) * Hp + h + p.hpad) * Wp + w + p.wpad) * p.C + chunk * 8;
    u32x4_t ow;
#pragma unroll
    for (int e = 0; e < 4; ++e) ow[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
    *(u32x4_t*)(p.out + o) = ow;
  }
}

// The same operation for channel counts whose 16-byte chunks divide the workgroup (C / 8 | 256: every VAE / upsampler
// width): one (frame-group, t, h) row of W positions per workgroup, thread -> fixed chunk, so the per-channel operands
// (gamma, beta, the group's mean / rstd) are unpacked once per thread and no index needs a division except the nearest-
// neighbour column of zq; element pairs for the bf16 roundings.  (The flat kernel above decomposes a 64-bit linear index
// per chunk and divides by the group width per element: several times the arithmetic of the normalisation itself.)
__global__ __launch_bounds__(256) void ld_gn_apply_rows_kernel(GnApplyParams p) {
  const int chunks = p.C >> 3;
  const int tid = threadIdx.x;
  const int chunk = tid % chunks, wstep = 256 / chunks;
  const int row = blockIdx.x;                         // (f, t, h)
  const int h = row % p.H, ft = row / p.H;
  const int t = ft % p.T, f = ft / p.T;
  const int cpg = p.C / p.G;
  ld_f32x2_t mean2[4], rstd2[4], gm2[4], bt2[4];
  {
    const u32x4_t gw = *(const u32x4_t*)(p.gamma + chunk * 8), bw = *(const u32x4_t*)(p.beta + chunk * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      gm2[e] = unpack_bf16x2(gw[e]); bt2[e] = unpack_bf16x2(bw[e]);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int i = f * p.G + (chunk * 8 + 2 * e + k) / cpg;
        const double mean = p.stats[2 * i] * p.inv_count;
        const double var = p.stats[2 * i + 1] * p.inv_count - mean * mean;
        mean2[e][k] = (float)mean;
        rstd2[e][k] = rsqrtf(fmaxf((float)var, 0.f) + p.eps);
      }
    }
  }
  int tz = 0, hz = 0;
  if (p.zy) {
    if (p.T > 1 && (p.T & 1)) tz = (t == 0) ? 0 : 1 + (int)(((long)(t - 1) * (p.Tz - 1)) / (p.T - 1));
    else tz = (int)(((long)t * p.Tz) / p.T);
    hz = (int)(((long)h * p.Hz) / p.H);
  }
  const bf16_t* xrow = p.x + (long)row * p.W * p.C + chunk * 8;
  const long Tp = p.T + p.tpad, Hp = p.H + 2 * p.hpad, Wp = p.W + 2 * p.wpad;
  bf16_t* orow = p.out + ((((long)f * Tp + t + p.tpad) * Hp + h + p.hpad) * Wp + p.wpad) * p.C + chunk * 8;
  const bf16_t* zyrow = p.zy ? p.zy + ((long)tz * p.Hz + hz) * p.Wz * p.C + chunk * 8 : nullptr;
  const bf16_t* zbrow = p.zy ? p.zb + ((long)tz * p.Hz + hz) * p.Wz * p.C + chunk * 8 : nullptr;
  // Round 5: GA_U positions per trip, all of their loads (x from HBM, the two zq rows from L2) requested before the first is
  // normalised -- the loop used to wait for one 16-byte load at a time.  Element-wise work: the same bits.
  constexpr int GA_U = 4;
  auto finish = [&](const u32x4_t a, const u32x4_t yw, const u32x4_t zw, int w) {
    u32x4_t ow;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ld_f32x2_t y = rbf2((unpack_bf16x2(a[e]) - mean2[e]) * rstd2[e] * gm2[e] + bt2[e]);      // GroupNorm output (bf16)
      if (p.zy) y = rbf2(rbf2(y * unpack_bf16x2(yw[e])) + unpack_bf16x2(zw[e]));              // norm_f * conv_y(zq) + conv_b(zq)
      if (p.swish) {                                                                                     // x * sigmoid(x)
        const ld_f32x2_t den = {1.0f + __expf(-y[0]), 1.0f + __expf(-y[1])};
        // Round 5: 1 / den as v_rcp_f32.  The IEEE division the compiler emits for `1.0f / x` is ten VALU instructions, eight times
        // per 16-byte chunk: together more than the rest of the normalisation, and enough to make this kernel VALU-bound
        // (3.8 TB/s of read + write at 480 x 720).  The quotient is rounded to bf16 right away; LD_GN_FAST_RCP=0: the division.
        const ld_f32x2_t sg = p.fast_rcp ? (ld_f32x2_t){__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])}
                                         : (ld_f32x2_t){1.0f / den[0], 1.0f / den[1]};
        y = y * rbf2(sg);
      }
      ow[e] = pack_bf16x2(y);
    }
    *(u32x4_t*)(orow + (long)w * p.C) = ow;
  };
  int w = tid / chunks;
  for (; w + (GA_U - 1) * wstep < p.W; w += GA_U * wstep) {
    u32x4_t a[GA_U], yw[GA_U], zw[GA_U];
#pragma unroll
    for (int u = 0; u < GA_U; ++u) {
      const int wu = w + u * wstep;
      a[u] = __builtin_nontemporal_load((const u32x4_t*)(xrow + (long)wu * p.C));
      yw[u] = zw[u] = (u32x4_t){0u, 0u, 0u, 0u};
      if (p.zy) {
        const int wz = p.wshift >= 0 ? (wu >> p.wshift) : (int)(((long)wu * p.Wz) / p.W);
        yw[u] = *(const u32x4_t*)(zyrow + (long)wz * p.C); zw[u] = *(const u32x4_t*)(zbrow + (long)wz * p.C);
      }
    }
#pragma unroll
    for (int u = 0; u < GA_U; ++u) finish(a[u], yw[u], zw[u], w + u * wstep);
  }
  for (; w < p.W; w += wstep) {
    const u32x4_t a = *(const u32x4_t*)(xrow + (long)w * p.C);
    u32x4_t yw = (u32x4_t){0u, 0u, 0u, 0u}, zw = yw;
    if (p.zy) {
      const int wz = p.wshift >= 0 ? (w >> p.wshift) : (int)(((long)w * p.Wz) / p.W);
      yw = *(const u32x4_t*)(zyrow + (long)wz * p.C); zw = *(const u32x4_t*)(zbrow + (long)wz * p.C);
    }
    finish(a, yw, zw, w);
  }
}

}  // namespace

LD_API int ld_layernorm(const void* x, int64_t ldx, int32_t x_f32, const void* w, const void* b, void* out, int64_t ldo,
                        int32_t out_f32, int64_t rows, int64_t D, float eps, const void* mod, int64_t mod_bstride,
                        int64_t shift_img, int64_t scale_img, int64_t shift_txt, int64_t scale_txt,
                        int64_t rows_per_batch, int64_t text_len, void* stream) {
  LD_REQUIRE(x && out, "ld_layernorm: null pointer");
  LD_REQUIRE(D % 8 == 0 && D <= 4096 && D > 0, "ld_layernorm: D=%ld must be a multiple of 8 and <= 4096", (long)D);
  LD_REQUIRE((w == nullptr) == (b == nullptr), "ld_layernorm: weight and bias go together");
  LnParams p{};
  p.x = x; p.out = out; p.w = (const bf16_t*)w; p.b = (const bf16_t*)b; p.mod = (const bf16_t*)mod;
  p.ldx = ldx; p.ldo = ldo; p.rows = (int)rows; p.D = (int)D; p.eps = eps; p.x_f32 = x_f32; p.out_f32 = out_f32;
  p.rows_per_batch = rows_per_batch > 0 ? (int)rows_per_batch : (1 << 30); p.text_len = (int)text_len;
  p.mod_bstride = mod_bstride; p.shift_img = shift_img; p.scale_img = scale_img; p.shift_txt = shift_txt; p.scale_txt = scale_txt;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  const int nc = (int)((D / 8 + 63) / 64);
  static const bool fast_ok = !(getenv("LD_LN_FAST") && atoi(getenv("LD_LN_FAST")) == 0);
  if (fast_ok && mod && w && !x_f32 && !out_f32 && nc <= 4) {        // the DiT's block LayerNorms: two rows per wave
    grid = dim3((unsigned)((rows + 7) / 8));
    switch (nc) {
      case 1: hipLaunchKernelGGL(ld_layernorm_mod_kernel<1>, grid, block, 0, st, p); break;
      case 2: hipLaunchKernelGGL(ld_layernorm_mod_kernel<2>, grid, block, 0, st, p); break;
      case 3: hipLaunchKernelGGL(ld_layernorm_mod_kernel<3>, grid, block, 0, st, p); break;
      default: hipLaunchKernelGGL(ld_layernorm_mod_kernel<4>, grid, block, 0, st, p); break;
    }
    return ld_check_launch("ld_layernorm");
  }
  switch (nc) {
    case 1: hipLaunchKernelGGL(ld_layernorm_kernel<1>, grid, block, 0, st, p); break;
    case 2: hipLaunchKernelGGL(ld_layernorm_kernel<2>, grid, block, 0, st, p); break;
    case 3: hipLaunchKernelGGL(ld_layernorm_kernel<3>, grid, block, 0, st, p); break;
    case 4: hipLaunchKernelGGL(ld_layernorm_kernel<4>, grid, block, 0, st, p); break;
    default: hipLaunchKernelGGL(ld_layernorm_kernel<8>, grid, block, 0, st, p); break;
  }
  return ld_check_launch("ld_layernorm");
}

LD_API int ld_qkv_split(const void* qkv, void* Q, void* K, void* Vt, int64_t B, int64_t N, int64_t H, int64_t Npad,
                        int32_t mode, const void* q_w, const void* q_b, const void* k_w, const void* k_b, float eps,
                        const float* cos_t, const float* sin_t, void* stream) {
  LD_REQUIRE(qkv && Q && K && Vt, "ld_qkv_split: null pointer");
  LD_REQUIRE(Npad % 64 == 0 && Npad >= N, "ld_qkv_split: Npad must be a multiple of 64 and >= N");
  LD_REQUIRE(mode == 0 ? (q_w && q_b && k_w && k_b) : (cos_t && sin_t), "ld_qkv_split: missing LN weights / RoPE table");
  SplitParams p{};
  p.qkv = (const bf16_t*)qkv; p.Q = (bf16_t*)Q; p.K = (bf16_t*)K; p.Vt = (bf16_t*)Vt;
  p.qw = (const bf16_t*)q_w; p.qb = (const bf16_t*)q_b; p.kw = (const bf16_t*)k_w; p.kb = (const bf16_t*)k_b;
  p.cos_t = cos_t; p.sin_t = sin_t; p.B = (int)B; p.N = (int)N; p.H = (int)H; p.Npad = (int)Npad; p.mode = mode; p.eps = eps;
  dim3 grid((unsigned)(Npad / 64), (unsigned)H, (unsigned)B), block(256);
  hipLaunchKernelGGL(ld_qkv_split_kernel, grid, block, 0, (hipStream_t)stream, p);
  return ld_check_launch("ld_qkv_split");
}

LD_API int64_t ld_groupnorm_stats_blocks(int64_t P) { return (P + GN_ROWS_PER_BLOCK - 1) / GN_ROWS_PER_BLOCK; }

LD_API int ld_groupnorm_stats(const void* x, double* stats, double* partials, int64_t F, int64_t P, int64_t C, int64_t G, void* stream) {
  LD_REQUIRE(x && stats && partials, "ld_groupnorm_stats: null pointer");
  LD_REQUIRE(C % 8 == 0 && C / 8 <= 256 && G <= 64 && C % G == 0, "ld_groupnorm_stats: unsupported C=%ld G=%ld", (long)C, (long)G);
  const int cpg = (int)(C / G);
  LD_REQUIRE(cpg % 8 == 0 || 8 % cpg == 0, "ld_groupnorm_stats: channels per group %d unsupported", cpg);
  const int rows_per_block = GN_ROWS_PER_BLOCK;
  const int nblk = (int)ld_groupnorm_stats_blocks(P);
  dim3 grid((unsigned)nblk, (unsigned)F), block(256);
  hipLaunchKernelGGL(ld_gn_stats_kernel, grid, block, 0, (hipStream_t)stream, (const bf16_t*)x, partials, (long)P, (int)C, (int)G, rows_per_block);
  hipLaunchKernelGGL(ld_gn_stats_reduce_kernel, dim3((unsigned)(F * G)), dim3(64), 0, (hipStream_t)stream, (const double*)partials, stats, nblk, (int)G);
  return ld_check_launch("ld_groupnorm_stats");
}

LD_API int ld_groupnorm_stats_from_conv(const float* gn_partials, double* stats, double* partials, int64_t P, int64_t C, int64_t G,
                                        void* stream) {
  LD_REQUIRE(gn_partials && stats && partials, "ld_groupnorm_stats_from_conv: null pointer");
  LD_REQUIRE(P > 0 && P < (1LL << 31) && C % 4 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0 && G > 0 && G <= 64 && (C / 4) % G == 0,
             "ld_groupnorm_stats_from_conv: unsupported P=%ld C=%ld G=%ld (C / 4 a power of two <= 256, whole quads per group)", (long)P,
             (long)C, (long)G);
  const int U = (int)((P + 63) / 64), C4 = (int)(C / 4);
  const int rows_at_once = 256 / C4;
  int upb = (U + LD_GN_FOLD_BLOCKS - 1) / LD_GN_FOLD_BLOCKS;                          // at most LD_GN_FOLD_BLOCKS workgroups ...
  upb = ((upb + rows_at_once - 1) / rows_at_once) * rows_at_once;                      // ... of whole trips
  const int nblk = (U + upb - 1) / upb;
  hipLaunchKernelGGL(ld_gn_fold_partials_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, gn_partials, partials, U, C4, (int)G, upb);
  hipLaunchKernelGGL(ld_gn_stats_reduce_kernel, dim3((unsigned)G), dim3(64), 0, (hipStream_t)stream, (const double*)partials, stats, nblk, (int)G);
  return ld_check_launch("ld_groupnorm_stats_from_conv");
}

LD_API int ld_groupnorm_apply(const void* x, void* out_padded, const double* stats, const void* gamma, const void* beta,
                              const void* zy, const void* zb, int64_t F, int64_t T, int64_t H, int64_t W, int64_t C,
                              int64_t G, int64_t Tz, int64_t Hz, int64_t Wz, int64_t tpad, int64_t hpad, int64_t wpad,
                              int32_t swish, float eps, void* stream) {
  LD_REQUIRE(x && out_padded && stats && gamma && beta, "ld_groupnorm_apply: null pointer");
  LD_REQUIRE((zy == nullptr) == (zb == nullptr), "ld_groupnorm_apply: zy and zb go together");
  LD_REQUIRE(C % 8 == 0 && C % G == 0, "ld_groupnorm_apply: bad channel count");
  LD_REQUIRE(F * G <= 512, "ld_groupnorm_apply: F*G=%ld exceeds 512", (long)(F * G));
  GnApplyParams p{};
  p.x = (const bf16_t*)x; p.out = (bf16_t*)out_padded; p.stats = stats; p.gamma = (const bf16_t*)gamma; p.beta = (const bf16_t*)beta;
  p.zy = (const bf16_t*)zy; p.zb = (const bf16_t*)zb;
  p.F = (int)F; p.T = (int)T; p.H = (int)H; p.W = (int)W; p.C = (int)C; p.G = (int)G;
  p.Tz = (int)Tz; p.Hz = (int)Hz; p.Wz = (int)Wz; p.tpad = (int)tpad; p.hpad = (int)hpad; p.wpad = (int)wpad;
  p.swish = swish; p.eps = eps;
  static int k_rcp = LD_KNOB_UNSET;
  p.fast_rcp = ld_knob("LD_GN_FAST_RCP", 1, &k_rcp) != 0;
  p.wshift = -1;
  if (zy && Wz > 0 && W % Wz == 0) {
    const long r = W / Wz;
    if ((r & (r - 1)) == 0) { p.wshift = 0; while ((1L << p.wshift) < r) ++p.wshift; }
  }
  p.inv_count = 1.0 / ((double)T * H * W * (C / G));
  const long total = F * T * H * W * (C / 8);
  const long blocks = (total + 255) / 256;
  dim3 grid((unsigned)(blocks < 8192 ? blocks : 8192)), block(256);
  static const bool rows_ok = !(getenv("LD_GN_ROWS") && atoi(getenv("LD_GN_ROWS")) == 0);
  if (rows_ok && 256 % (C / 8) == 0 && F * T * H < (1l << 30))
    hipLaunchKernelGGL(ld_gn_apply_rows_kernel, dim3((unsigned)(F * T * H)), block, 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(ld_gn_apply_kernel, grid, block, 0, (hipStream_t)stream, p);
  return ld_check_launch("ld_groupnorm_apply");
}

LD_API int ld_layernorm_mxfp8(const void* x, int64_t ldx, const void* w, const void* b, void* q, int64_t ldq, void* scales,
                              int64_t lds, int64_t rows, int64_t D, float eps, const void* mod, int64_t mod_bstride,
                              int64_t shift_img, int64_t scale_img, int64_t shift_txt, int64_t scale_txt,
                              int64_t rows_per_batch, int64_t text_len, void* stream) {
  LD_REQUIRE(x && q && scales, "ld_layernorm_mxfp8: null pointer");
  LD_REQUIRE(D % 128 == 0 && D <= 2048 && D > 0 && ldq % 8 == 0 && lds >= rows, "ld_layernorm_mxfp8: D=%ld must be a multiple of 128 and <= 2048, lds >= rows", (long)D);
  LD_REQUIRE((w == nullptr) == (b == nullptr), "ld_layernorm_mxfp8: weight and bias go together");
  LnParams p{};
  p.x = x; p.out = q; p.w = (const bf16_t*)w; p.b = (const bf16_t*)b; p.mod = (const bf16_t*)mod;
  p.ldx = ldx; p.ldo = ldq; p.rows = (int)rows; p.D = (int)D; p.eps = eps; p.x_f32 = 0; p.out_f32 = 0;
  p.rows_per_batch = rows_per_batch > 0 ? (int)rows_per_batch : (1 << 30); p.text_len = (int)text_len;
  p.mod_bstride = mod_bstride; p.shift_img = shift_img; p.scale_img = scale_img; p.shift_txt = shift_txt; p.scale_txt = scale_txt;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  const int nc = (int)((D / 8 + 63) / 64);
  static const bool fast_ok = !(getenv("LD_LN_FAST") && atoi(getenv("LD_LN_FAST")) == 0);
  if (fast_ok && mod && w) {                          // the DiT's block LayerNorms: two rows per wave, packed math
    grid = dim3((unsigned)((rows + 7) / 8));
    switch (nc) {
      case 1: hipLaunchKernelGGL(ld_layernorm_mx2_kernel<1>, grid, block, 0, st, p, (unsigned char*)scales, (long)lds); break;
      case 2: hipLaunchKernelGGL(ld_layernorm_mx2_kernel<2>, grid, block, 0, st, p, (unsigned char*)scales, (long)lds); break;
      case 3: hipLaunchKernelGGL(ld_layernorm_mx2_kernel<3>, grid, block, 0, st, p, (unsigned char*)scales, (long)lds); break;
      default: hipLaunchKernelGGL(ld_layernorm_mx2_kernel<4>, grid, block, 0, st, p, (unsigned char*)scales, (long)lds); break;
    }
    return ld_check_launch("ld_layernorm_mxfp8");
  }
  switch (nc) {
    case 1: hipLaunchKernelGGL(ld_layernorm_mx_kernel<1>, grid, block, 0, st, p, (unsigned char*)scales, (long)lds); break;
    case 2: hipLaunchKernelGGL(ld_layernorm_mx_kernel<2>, grid, block, 0, st, p, (unsigned char*)scales, (long)lds); break;
    case 3: hipLaunchKernelGGL(ld_layernorm_mx_kernel<3>, grid, block, 0, st, p, (unsigned char*)scales, (long)lds); break;
    default: hipLaunchKernelGGL(ld_layernorm_mx_kernel<4>, grid, block, 0, st, p, (unsigned char*)scales, (long)lds); break;
  }
  return ld_check_launch("ld_layernorm_mxfp8");
}

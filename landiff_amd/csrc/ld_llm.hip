// HBM-bound small-batch kernels: weight-streaming GEMV (batch <= 4), RMSNorm, RoPE + KV append +
// KV attention (head_dim 128), embedding gather, logits -> probabilities.
//
// Replaces: LlamaTransformerBlock / TransformerBlock.local_kvcache_inference
// (landiff/llm/modules/transformer_blocks.py:22-40,67-88,128-236), GPT.sample (landiff/llm/models/transformer.py:91-119),
// apply_rope (landiff/modules/pos_emb.py:16-46), the per-step logits math of Semantic1DLM.sample
// (landiff/llm/models/lm_model.py:417-454), and the batch-2 Linear layers of the DiT's conditioning path
// (time_embed / adaLN_modulation, landiff/diffusion/dit_video_concat.py:568,764-768).
//
// MI355X notes: the decode step is pure weight streaming (4.06 GB bf16 per step); every weight row is
// read exactly once with 16-byte loads straight into VGPRs (no LDS round trip for an operand that is
// not shared across waves), one wave64 per output row, activations (<= 4 x K bf16) stay in L1/L2.
// All per-step scalars (position, token) live in device memory so the whole step is graph-capturable.
#include "ld_common.h"
#include "ld_llm_dev.h"
#include <stdio.h>
#include <stdlib.h>
#include "../../include/landiff_hip.h"

namespace {

constexpr int MAXB = 4;

// ---------------------------------------------------------------------------------------------
// GEMV: y[b][n] = epi( sum_k in_act(x[b][k]) * W[n][k] ), one wave per output row.
// ---------------------------------------------------------------------------------------------
struct GemvParams {
  const void* x;        // [B][ldx] bf16 or f32
  const void* W;        // [N][K] bf16 or f32
  const void* W2;       // optional second matrix (gated MLP: act(W x) * (W2 x))
  const bf16_t* bias;   // [N] or null
  const void* resid;    // [B][ldr] or null (bf16, or f32 when out_f32)
  void* out;            // [B][ldo]
  int B, N, K;
  long ldx, ldo, ldr;
  int x_f32, w_f32, out_f32;
  int in_act, act;
  const float* norm_w;  // optional fused RMSNorm of x (bf16 x only): xn = bf16(x * rsqrt(mean(x^2)+eps) * norm_w)
  float norm_eps;
  ChainSync cs;         // dependent-launch form (ld_gemv_reg_kernel<..., CHAIN = true> only)
};

// Weight-streaming GEMV.  One workgroup = 4 waves x R rows.  The activation rows (B x K, optionally RMS-normalised,
// optionally passed through in_act) are staged ONCE per workgroup in LDS as bf16/fp32 -- letting every wave re-read x
// from L2 made all CUs hammer the same few cache lines -- and the first trip's weight loads are issued BEFORE the
// staging so the HBM latency of the weights overlaps it.  Each lane keeps R*U 16-byte weight loads in flight.
template <int B, bool WF32, int R, int U = 4>     // U: 16-byte loads per lane, row and trip (short rows, K <= 512: U = 1 and more rows per wave)
__global__ __launch_bounds__(256) void ld_gemv_kernel(GemvParams p) {
  extern __shared__ __attribute__((aligned(16))) char gsm[];   // x staged: [B][K] (bf16, or f32 when WF32) + 64 B scratch
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool gated = !WF32 && p.W2 != nullptr;       // gated form runs with R == 1: rows n0 of W and of W2
  const int nchunk = p.K >> 3;                       // 8 elements per chunk
  const int nrow = p.N;

  // work item = (row group, K trip); a wave walks its items in order, always holding the NEXT item's weight vectors
  // in registers (raw 16-byte loads, converted to fp32 only when consumed)
  const int ntrip = (nchunk + 64 * U - 1) / (64 * U);
  const int ngroups = (nrow + R - 1) / R;
  const int wave_global = blockIdx.x * 4 + wave, nwaves = gridDim.x * 4;
  // epilogue operands of an item (lane r < R owns output row grp * R + r): requested with the item's last weight trip, so
  // that their latency is not paid per item after the reduction (it was: 1.3 TB/s on short rows)
  float e_bias_n = 0.f, e_res_n[B];
#pragma unroll
  for (int b = 0; b < B; ++b) e_res_n[b] = 0.f;
  auto load_w = [&](u32x4_t (&w)[R][U], u32x4_t (&w2)[U], int grp, int trip) {
    if (trip == ntrip - 1 && lane < R && grp < ngroups && grp * R + lane < nrow) {
      const int n = grp * R + lane;
      if (p.bias) e_bias_n = bf2f(p.bias[n]);
      if (p.resid) {
#pragma unroll
        for (int b = 0; b < B; ++b)
          e_res_n[b] = p.out_f32 ? ((const float*)p.resid)[b * p.ldr + n] : bf2f(((const bf16_t*)p.resid)[b * p.ldr + n]);
      }
    }
    const int c0 = lane + trip * 64 * U;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + 64 * u;
      const bool ok = (c < nchunk) && (grp < ngroups);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int n = (grp * R + r < nrow) ? grp * R + r : nrow - 1;
        w[r][u] = (u32x4_t){0u, 0u, 0u, 0u};
        if (ok) {
          if (WF32) {
            // fp32 weights: two 16-byte halves are fetched at use time (head GEMV only; not worth double buffering)
          } else {
            w[r][u] = __builtin_nontemporal_load((const u32x4_t*)((const bf16_t*)p.W + (long)n * p.K + c * 8));
          }
        }
      }
      if (gated) {
        const int n = (grp < nrow) ? grp : nrow - 1;
        w2[u] = (u32x4_t){0u, 0u, 0u, 0u};
        if (ok) w2[u] = __builtin_nontemporal_load((const u32x4_t*)((const bf16_t*)p.W2 + (long)n * p.K + c * 8));
      }
    }
  };

  u32x4_t wn[R][U], w2n[U];
  int grp = wave_global, trip = 0;
  load_w(wn, w2n, grp, trip);                        // first item in flight while x is staged

  // ---- stage x (with the optional fused RMSNorm / input activation) ----
  float* red = (float*)(gsm + (size_t)B * p.K * (WF32 ? 4 : 2));
  float rs[B];
#pragma unroll
  for (int b = 0; b < B; ++b) rs[b] = 1.f;
  if (p.norm_w) {                                    // RMSNorm scale per row (transformer_blocks.py:22-40)
    float ss[B];
#pragma unroll
    for (int b = 0; b < B; ++b) {
      ss[b] = 0.f;
      for (int c = tid; c < nchunk; c += 256) {
        const u32x4_t a = *(const u32x4_t*)((const bf16_t*)p.x + b * p.ldx + c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float lo = bf_lo(a[e]), hi = bf_hi(a[e]); ss[b] += lo * lo + hi * hi; }
      }
      ss[b] = wave_sum(ss[b]);
      if (lane == 0) red[wave * B + b] = ss[b];
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < B; ++b)
      rs[b] = rsqrtf((red[b] + red[B + b] + red[2 * B + b] + red[3 * B + b]) / (float)p.K + p.norm_eps);
  }
  for (int i = tid; i < B * nchunk; i += 256) {
    const int b = i / nchunk, c = i - b * nchunk;
    float xv[8];
    if (p.x_f32) {
      const f32x4_t a = *(const f32x4_t*)((const float*)p.x + b * p.ldx + c * 8);
      const f32x4_t b4 = *(const f32x4_t*)((const float*)p.x + b * p.ldx + c * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { xv[e] = a[e]; xv[4 + e] = b4[e]; }
    } else {
      const u32x4_t a = *(const u32x4_t*)((const bf16_t*)p.x + b * p.ldx + c * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) { xv[2 * e] = bf_lo(a[e]); xv[2 * e + 1] = bf_hi(a[e]); }
    }
    if (p.in_act) {
#pragma unroll
      for (int e = 0; e < 8; ++e) xv[e] = rbf(apply_act(p.in_act, xv[e]));
    }
    if (p.norm_w) {
      const f32x4_t g0 = *(const f32x4_t*)(p.norm_w + c * 8), g1 = *(const f32x4_t*)(p.norm_w + c * 8 + 4);
      float r1 = rs[0];
#pragma unroll
      for (int bb = 1; bb < B; ++bb) r1 = (b == bb) ? rs[bb] : r1;
#pragma unroll
      for (int e = 0; e < 4; ++e) { xv[e] = xv[e] * r1 * g0[e]; xv[4 + e] = xv[4 + e] * r1 * g1[e]; }
    }
    if (WF32) {
      float* d = (float*)gsm + (long)b * p.K + c * 8;
      *(f32x4_t*)d = (f32x4_t){xv[0], xv[1], xv[2], xv[3]};
      *(f32x4_t*)(d + 4) = (f32x4_t){xv[4], xv[5], xv[6], xv[7]};
    } else {
      u32x4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(xv[2 * e], xv[2 * e + 1]);     // bf16 rounding of the normed x
      *(u32x4_t*)((bf16_t*)gsm + (long)b * p.K + c * 8) = o;
    }
  }
  __syncthreads();

  float acc[R][B], acc2[B];
  while (grp < ngroups) {
    u32x4_t wc[R][U], w2c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int r = 0; r < R; ++r) wc[r][u] = wn[r][u];
      w2c[u] = w2n[u];
    }
    const int cgrp = grp, ctrip = trip;
    const float e_bias = e_bias_n;
    float e_res[B];
#pragma unroll
    for (int b = 0; b < B; ++b) e_res[b] = e_res_n[b];
    if (++trip == ntrip) { trip = 0; grp += nwaves; }
    load_w(wn, w2n, grp, trip);                       // prefetch the next item
    if (ctrip == 0) {
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int b = 0; b < B; ++b) acc[r][b] = 0.f;
#pragma unroll
      for (int b = 0; b < B; ++b) acc2[b] = 0.f;
    }
    const int c0 = lane + ctrip * 64 * U;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + 64 * u;
      if (c >= nchunk) continue;
      float wf[R][8], w2f[8];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (WF32) {
          const int n = (cgrp * R + r < nrow) ? cgrp * R + r : nrow - 1;
          const f32x4_t a = *(const f32x4_t*)((const float*)p.W + (long)n * p.K + c * 8);
          const f32x4_t b4 = *(const f32x4_t*)((const float*)p.W + (long)n * p.K + c * 8 + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { wf[r][e] = a[e]; wf[r][4 + e] = b4[e]; }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) { wf[r][2 * e] = bf_lo(wc[r][u][e]); wf[r][2 * e + 1] = bf_hi(wc[r][u][e]); }
        }
      }
      if (gated) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { w2f[2 * e] = bf_lo(w2c[u][e]); w2f[2 * e + 1] = bf_hi(w2c[u][e]); }
      }
#pragma unroll
      for (int b = 0; b < B; ++b) {
        float xv[8];
        if (WF32) {
          const float* d = (const float*)gsm + (long)b * p.K + c * 8;
          const f32x4_t a = *(const f32x4_t*)d, b4 = *(const f32x4_t*)(d + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { xv[e] = a[e]; xv[4 + e] = b4[e]; }
        } else {
          const u32x4_t a = *(const u32x4_t*)((const bf16_t*)gsm + (long)b * p.K + c * 8);
#pragma unroll
          for (int e = 0; e < 4; ++e) { xv[2 * e] = bf_lo(a[e]); xv[2 * e + 1] = bf_hi(a[e]); }
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[r][b] = fmaf(xv[e], wf[r][e], acc[r][b]);
        if (gated) {
#pragma unroll
          for (int e = 0; e < 8; ++e) acc2[b] = fmaf(xv[e], w2f[e], acc2[b]);
        }
      }
    }
    if (ctrip == ntrip - 1) {
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int b = 0; b < B; ++b) acc[r][b] = wave_sum(acc[r][b]);
      if (gated) {
#pragma unroll
        for (int b = 0; b < B; ++b) acc2[b] = wave_sum(acc2[b]);
      }
      // every lane holds all R x B totals after the butterflies: lane r finishes row r
      const int n = cgrp * R + lane;
      if (lane < R && n < nrow) {
#pragma unroll
        for (int b = 0; b < B; ++b) {
          float v = 0.f, v2 = 0.f;
#pragma unroll
          for (int r = 0; r < R; ++r) v = (lane == r) ? acc[r][b] : v;
          if (gated) v2 = acc2[b];
          if (p.bias) v += e_bias;
          if (!WF32) v = rbf(v);                         // bf16 Linear output
          if (p.act) v = rbf(apply_act(p.act, v));
          if (gated) v = rbf(v * rbf(v2));
          if (p.resid) v = p.out_f32 ? e_res[b] + v : rbf(e_res[b] + v);
          if (p.out_f32) ((float*)p.out)[b * p.ldo + n] = v;
          else ((bf16_t*)p.out)[b * p.ldo + n] = f2bf(v);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Register-resident GEMV for bf16 x / bf16 W (the LLM decode shapes).  One workgroup = R consecutive output rows; its
// 256 threads split K (thread t owns the 16-byte chunks t, t+256, ...: J of them), so every load instruction of the
// workgroup covers 4 KB of one weight row.  There is no loop: each thread issues its x chunks, then ALL of its R*J
// (x2 when gated) weight loads, and only then starts on the RMSNorm -- the whole matrix is in flight across the
// resident workgroups from the first microsecond, which is what a 5-25 us kernel needs to get near the HBM rate
// (tools/probe/hbm_probe.hip: 5.0 / 5.9 / 6.4 TB/s for 25 / 45 / 90 MB reads at this launch granularity).
// Dot products use v_dot2_f32_bf16 on the packed operands (bf16 products are exact in fp32); the R*B (x2) partial
// sums are reduced across the wave with a transposing butterfly (V + 6 - log2 V shuffles for V values instead of
// 6 V) and across the 4 waves through LDS.
// ---------------------------------------------------------------------------------------------
// CHAIN: the dependent-launch form (ld_llm_dev.h: ChainSync) -- the first batch of weight rows is requested BEFORE the wait for
// the previous operation, x / residual / outputs go through sc1 accesses, the workgroup arrives on its counter at the end;
// at most 128 registers so that two such launches are always resident side by side (grids capped at 512 workgroups).
template <int B, int R, int J, bool GATED, bool NORM, bool CHAIN = false>
__global__ __launch_bounds__(256, CHAIN ? 4 : 1) void ld_gemv_reg_kernel(GemvParams p, int nbatch) {
  constexpr int NV = R * B * (GATED ? 2 : 1), V = ceil_pow2(NV), LOGV = ilog2(V);
  constexpr int XAUX = CHAIN ? 16 : 0;                 // sc1 on the activation loads
  __shared__ float red[2][4][V];
  __shared__ float ssq[4][B];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nchunk = p.K >> 3;
  const int er = tid / B, eb = tid - er * B;          // this thread's epilogue output within a batch: (row er, batch row eb)

  u32x4_t wn[R][J], w2n[GATED ? R : 1][J];
  // epilogue operands are kept as loaded (raw bits) until the epilogue converts them: a conversion here would make wave 0 wait
  // for the batch it has just requested (loads return in order) before it computes the current one
  uint32_t e_bias_n = 0u, e_res_n = 0u;
  // request one batch (R consecutive weight rows + the epilogue operands of its outputs)
  auto request = [&](int batch) {
    const int n0 = batch * R;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const long row = (n0 + r < p.N) ? n0 + r : p.N - 1;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int c = j * 256 + tid;
        wn[r][j] = (u32x4_t){0u, 0u, 0u, 0u};
        if (c < nchunk) wn[r][j] = __builtin_nontemporal_load((const u32x4_t*)((const bf16_t*)p.W + row * p.K + c * 8));
        if (GATED) {
          w2n[r][j] = (u32x4_t){0u, 0u, 0u, 0u};
          if (c < nchunk) w2n[r][j] = __builtin_nontemporal_load((const u32x4_t*)((const bf16_t*)p.W2 + row * p.K + c * 8));
        }
      }
    }
    const int en = n0 + er;
    if (tid < R * B && en < p.N) {
      if (p.bias) e_bias_n = p.bias[en];
      if (p.resid) {
        if (CHAIN) e_res_n = __hip_atomic_load((const bf16_t*)p.resid + eb * p.ldr + en, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else e_res_n = p.out_f32 ? ((const uint32_t*)p.resid)[eb * p.ldr + en] : (uint32_t)((const bf16_t*)p.resid)[eb * p.ldr + en];
      }
    }
  };

  // ---- x (L2) and the RMSNorm gains first, then the first batch of weight rows (HBM): all in flight together ----
  u32x4_t xq[B][J];
  f32x4_t g0[NORM ? J : 1], g1[NORM ? J : 1];
  // bounds-checked buffer loads (chunks past K read as zero): no branch and no zero-initialised destination -- with predicated
  // global loads the register allocator placed a v_mov behind every x load and the wave waited for each load in turn BEFORE the
  // first weight row had been requested (one L2 round trip per launch, six in the K = 11008 form)
  int batch = blockIdx.x;
  if (CHAIN) {                                         // weights do not depend on the previous operation: request, then wait for it
    request(batch);
    chain_wait(p.cs, tid);
  }
#pragma unroll
  for (int b = 0; b < B; ++b) {
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)((const bf16_t*)p.x + b * p.ldx), 0, p.K * 2, 0x00020000);
#pragma unroll
    for (int j = 0; j < J; ++j) xq[b][j] = __builtin_amdgcn_raw_buffer_load_b128(rx, (j * 256 + tid) * 16, 0, XAUX);
  }
  if (NORM) {
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)p.norm_w, 0, p.K * 4, 0x00020000);
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const u32x4_t a = __builtin_amdgcn_raw_buffer_load_b128(rg, (j * 256 + tid) * 32, 0, 0);
      const u32x4_t b = __builtin_amdgcn_raw_buffer_load_b128(rg, (j * 256 + tid) * 32 + 16, 0, 0);
      g0[j] = __builtin_bit_cast(f32x4_t, a); g1[j] = __builtin_bit_cast(f32x4_t, b);
    }
  }
  if (!CHAIN) request(batch);

  // ---- x: optional fused RMSNorm (transformer_blocks.py:22-40), back to packed bf16 ----
  if (NORM) {
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const float ss = wave_sum(chunks_sumsq<J>(xq[b]));
      if (lane == 0) ssq[wave][b] = ss;
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const float r1 = rsqrtf((ssq[0][b] + ssq[1][b] + ssq[2][b] + ssq[3][b]) / (float)p.K + p.norm_eps);
#pragma unroll
      for (int j = 0; j < J; ++j) {
        if (j * 256 + tid < nchunk) {
          const u32x4_t a = xq[b][j];
          xq[b][j] = (u32x4_t){pack_bf16x2(bf_lo(a[0]) * r1 * g0[j][0], bf_hi(a[0]) * r1 * g0[j][1]),
                               pack_bf16x2(bf_lo(a[1]) * r1 * g0[j][2], bf_hi(a[1]) * r1 * g0[j][3]),
                               pack_bf16x2(bf_lo(a[2]) * r1 * g1[j][0], bf_hi(a[2]) * r1 * g1[j][1]),
                               pack_bf16x2(bf_lo(a[3]) * r1 * g1[j][2], bf_hi(a[3]) * r1 * g1[j][3])};
        }
      }
    }
  }

  // ---- batches batch, batch + gridDim.x, ...: the next batch's rows are requested before this one is consumed ----
  int par = 0;
#pragma unroll 1
  for (; batch < nbatch; batch += gridDim.x, par ^= 1) {
    u32x4_t w[R][J], w2[GATED ? R : 1][J];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int j = 0; j < J; ++j) { w[r][j] = wn[r][j]; if (GATED) w2[r][j] = w2n[r][j]; }
    const uint32_t e_bias_raw = e_bias_n, e_res_raw = e_res_n;
    if (batch + (int)gridDim.x < nbatch) request(batch + gridDim.x);

    float v[V];
#pragma unroll
    for (int i = 0; i < V; ++i) v[i] = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int j = 0; j < J; ++j)
#pragma unroll
        for (int b = 0; b < B; ++b)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[r * B + b] = dot2_bf16(w[r][j][e], xq[b][j][e], v[r * B + b]);
            if (GATED) v[R * B + r * B + b] = dot2_bf16(w2[r][j][e], xq[b][j][e], v[R * B + r * B + b]);
          }
    wave_sum_multi<V>(v, lane);
    if ((lane & ((64 >> LOGV) - 1)) == 0) red[par][wave][lane >> (6 - LOGV)] = v[0];
    __syncthreads();
    const int en = batch * R + er;
    if (tid < R * B && en < p.N) {
      const float(&rd)[4][V] = red[par];
      float a = rd[0][tid] + rd[1][tid] + rd[2][tid] + rd[3][tid];
      const float e_bias = bf2f((bf16_t)e_bias_raw), e_res = p.out_f32 ? __uint_as_float(e_res_raw) : bf2f((bf16_t)e_res_raw);
      if (p.bias) a += e_bias;
      a = rbf(a);                                        // bf16 Linear output
      if (p.act) a = rbf(apply_act(p.act, a));
      if (GATED) a = rbf(a * rbf(rd[0][R * B + tid] + rd[1][R * B + tid] + rd[2][R * B + tid] + rd[3][R * B + tid]));
      if (p.resid) a = p.out_f32 ? e_res + a : rbf(e_res + a);
      if (CHAIN) __hip_atomic_store((bf16_t*)p.out + eb * p.ldo + en, f2bf(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (p.out_f32) ((float*)p.out)[eb * p.ldo + en] = a;
      else ((bf16_t*)p.out)[eb * p.ldo + en] = f2bf(a);
    }
  }
  if (CHAIN) chain_arrive(p.cs, tid, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// RMSNorm / LayerNorm over short rows (fp32 math), one wave per row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ld_rmsnorm_kernel(const bf16_t* x, const float* w, bf16_t* out,
                                                         int rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const bf16_t* xr = x + (long)r * D;
  float ss = 0.f;
  for (int c = lane; c < (D >> 3); c += 64) {
    const u32x4_t a = *(const u32x4_t*)(xr + c * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float lo = bf_lo(a[e]), hi = bf_hi(a[e]); ss += lo * lo + hi * hi; }
  }
  ss = wave_sum(ss);
  const float rs = rsqrtf(ss / (float)D + eps);
  for (int c = lane; c < (D >> 3); c += 64) {
    const u32x4_t a = *(const u32x4_t*)(xr + c * 8);
    const f32x4_t w0 = *(const f32x4_t*)(w + c * 8), w1 = *(const f32x4_t*)(w + c * 8 + 4);
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = bf_lo(a[e]) * rs; v[2 * e + 1] = bf_hi(a[e]) * rs; }
    u32x4_t o;
    o[0] = pack_bf16x2(v[0] * w0[0], v[1] * w0[1]); o[1] = pack_bf16x2(v[2] * w0[2], v[3] * w0[3]);
    o[2] = pack_bf16x2(v[4] * w1[0], v[5] * w1[1]); o[3] = pack_bf16x2(v[6] * w1[2], v[7] * w1[3]);
    *(u32x4_t*)(out + (long)r * D + c * 8) = o;
  }
}

// final LayerNorm of GPT.sample: bf16 rows (row stride ldx) -> fp32, fp32 affine
__global__ __launch_bounds__(256) void ld_ln_f32out_kernel(const bf16_t* x, long ldx, const float* w, const float* b,
                                                           float* out, int rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const bf16_t* xr = x + (long)r * ldx;
  float s = 0.f, ss = 0.f;
  for (int i = lane; i < D; i += 64) { const float v = bf2f(xr[i]); s += v; }
  s = wave_sum(s);
  const float mean = s / (float)D;
  for (int i = lane; i < D; i += 64) { const float d = bf2f(xr[i]) - mean; ss += d * d; }
  ss = wave_sum(ss);
  const float rs = rsqrtf(ss / (float)D + eps);
  for (int i = lane; i < D; i += 64) out[(long)r * D + i] = (bf2f(xr[i]) - mean) * rs * w[i] + b[i];
}

// Same, for rows of at most 4096 elements (D % 8 == 0): the row is read once with 16-byte loads and stays in registers
// for the mean, the centred second moment and the affine output (the three-pass kernel above spent 22 us on the two
// rows of a decode step, all of it dependent 2-byte loads).
__global__ __launch_bounds__(256) void ld_ln_f32out_reg_kernel(const bf16_t* x, long ldx, const float* w, const float* b,
                                                               float* out, int rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const bf16_t* xr = x + (long)r * ldx;
  const int nchunk = D >> 3;
  float v[8][8];
  float s = 0.f, ss = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = lane + 64 * k;
    u32x4_t a = (u32x4_t){0u, 0u, 0u, 0u};
    if (c < nchunk) a = *(const u32x4_t*)(xr + c * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[k][2 * e] = bf_lo(a[e]); v[k][2 * e + 1] = bf_hi(a[e]); s += v[k][2 * e] + v[k][2 * e + 1]; }
  }
  s = wave_sum(s);
  const float mean = s / (float)D;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (lane + 64 * k < nchunk) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[k][e] - mean; ss += d * d; }
    }
  }
  ss = wave_sum(ss);
  const float rs = rsqrtf(ss / (float)D + eps);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = lane + 64 * k;
    if (c < nchunk) {
      const f32x4_t w0 = *(const f32x4_t*)(w + c * 8), w1 = *(const f32x4_t*)(w + c * 8 + 4);
      const f32x4_t b0 = *(const f32x4_t*)(b + c * 8), b1 = *(const f32x4_t*)(b + c * 8 + 4);
      f32x4_t o0, o1;
#pragma unroll
      for (int e = 0; e < 4; ++e) { o0[e] = (v[k][e] - mean) * rs * w0[e] + b0[e]; o1[e] = (v[k][4 + e] - mean) * rs * w1[e] + b1[e]; }
      *(f32x4_t*)(out + (long)r * D + c * 8) = o0;
      *(f32x4_t*)(out + (long)r * D + c * 8 + 4) = o1;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// RoPE on q,k + KV append.  qkv [B][m][3][H][128] bf16; cache K/V [B][Lmax][H][128];
// position of token j = *pos + j.  q_out [B][m][H][128].
// ---------------------------------------------------------------------------------------------
__global__ void ld_rope_append_kernel(const bf16_t* qkv, const float* cos_t, const float* sin_t, const int* pos_ptr,
                                      bf16_t* q_out, bf16_t* kc, bf16_t* vc, int B, int m, int H, int Lmax) {
  const int D = 128;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (b, j, h, pair)
  const int total = B * m * H * (D / 2);
  if (idx >= total) return;
  const int pr = idx % (D / 2);
  const int h = (idx / (D / 2)) % H;
  const int j = (idx / (D / 2) / H) % m;
  const int b = idx / (D / 2) / H / m;
  const int pos = *pos_ptr + j;
  const float c = cos_t[pos * (D / 2) + pr], s = sin_t[pos * (D / 2) + pr];
  const long base = (((long)(b * m + j) * 3) * H + h) * D + 2 * pr;
  const long hd = (long)H * D;
  const float qa = bf2f(qkv[base]), qb = bf2f(qkv[base + 1]);
  const float ka = bf2f(qkv[base + hd]), kb = bf2f(qkv[base + hd + 1]);
  const long qo = ((long)(b * m + j) * H + h) * D + 2 * pr;
  q_out[qo] = f2bf(qa * c - qb * s);
  q_out[qo + 1] = f2bf(qa * s + qb * c);
  const long co = (((long)b * Lmax + pos) * H + h) * D + 2 * pr;
  kc[co] = f2bf(ka * c - kb * s);
  kc[co + 1] = f2bf(ka * s + kb * c);
  vc[co] = qkv[base + 2 * hd];
  vc[co + 1] = qkv[base + 2 * hd + 1];
}

// ---------------------------------------------------------------------------------------------
// KV attention, head_dim 128: query j (position *pos + j) attends keys [0, *pos + j].
// Mirrors the reference's dtype flow: bf16 scores, bf16 (score / sqrt(d)), fp32 softmax -> bf16 p, bf16 out.
// grid (B*H, m), 256 threads; scores staged in LDS (fp32, up to Lmax entries).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ld_kv_attn_kernel(const bf16_t* q, const bf16_t* kc, const bf16_t* vc,
                                                         const int* pos_ptr, bf16_t* out, int B, int m, int H, int Lmax) {
  extern __shared__ float sc[];       // [Lmax] scores, then [2][128] partial outputs + 8 reduction slots
  const int D = 128;
  const int bh = blockIdx.x, j = blockIdx.y;
  const int b = bh / H, h = bh - b * H;
  const int L = *pos_ptr + j + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bf16_t* qv = q + ((long)(b * m + j) * H + h) * D;
  // 16 lanes x 8 dims per key, 4 keys per wave iteration
  const int sub = lane & 15, kq = lane >> 4;
  float qreg[8];
  {
    const u32x4_t a = *(const u32x4_t*)(qv + sub * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) { qreg[2 * e] = bf_lo(a[e]); qreg[2 * e + 1] = bf_hi(a[e]); }
  }
  const float inv_sqrt_d = 0.08838834764831845f;   // 1/sqrt(128)
  float lmax = -3.0e38f;
  for (int k0 = wave * 4; k0 < L; k0 += 16) {
    const int key = k0 + kq;
    float d = 0.f;
    if (key < L) {
      const u32x4_t a = *(const u32x4_t*)(kc + (((long)b * Lmax + key) * H + h) * D + sub * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) { d = fmaf(qreg[2 * e], bf_lo(a[e]), d); d = fmaf(qreg[2 * e + 1], bf_hi(a[e]), d); }
    }
    d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64); d += __shfl_xor(d, 8, 64);
    if (key < L) {
      const float s = rbf(rbf(d) * inv_sqrt_d);    // einsum -> bf16, then "/ sqrt(d)" -> bf16
      if (sub == 0) sc[key] = s;
      lmax = fmaxf(lmax, s);
    }
  }
  float* red = sc + Lmax;
  lmax = wave_max(lmax);
  if (lane == 0) red[wave] = lmax;
  __syncthreads();
  const float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float lsum = 0.f;
  for (int k = tid; k < L; k += 256) { const float e = __expf(sc[k] - mx); sc[k] = e; lsum += e; }
  lsum = wave_sum(lsum);
  __syncthreads();
  if (lane == 0) red[4 + wave] = lsum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  // out[d] = sum_k bf16(p_k) * v[k][d]; thread -> (d = tid & 127, key parity = tid >> 7)
  const int d = tid & 127, par = tid >> 7;
  float acc = 0.f;
  for (int k = par; k < L; k += 2) {
    const float pk = rbf(sc[k] * inv);
    acc = fmaf(pk, bf2f(vc[(((long)b * Lmax + k) * H + h) * D + d]), acc);
  }
  float* part = red + 8;
  part[par * 128 + d] = acc;
  __syncthreads();
  if (tid < 128) out[((long)(b * m + j) * H + h) * D + tid] = f2bf(part[tid] + part[128 + tid]);
}

// ---------------------------------------------------------------------------------------------
// Decode-step (m = 1) KV attention split over the key axis ("flash decoding"): grid (B*H, nsplit) so that the
// 2 x 16 (batch, head) pairs still fill 256 CUs.  Each workgroup writes (max, sum, unnormalised out[128]) for its
// key range; ld_kv_attn_combine_kernel merges them.  16 lanes x 8 dims per key, 16 keys per workgroup iteration,
// every K/V row is read once with 16-byte loads.  (p stays fp32 here; the reference rounds the normalised p to
// bf16 before the PV product -- a <= 2^-9 relative, zero-mean difference per term.)
// ---------------------------------------------------------------------------------------------
// If qkv != nullptr the kernel also does apply_rope + the KV append of the current token itself (q is then ignored):
// every workgroup rotates q on the fly; the workgroup whose key range holds position *pos rotates/stores the new k
// and v into the cache before using them.
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The merge of the splits rides in the same launch: every workgroup writes its partial result device-coherently (sc1), drains
// the stores and arrives on the (batch row, head)'s counter; the LAST one to arrive reads all partial results (sc1 loads),
// writes the attention output and leaves the counter at zero for the next launch.  Nobody waits for anybody (no co-residency
// requirement); one launch and one ~5 us dependent kernel per block less than split + combine.
// CHAIN: dependent-launch form (ChainSync): the K / V rows (earlier steps' data) are requested before the wait for the qkv
// operation, the new token's q / k / v come through sc1 loads, the merged output is written sc1, every workgroup arrives at the
// end; pos_value >= 0 replaces the device-side position (no dependence on the previous step's sampling launch).
template <bool CHAIN>
__global__ __launch_bounds__(256, CHAIN ? 2 : 1) void ld_kv_attn_split_kernel(const bf16_t* q, const bf16_t* qkv, const float* cos_t,
                                                               const float* sin_t, bf16_t* kc, bf16_t* vc,
                                                               const int* pos_ptr, int pos_value, float* ws, unsigned* counters,
                                                               bf16_t* out, int B, int H, int Lmax, int nsplit, KvSplitRule rule,
                                                               ChainSync cs) {
  __shared__ float red[8 + 4 * 128];
  __shared__ int is_last;
  const int D = 128;
  const int bh = blockIdx.x, sp = blockIdx.y;
  const int b = bh / H, h = bh - b * H;
  const int L = (pos_value >= 0 ? pos_value : *pos_ptr) + 1;
  const int ns = kv_eff_splits(L, nsplit, rule);       // splits in use at this context length; partial results keep the [bh][nsplit] layout
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (sp >= ns) {                                      // a split this step does not use
    if (CHAIN) { chain_wait(cs, tid); chain_arrive(cs, tid, blockIdx.y * gridDim.x + blockIdx.x); }
    return;
  }
  const int chunk = (L + ns - 1) / ns;
  const int k_begin = sp * chunk, k_end = min(L, k_begin + chunk);
  const int n = max(0, k_end - k_begin);
  const int sub = lane & 15, kq = lane >> 4;
  float* out_ws = ws + ((long)bh * nsplit + sp) * (D + 2);
  bf16_t* direct = (ns == 1 && !CHAIN) ? out + (long)bh * D : nullptr;
  KvRows rows;
  if (CHAIN) {
    if (n > 0) kv_rows_request(rows, kc, vc, (long)b * Lmax + k_begin, H, h, n, wave, kq, sub);
    chain_wait(cs, tid);
  }
  if (n == 0) {
    if (tid < D) st_agent(out_ws + 2 + tid, 0.f);
    if (tid == 0) { st_agent(out_ws, -3.0e38f); st_agent(out_ws + 1, 0.f); }
  } else {
    // Request order = dependency order: the new token's q/k/v + rotation factors first (short, L2), then this wave's K
    // rows AND V rows (4 keys per trip, <= KV_MAXIT trips) all at once -- one exposed HBM latency per launch, and the RoPE
    // arithmetic runs underneath it.  The key being appended this step is taken from qkv directly (its cache slot is
    // written for later steps but not read back here), so the prologue needs no barrier.
    u32x4_t a_q = (u32x4_t){0u, 0u, 0u, 0u}, a_k = a_q, a_v = a_q;
    float cs_[4] = {0.f, 0.f, 0.f, 0.f}, sn[4] = {0.f, 0.f, 0.f, 0.f};
    const int pos = L - 1;
    if (qkv) {
      const bf16_t* src = qkv + ((long)b * 3 * H + h) * D;      // [B][3][H][128]: q at +0, k at +H*D, v at +2*H*D
      if (CHAIN) {
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (2 * H + 1) * D * 2, 0x00020000);
        a_q = __builtin_amdgcn_raw_buffer_load_b128(rq, sub * 16, 0, 16);
        a_k = __builtin_amdgcn_raw_buffer_load_b128(rq, (H * D + sub * 8) * 2, 0, 16);
        a_v = __builtin_amdgcn_raw_buffer_load_b128(rq, (2 * H * D + sub * 8) * 2, 0, 16);
      } else {
        a_q = *(const u32x4_t*)(src + sub * 8);
        a_k = *(const u32x4_t*)(src + (long)H * D + sub * 8);
        a_v = *(const u32x4_t*)(src + 2L * H * D + sub * 8);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { cs_[e] = cos_t[pos * 64 + sub * 4 + e]; sn[e] = sin_t[pos * 64 + sub * 4 + e]; }
    } else {
      a_q = *(const u32x4_t*)(q + ((long)b * H + h) * D + sub * 8);
    }
    if (!CHAIN) kv_rows_request(rows, kc, vc, (long)b * Lmax + k_begin, H, h, n, wave, kq, sub);
    kv_attn_split_core(rows, a_q, a_k, a_v, cs_, sn, qkv != nullptr, kc, vc, (long)b * Lmax + pos, H, h, pos - k_begin, n, true,
                       out_ws, red, red + 8, tid, lane, wave, [](float* p, float v) { st_agent(p, v); }, direct);
  }
  if (direct) return;                                  // one split: the output row is written, nothing to merge, no counter traffic
  // Hand-off in the write-through form (cdna_hip_programming.md section 6 Guideline 16, recipe R1; the second valid form of
  // MI355X_MICROARCH.md "Valid forms besides R1/R2": {sc1 stores and loads both sides}): the partial results were stored sc1
  // (st_agent: they leave this XCD's L2), EVERY storing wave drains vmcnt, the workgroup meets, ONE lane arrives with a relaxed
  // agent-scope add; the last arriver reads the partials with sc1 loads (ld_agent: L1 bypassed), which stand in for the acquire
  // because the producers stored sc1.  This relies on the ISA-level behaviour of sc1 accesses on gfx950 that the guide documents,
  // not on the HIP memory model (a relaxed counter orders nothing there); tests/test_gpu_llm_longctx.py and the bit-identity tests
  // of the decode forms are its check.  A release/acquire pair on the counter instead would put a buffer_wbl2 + buffer_inv
  // (~3.5 us, MI355X_MICROARCH.md) into each of the 24 x 1244 launches of a decode.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this thread's part of the partial result is at the coherence point
  __syncthreads();
  if (tid == 0)
    is_last = __hip_atomic_fetch_add(counters + bh, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(ns - 1);
  __syncthreads();
  if (is_last) {
    if (tid < D) {
      const float r = kv_attn_combine_core(ws + (long)bh * nsplit * (D + 2), ns, tid, [](const float* p) { return ld_agent(p); });
      if (CHAIN) __hip_atomic_store(out + (long)bh * D + tid, f2bf(r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else out[(long)bh * D + tid] = f2bf(r);
    }
    if (tid == 0) __hip_atomic_store(counters + bh, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (CHAIN) chain_arrive(cs, tid, blockIdx.y * gridDim.x + blockIdx.x);
}

// token embedding rows (fp32 table) -> bf16 features, same token for every batch row
__global__ void ld_embed_kernel(const float* table, const long* token, bf16_t* out, int B, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D) return;
  const long t = *token;
  out[i] = f2bf(table[t * D + (i % D)]);
}

// logits [2][V] (cond, uncond) or [1][V] -> probs [V]: CFG, /temperature, optional restriction, optional top-k,
// softmax, optional top-p.  Single block; V <= LD_SAMPLE_MAXV.  (lm_model.py:417-454, utils.py:345-359)
#define LD_SAMPLE_MAXV 4096
__device__ __forceinline__ float block_sum_1024(float v, float* red, int tid, int nthreads) {
  v = wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  float tot = 0.f;
  for (int w = 0; w < (nthreads >> 6); ++w) tot += red[w];
  return tot;
}
// SAMPLE: the rest of the step in the same launch -- the draw that torch.multinomial makes for one sample,
// argmax_i(probs[i] / q[i]) with q ~ Exp(1) from the caller's generator (`noise`; ATen's multinomial is exactly
// empty_like(p).exponential_(1, gen), div, argmax: ties -> lowest index), the forced-token schedule / token record /
// position advance of ld_decode_advance_kernel, and the embedding rows of the token the next step starts from.
struct SampleArgs {
  const float* noise;      // [V] Exp(1) draws
  const int* forced; int* pos; long* token; long* out_tokens; int* out_count; long* sampled;
  const float* emb; bf16_t* x; int B, D;
};
template <bool SAMPLE>
__global__ __launch_bounds__(1024) void ld_logits_to_probs_kernel(const float* logits, float* probs, float* cfg_logits,
                                                                  int V, int guided, float scale, float temperature,
                                                                  const int* pos_ptr, const int* allowed, int n_allowed_stride,
                                                                  int top_k, float top_p, SampleArgs sa) {
  __shared__ float red[32];
  __shared__ int redi[32];
  __shared__ float sv[LD_SAMPLE_MAXV];     // value per vocabulary id
  __shared__ float ss[LD_SAMPLE_MAXV];     // values in descending order (top-p)
  __shared__ float thr_s;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int* al = nullptr;
  int nal = 0;
  if (allowed) {
    al = allowed + (long)(*pos_ptr + 1) * n_allowed_stride;   // table indexed by the position being generated
    nal = al[0];
  }
  for (int i = tid; i < V; i += nt) {
    float l = logits[i];
    if (guided) { const float u = logits[V + i]; l = u + scale * (l - u); }
    if (cfg_logits) cfg_logits[i] = l;
    l = l / temperature;
    if (nal > 0) {
      bool ok = false;
      for (int a = 0; a < nal; ++a) ok |= (al[1 + a] == i);
      if (!ok) l = -INFINITY;
    }
    sv[i] = l;
  }
  __syncthreads();
  // top-k (unrestricted positions only): everything below the k-th largest logit -> -inf; ties at the threshold stay
  if (top_k > 0 && top_k < V && nal == 0) {
    for (int i = tid; i < V; i += nt) {
      const float v = sv[i];
      int gt = 0, ge = 0;
      for (int j = 0; j < V; ++j) { const float o = sv[j]; gt += (o > v); ge += (o >= v); }
      if (gt < top_k && top_k <= ge) thr_s = v;               // every writer holds the same value
    }
    __syncthreads();
    const float thr = thr_s;
    for (int i = tid; i < V; i += nt) if (sv[i] < thr) sv[i] = -INFINITY;
    __syncthreads();
  }
  float mx = -3.0e38f;
  for (int i = tid; i < V; i += nt) mx = fmaxf(mx, sv[i]);
  mx = wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  float m2 = red[0];
  for (int w = 1; w < (nt >> 6); ++w) m2 = fmaxf(m2, red[w]);
  float s = 0.f;
  for (int i = tid; i < V; i += nt) { const float e = expf(sv[i] - m2); sv[i] = e; s += e; }
  const float tot = block_sum_1024(s, red, tid, nt);
  for (int i = tid; i < V; i += nt) sv[i] = sv[i] / tot;
  __syncthreads();
  // top-p (unrestricted positions only): drop sorted position j >= 1 when cumsum[j-1] >= top_p, renormalise
  if (top_p >= 0.f && nal == 0) {
    int rk[(LD_SAMPLE_MAXV + 1023) / 1024];
    int c = 0;
    for (int i = tid; i < V; i += nt, ++c) {
      const float v = sv[i];
      int r = 0;
      for (int j = 0; j < V; ++j) { const float o = sv[j]; r += (o > v) || (o == v && j < i); }   // stable descending rank
      rk[c] = r;
      ss[r] = v;
    }
    __syncthreads();
    if (tid == 0) {                                            // sequential fp32 cumsum (torch.cumsum's CPU order)
      float acc = 0.f;
      for (int j = 0; j < V; ++j) { acc += ss[j]; ss[j] = acc; }
    }
    __syncthreads();
    float ks = 0.f;
    c = 0;
    for (int i = tid; i < V; i += nt, ++c) {
      const int r = rk[c];
      if (r >= 1 && ss[r - 1] >= top_p) sv[i] = 0.f;
      ks += sv[i];
    }
    const float kept = block_sum_1024(ks, red, tid, nt);
    for (int i = tid; i < V; i += nt) sv[i] = sv[i] / kept;
    __syncthreads();
  }
  for (int i = tid; i < V; i += nt) if (probs) probs[i] = sv[i];
  if constexpr (SAMPLE) {
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int i = tid; i < V; i += nt) {
      const float r = sv[i] / sa.noise[i];                   // IEEE division, as at::div
      if (r > best || (r == best && i < bi) || bi == 0x7fffffff) { best = r; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    __syncthreads();
    if ((tid & 63) == 0) { red[tid >> 6] = best; redi[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < (nt >> 6); ++w)
        if (red[w] > best || (red[w] == best && redi[w] < bi)) { best = red[w]; bi = redi[w]; }
      // ld_decode_advance_kernel: the token generated now sits at position pos + 1
      const int pos = *sa.pos;
      const int f = sa.forced[pos + 1];
      long t = bi;
      if (sa.sampled) *sa.sampled = t;
      if (f >= 0) t = f;
      else { sa.out_tokens[*sa.out_count] = t; *sa.out_count = *sa.out_count + 1; }
      *sa.token = t;
      *sa.pos = pos + 1;
      redi[0] = (int)t;
    }
    __syncthreads();
    const long t = redi[0];
    for (int i = tid; i < sa.B * sa.D; i += nt) sa.x[i] = f2bf(sa.emb[t * sa.D + (i % sa.D)]);      // ld_embed_kernel
  }
}

// after torch.multinomial: apply the forced-token schedule, record the sampled token, advance position
__global__ void ld_decode_advance_kernel(const long* sampled, const int* forced, int* pos_ptr, long* token,
                                         long* out_tokens, int* out_count) {
  const int pos = *pos_ptr;
  const int f = forced[pos + 1];   // token generated now sits at position pos + 1
  long t = *sampled;
  if (f >= 0) t = f;
  else { out_tokens[*out_count] = t; *out_count = *out_count + 1; }
  *token = t;
  *pos_ptr = pos + 1;
}

template <int B, bool WF32, int R, int U = 4>
int launch_gemv_cfg(const GemvParams& p, hipStream_t st) {
  const size_t smem = (size_t)B * p.K * (WF32 ? 4 : 2) + 64;
  static thread_local LdSmemCache cache{};      // per instantiation
  if (int rc = ld_ensure_dyn_smem((const void*)ld_gemv_kernel<B, WF32, R, U>, smem, &cache)) return rc;
  // persistent grid: ~2 workgroups per CU (or fewer when there are not enough rows); every wave loops over row groups
  long blocks = (p.N + 4 * R - 1) / (4 * R);
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL((ld_gemv_kernel<B, WF32, R, U>), dim3((unsigned)blocks), dim3(256), smem, st, p);
  return ld_check_launch("ld_gemv");
}

template <int B, int R, int J, bool GATED, bool NORM>
int launch_gemv_reg(const GemvParams& p, hipStream_t st) {
  // nbatch batches of R rows over at most `cap` workgroups, every workgroup taking the same number of batches
  static const int cap = getenv("LD_GEMV_WGS") ? atoi(getenv("LD_GEMV_WGS")) : 768;
  const int nbatch = (p.N + R - 1) / R;
  const int trips = (nbatch + cap - 1) / cap;
  const int grid = (nbatch + trips - 1) / trips;
  hipLaunchKernelGGL((ld_gemv_reg_kernel<B, R, J, GATED, NORM>), dim3((unsigned)grid), dim3(256), 0, st, p, nbatch);
  return ld_check_launch("ld_gemv");
}

// R rows per batch: R*J (x2 gated) 16-byte loads per thread in flight per batch, two batches deep.
template <int B, int J, bool GATED, bool NORM>
int launch_gemv_reg_r(const GemvParams& p, int R, hipStream_t st) {
  constexpr int L1 = J * (GATED ? 2 : 1);            // loads per thread per row
  if constexpr (L1 <= 2 && B <= 2) { if (R >= 8) return launch_gemv_reg<B, 8, J, GATED, NORM>(p, st); }
  if constexpr (L1 <= 4) { if (R >= 4) return launch_gemv_reg<B, 4, J, GATED, NORM>(p, st); }
  if (R >= 2) return launch_gemv_reg<B, 2, J, GATED, NORM>(p, st);
  return launch_gemv_reg<B, 1, J, GATED, NORM>(p, st);
}

template <int B, int J>
int launch_gemv_reg_j(const GemvParams& p, int R, hipStream_t st) {
  const bool gated = p.W2 != nullptr, norm = p.norm_w != nullptr;
  if constexpr (J <= 2) {
    if (gated) return norm ? launch_gemv_reg_r<B, J, true, true>(p, R, st) : launch_gemv_reg_r<B, J, true, false>(p, R, st);
  }
  return norm ? launch_gemv_reg_r<B, J, false, true>(p, R, st) : launch_gemv_reg_r<B, J, false, false>(p, R, st);
}

template <int B>
int launch_gemv_b(const GemvParams& p, hipStream_t st) {
  static const int mode = getenv("LD_GEMV_MODE") ? atoi(getenv("LD_GEMV_MODE")) : 0;     // 1 = streaming-loop kernel only
  static const int forced_r = getenv("LD_GEMV_R") ? atoi(getenv("LD_GEMV_R")) : 0;
  const int nchunk = p.K >> 3;
  const bool gated = p.W2 != nullptr;
  if (mode != 1 && !p.w_f32 && !p.x_f32 && !p.in_act) {
    if (nchunk <= 256) return launch_gemv_reg_j<B, 1>(p, forced_r ? forced_r : 4, st);     // measured: tools/gemv_shapes.py
    if (nchunk <= 512) return launch_gemv_reg_j<B, 2>(p, forced_r ? forced_r : (gated ? 2 : 4), st);
    if constexpr (B <= 2) {
      if (nchunk <= 1536 && !gated) return launch_gemv_reg_j<B, 6>(p, forced_r ? forced_r : 2, st);
    }
  }
  if (p.w_f32) return launch_gemv_cfg<B, true, 1>(p, st);
  // short rows (one 16-byte load per lane covers a row: the DiT's adaLN modulations, K = 512): eight rows per wave and trip
  // keep eight loads per lane in flight (1.3 -> 4+ TB/s on the 708 MB all-layers matrix)
  if (!p.W2 && nchunk <= 64 && p.N >= 4096) return launch_gemv_cfg<B, false, 8, 1>(p, st);
  if (p.W2 || p.N < 4096) return launch_gemv_cfg<B, false, 1>(p, st);   // gated MLP, or few rows: keep all 256 CUs busy
  return launch_gemv_cfg<B, false, 2>(p, st);     // (one row per wave measured slower at N = 6144: 12.4 vs 11.5 us)
}

}  // namespace

KvSplitRule ld_kv_split_rule() {
  static const KvSplitRule rule = [] {
    KvSplitRule r{KV_SPLIT_T1, KV_SPLIT_T2, KV_SPLIT_T4};
    if (const char* e = getenv("LD_KV_SPLIT_T")) {
      int a = 0, b = 0, c = 0;
      if (sscanf(e, "%d,%d,%d", &a, &b, &c) == 3) r = KvSplitRule{a, b, c};
    }
    return r;
  }();
  return rule;
}

LD_API int ld_gemv(const void* x, int64_t ldx, int32_t x_f32, const void* W, const void* W2, int32_t w_f32,
                   const void* bias, const void* resid, int64_t ldr, void* out, int64_t ldo, int32_t out_f32,
                   int64_t B, int64_t N, int64_t K, int32_t in_act, int32_t act, const float* norm_w, float norm_eps,
                   void* stream) {
  LD_REQUIRE(x && W && out, "ld_gemv: null pointer");
  LD_REQUIRE(B >= 1 && B <= MAXB, "ld_gemv: batch %ld not in [1,%d]", (long)B, MAXB);
  LD_REQUIRE(K % 8 == 0 && ldx % 8 == 0, "ld_gemv: K and ldx must be multiples of 8");
  LD_REQUIRE((size_t)B * K * (w_f32 ? 4 : 2) + 64 <= 160 * 1024, "ld_gemv: B*K too large for the LDS-staged activations");
  LD_REQUIRE(!(w_f32 && W2), "ld_gemv: gated form needs bf16 weights");
  LD_REQUIRE(!w_f32 || (x_f32 && out_f32), "ld_gemv: fp32 weights need fp32 in/out");
  GemvParams p{};
  p.x = x; p.W = W; p.W2 = W2; p.bias = (const bf16_t*)bias; p.resid = resid; p.out = out;
  p.B = (int)B; p.N = (int)N; p.K = (int)K; p.ldx = ldx; p.ldo = ldo; p.ldr = ldr;
  p.x_f32 = x_f32; p.w_f32 = w_f32; p.out_f32 = out_f32; p.in_act = in_act; p.act = act;
  p.norm_w = norm_w; p.norm_eps = norm_eps;
  LD_REQUIRE(!norm_w || (!x_f32 && !w_f32), "ld_gemv: fused RMSNorm needs bf16 x and weights");
  hipStream_t st = (hipStream_t)stream;
  switch (B) {
    case 1: return launch_gemv_b<1>(p, st);
    case 2: return launch_gemv_b<2>(p, st);
    case 3: return launch_gemv_b<3>(p, st);
    default: return launch_gemv_b<4>(p, st);
  }
}

LD_API int ld_rmsnorm_bf16(const void* x, const float* w, void* out, int64_t rows, int64_t D, float eps, void* stream) {
  LD_REQUIRE(x && w && out && D % 8 == 0, "ld_rmsnorm_bf16: bad args");
  hipLaunchKernelGGL(ld_rmsnorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, w, (bf16_t*)out, (int)rows, (int)D, eps);
  return ld_check_launch("ld_rmsnorm_bf16");
}

LD_API int ld_layernorm_bf16_to_f32(const void* x, int64_t ldx, const float* w, const float* b, float* out,
                                    int64_t rows, int64_t D, float eps, void* stream) {
  LD_REQUIRE(x && w && b && out, "ld_layernorm_bf16_to_f32: null pointer");
  if (D % 8 == 0 && D <= 4096 && ldx % 8 == 0)
    hipLaunchKernelGGL(ld_ln_f32out_reg_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, (long)ldx, w, b, out, (int)rows, (int)D, eps);
  else
    hipLaunchKernelGGL(ld_ln_f32out_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, (long)ldx, w, b, out, (int)rows, (int)D, eps);
  return ld_check_launch("ld_layernorm_bf16_to_f32");
}

LD_API int ld_llm_rope_append(const void* qkv, const float* cos_t, const float* sin_t, const int32_t* pos,
                              void* q_out, void* k_cache, void* v_cache, int64_t B, int64_t m, int64_t H,
                              int64_t Lmax, void* stream) {
  LD_REQUIRE(qkv && cos_t && sin_t && pos && q_out && k_cache && v_cache, "ld_llm_rope_append: null pointer");
  const long total = B * m * H * 64;
  hipLaunchKernelGGL(ld_rope_append_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)qkv, cos_t, sin_t, (const int*)pos, (bf16_t*)q_out, (bf16_t*)k_cache,
                     (bf16_t*)v_cache, (int)B, (int)m, (int)H, (int)Lmax);
  return ld_check_launch("ld_llm_rope_append");
}

static int kv_attn_impl(const void* q, const void* k_cache, const void* v_cache, const int32_t* pos, int32_t pos_value, void* out,
                        int64_t B, int64_t m, int64_t H, int64_t Lmax, float* workspace, int64_t nsplit,
                        const void* qkv_fused, const float* cos_t, const float* sin_t, void* stream);

LD_API int ld_llm_kv_attn(const void* q, const void* k_cache, const void* v_cache, const int32_t* pos, void* out,
                          int64_t B, int64_t m, int64_t H, int64_t Lmax, float* workspace, int64_t nsplit,
                          const void* qkv_fused, const float* cos_t, const float* sin_t, void* stream) {
  return kv_attn_impl(q, k_cache, v_cache, pos, -1, out, B, m, H, Lmax, workspace, nsplit, qkv_fused, cos_t, sin_t, stream);
}

// pos_value >= 0: the position is known on the host (decode loop): the split launch does not wait for a load of *pos
static int kv_attn_impl(const void* q, const void* k_cache, const void* v_cache, const int32_t* pos, int32_t pos_value, void* out,
                        int64_t B, int64_t m, int64_t H, int64_t Lmax, float* workspace, int64_t nsplit,
                        const void* qkv_fused, const float* cos_t, const float* sin_t, void* stream) {
  LD_REQUIRE(k_cache && v_cache && pos && out, "ld_llm_kv_attn: null pointer");
  LD_REQUIRE(q || qkv_fused, "ld_llm_kv_attn: need q or qkv_fused");
  hipStream_t st = (hipStream_t)stream;
  if (m == 1 && nsplit > 1) {
    LD_REQUIRE(workspace, "ld_llm_kv_attn: split path needs a workspace of B*H*(nsplit*130 + 1) floats, the last B*H words zero");
    LD_REQUIRE(!qkv_fused || (cos_t && sin_t), "ld_llm_kv_attn: fused RoPE needs the cos/sin tables");
    LD_REQUIRE((Lmax + nsplit - 1) / nsplit <= 16 * KV_MAXIT, "ld_llm_kv_attn: Lmax=%ld needs nsplit >= %ld (<= 256 keys per split)",
               (long)Lmax, (long)((Lmax + 255) / 256));
    // the position is known on the host: do not even launch the splits this step leaves unused
    const int ny = pos_value >= 0 ? kv_eff_splits(pos_value + 1, (int)nsplit, ld_kv_split_rule()) : (int)nsplit;
    hipLaunchKernelGGL(ld_kv_attn_split_kernel<false>, dim3((unsigned)(B * H), (unsigned)ny), dim3(256), 0, st,
                       (const bf16_t*)q, (const bf16_t*)qkv_fused, cos_t, sin_t, (bf16_t*)k_cache, (bf16_t*)v_cache,
                       (const int*)pos, (int)pos_value, workspace, (unsigned*)(workspace + B * H * nsplit * 130), (bf16_t*)out, (int)B,
                       (int)H, (int)Lmax, (int)nsplit, ld_kv_split_rule(), ChainSync{});
    return ld_check_launch("ld_llm_kv_attn(split)");
  }
  LD_REQUIRE(q && !qkv_fused, "ld_llm_kv_attn: the fused RoPE/append form exists only for the decode split path");
  const size_t smem = (size_t)(Lmax + 8 + 256) * sizeof(float);
  LD_REQUIRE(smem <= 160 * 1024, "ld_llm_kv_attn: Lmax=%ld too long for the LDS score buffer", (long)Lmax);
  static thread_local LdSmemCache cache{};
  if (int rc = ld_ensure_dyn_smem((const void*)ld_kv_attn_kernel, smem, &cache)) return rc;
  hipLaunchKernelGGL(ld_kv_attn_kernel, dim3((unsigned)(B * H), (unsigned)m), dim3(256), smem, st,
                     (const bf16_t*)q, (const bf16_t*)k_cache, (const bf16_t*)v_cache, (const int*)pos,
                     (bf16_t*)out, (int)B, (int)m, (int)H, (int)Lmax);
  return ld_check_launch("ld_llm_kv_attn");
}

LD_API int ld_llm_embed(const float* table, const int64_t* token, void* out, int64_t B, int64_t D, void* stream) {
  LD_REQUIRE(table && token && out, "ld_llm_embed: null pointer");
  hipLaunchKernelGGL(ld_embed_kernel, dim3((B * D + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     table, (const long*)token, (bf16_t*)out, (int)B, (int)D);
  return ld_check_launch("ld_llm_embed");
}

LD_API int ld_llm_decode_forward(const ld_llm_layer* layers, int64_t n_layers, const float* emb_table, const int64_t* token,
                                 const int32_t* pos, int32_t pos_value, void* x, void* qkv, void* att, void* gate, float* attn_ws,
                                 const float* cos_t, const float* sin_t, const float* lnf_w, const float* lnf_b,
                                 float* lnf_out, const float* head_w, float* logits, int64_t B, int64_t hidden,
                                 int64_t heads, int64_t mlp, int64_t vocab, int64_t Lmax, int64_t nsplit, float rms_eps,
                                 float ln_eps, void* stream) {
  LD_REQUIRE(layers && n_layers > 0 && (emb_table == nullptr || token) && pos && x && qkv && att && gate && attn_ws && cos_t && sin_t &&
             lnf_w && lnf_b && lnf_out && head_w && logits, "ld_llm_decode_forward: null pointer");
  LD_REQUIRE(hidden == heads * 128, "ld_llm_decode_forward: head_dim must be 128 (hidden=%ld heads=%ld)", (long)hidden, (long)heads);
  LD_REQUIRE(nsplit > 1, "ld_llm_decode_forward: the decode path is the key-split attention (nsplit > 1)");
  LD_REQUIRE(pos_value < Lmax, "ld_llm_decode_forward: pos_value %d outside [0, Lmax)", (int)pos_value);
  int rc = emb_table ? ld_llm_embed(emb_table, token, x, B, hidden, stream) : 0;      // null: x already holds the token's embedding rows
  for (int64_t i = 0; i < n_layers && rc == 0; ++i) {
    const ld_llm_layer& w = layers[i];
    LD_REQUIRE(w.wqkv && w.wo && w.w1 && w.w3 && w.w2 && w.n0 && w.n1 && w.k_cache && w.v_cache,
               "ld_llm_decode_forward: layer %ld has a null pointer", (long)i);
    rc = ld_gemv(x, hidden, 0, w.wqkv, nullptr, 0, nullptr, nullptr, 0, qkv, 3 * hidden, 0, B, 3 * hidden, hidden, 0, 0,
                 w.n0, rms_eps, stream);
    if (rc) break;
    rc = kv_attn_impl(nullptr, w.k_cache, w.v_cache, pos, pos_value, att, B, 1, heads, Lmax, attn_ws, nsplit, qkv, cos_t, sin_t, stream);
    if (rc) break;
    rc = ld_gemv(att, hidden, 0, w.wo, nullptr, 0, nullptr, x, hidden, x, hidden, 0, B, hidden, hidden, 0, 0, nullptr, 0.f, stream);
    if (rc) break;
    rc = ld_gemv(x, hidden, 0, w.w1, w.w3, 0, nullptr, nullptr, 0, gate, mlp, 0, B, mlp, hidden, 0, LD_ACT_GELU_TANH,
                 w.n1, rms_eps, stream);
    if (rc) break;
    rc = ld_gemv(gate, mlp, 0, w.w2, nullptr, 0, nullptr, x, hidden, x, hidden, 0, B, hidden, mlp, 0, 0, nullptr, 0.f, stream);
  }
  if (rc) return rc;
  rc = ld_layernorm_bf16_to_f32(x, hidden, lnf_w, lnf_b, lnf_out, B, hidden, ln_eps, stream);
  if (rc) return rc;
  return ld_gemv(lnf_out, hidden, 1, head_w, nullptr, 1, nullptr, nullptr, 0, logits, vocab, 1, B, vocab, hidden, 0, 0,
                 nullptr, 0.f, stream);
}

#ifdef LD_VARIANTS   // the dependent-launch form of a decode step: measured slower, only in the variants build
namespace {
// the register GEMV in its dependent-launch form (B = 2): variant choice of launch_gemv_b, <= 128 registers, <= 512 workgroups
template <int R, int J, bool GATED, bool NORM>
int launch_gemv_chain(const GemvParams& p, hipStream_t st, int* grid_out) {
  const int cap = 512;
  const int nbatch = (p.N + R - 1) / R;
  const int trips = (nbatch + cap - 1) / cap;
  const int grid = (nbatch + trips - 1) / trips;
  *grid_out = grid;
  hipLaunchKernelGGL((ld_gemv_reg_kernel<2, R, J, GATED, NORM, true>), dim3((unsigned)grid), dim3(256), 0, st, p, nbatch);
  return ld_check_launch("ld_gemv(chained)");
}
int gemv_chain(const GemvParams& p, hipStream_t st, int* grid_out) {
  const int nchunk = p.K >> 3;
  const bool gated = p.W2 != nullptr, norm = p.norm_w != nullptr;
  if (nchunk <= 256) {
    if (gated) return norm ? launch_gemv_chain<4, 1, true, true>(p, st, grid_out) : ld_set_error(LD_ERR_UNSUPPORTED, "chained gemv: gated without norm");
    return norm ? launch_gemv_chain<4, 1, false, true>(p, st, grid_out) : launch_gemv_chain<4, 1, false, false>(p, st, grid_out);
  }
  if (nchunk <= 512) {
    if (gated) return norm ? launch_gemv_chain<2, 2, true, true>(p, st, grid_out) : ld_set_error(LD_ERR_UNSUPPORTED, "chained gemv: gated without norm");
    return norm ? launch_gemv_chain<4, 2, false, true>(p, st, grid_out) : launch_gemv_chain<4, 2, false, false>(p, st, grid_out);
  }
  if (nchunk <= 1536 && !gated && !norm) return launch_gemv_chain<1, 6, false, false>(p, st, grid_out);
  return ld_set_error(LD_ERR_UNSUPPORTED, "chained gemv: K = %d outside the register forms", p.K);
}
}  // namespace

LD_API int ld_llm_decode_blocks_chained(const ld_llm_layer* layers, int64_t n_layers, int32_t pos_value, void* x, void* qkv, void* att,
                                        void* gate, float* attn_ws, const float* cos_t, const float* sin_t, int64_t B, int64_t hidden,
                                        int64_t heads, int64_t mlp, int64_t Lmax, int64_t nsplit, float rms_eps, uint32_t* ctl,
                                        uint32_t epoch, void* stream0, void* stream1) {
  LD_REQUIRE(layers && n_layers > 0 && x && qkv && att && gate && attn_ws && cos_t && sin_t && ctl, "ld_llm_decode_blocks_chained: null pointer");
  LD_REQUIRE(pos_value >= 0 && pos_value < Lmax, "ld_llm_decode_blocks_chained: position %d outside [0, Lmax)", (int)pos_value);
  LD_REQUIRE(stream0 != stream1, "ld_llm_decode_blocks_chained: needs two different streams");
  if (B != 2 || hidden != heads * 128 || hidden % 8 || hidden > 4096 || mlp % 8 || mlp > 12288 || nsplit < 2 ||
      (Lmax + nsplit - 1) / nsplit > 16 * KV_MAXIT || 5 * n_layers > LD_LLM_CHAIN_MAX_OPS)
    return ld_set_error(LD_ERR_UNSUPPORTED, "ld_llm_decode_blocks_chained: B=%ld hidden=%ld heads=%ld mlp=%ld Lmax=%ld nsplit=%ld layers=%ld "
                        "outside the chained form", (long)B, (long)hidden, (long)heads, (long)mlp, (long)Lmax, (long)nsplit, (long)n_layers);
  hipStream_t st[2] = {(hipStream_t)stream0, (hipStream_t)stream1};
  int slot = 0, prev_grid = 0, rc = 0;
  auto sync = [&]() { return ChainSync{(unsigned*)ctl, slot, prev_grid, epoch + 1u}; };
  auto gemv = [&](const void* xin, int64_t ldx, const void* W, const void* W2, const void* resid, void* out, int64_t ldo, int64_t N, int64_t K,
                  int act, const float* norm_w) {
    GemvParams p{};
    p.x = xin; p.W = W; p.W2 = W2; p.resid = resid; p.out = out; p.B = 2; p.N = (int)N; p.K = (int)K; p.ldx = ldx; p.ldo = ldo; p.ldr = ldo;
    p.act = act; p.norm_w = norm_w; p.norm_eps = rms_eps; p.cs = sync();
    int grid = 0;
    const int r = gemv_chain(p, st[slot & 1], &grid);
    prev_grid = grid; ++slot;
    return r;
  };
  for (int64_t i = 0; i < n_layers && rc == 0; ++i) {
    const ld_llm_layer& w = layers[i];
    LD_REQUIRE(w.wqkv && w.wo && w.w1 && w.w3 && w.w2 && w.n0 && w.n1 && w.k_cache && w.v_cache,
               "ld_llm_decode_blocks_chained: layer %ld has a null pointer", (long)i);
    rc = gemv(x, hidden, w.wqkv, nullptr, nullptr, qkv, 3 * hidden, 3 * hidden, hidden, 0, w.n0);
    if (rc) break;
    hipLaunchKernelGGL(ld_kv_attn_split_kernel<true>, dim3((unsigned)(B * heads), (unsigned)nsplit), dim3(256), 0, st[slot & 1],
                       (const bf16_t*)nullptr, (const bf16_t*)qkv, cos_t, sin_t, (bf16_t*)w.k_cache, (bf16_t*)w.v_cache, (const int*)nullptr,
                       (int)pos_value, attn_ws, (unsigned*)(attn_ws + B * heads * nsplit * 130), (bf16_t*)att, (int)B, (int)heads,
                       (int)Lmax, (int)nsplit, ld_kv_split_rule(), sync());
    rc = ld_check_launch("ld_llm_kv_attn(chained)");
    if (rc) break;
    prev_grid = (int)(B * heads * nsplit); ++slot;
    rc = gemv(att, hidden, w.wo, nullptr, x, x, hidden, hidden, hidden, 0, nullptr);
    if (rc) break;
    rc = gemv(x, hidden, w.w1, w.w3, nullptr, gate, mlp, mlp, hidden, LD_ACT_GELU_TANH, w.n1);
    if (rc) break;
    rc = gemv(gate, mlp, w.w2, nullptr, x, x, hidden, hidden, mlp, 0, nullptr);
  }
  return rc;
}
#endif  // LD_VARIANTS

LD_API int ld_llm_logits_to_probs(const float* logits, float* probs, float* cfg_logits, int64_t V, int32_t guided,
                                  float scale, float temperature, const int32_t* pos, const int32_t* allowed,
                                  int64_t allowed_stride, int32_t top_k, float top_p, void* stream) {
  LD_REQUIRE(logits && probs && V > 0 && V <= LD_SAMPLE_MAXV, "ld_llm_logits_to_probs: bad args (V=%ld, max %d)", (long)V, LD_SAMPLE_MAXV);
  LD_REQUIRE(!allowed || pos, "ld_llm_logits_to_probs: allowed table needs pos");
  hipLaunchKernelGGL(ld_logits_to_probs_kernel<false>, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, probs, cfg_logits,
                     (int)V, guided, scale, temperature, (const int*)pos, (const int*)allowed, (int)allowed_stride,
                     (int)top_k, top_p, SampleArgs{});
  return ld_check_launch("ld_llm_logits_to_probs");
}

LD_API int ld_llm_sample_advance(const float* logits, float* probs, float* cfg_logits, int64_t V, int32_t guided, float scale,
                                 float temperature, int32_t* pos, const int32_t* allowed, int64_t allowed_stride,
                                 int32_t top_k, float top_p, const float* noise, const int32_t* forced, int64_t* token,
                                 int64_t* out_tokens, int32_t* out_count, int64_t* sampled, const float* emb_table, void* x,
                                 int64_t B, int64_t D, void* stream) {
  LD_REQUIRE(logits && V > 0 && V <= LD_SAMPLE_MAXV, "ld_llm_sample_advance: bad args (V=%ld, max %d)", (long)V, LD_SAMPLE_MAXV);
  LD_REQUIRE(pos && noise && forced && token && out_tokens && out_count && emb_table && x, "ld_llm_sample_advance: null pointer");
  SampleArgs sa{noise, (const int*)forced, (int*)pos, (long*)token, (long*)out_tokens, (int*)out_count, (long*)sampled,
                emb_table, (bf16_t*)x, (int)B, (int)D};
  hipLaunchKernelGGL(ld_logits_to_probs_kernel<true>, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, probs, cfg_logits,
                     (int)V, guided, scale, temperature, (const int*)pos, (const int*)allowed, (int)allowed_stride,
                     (int)top_k, top_p, sa);
  return ld_check_launch("ld_llm_sample_advance");
}

LD_API int ld_llm_decode_advance(const int64_t* sampled, const int32_t* forced, int32_t* pos, int64_t* token,
                                 int64_t* out_tokens, int32_t* out_count, void* stream) {
  LD_REQUIRE(sampled && forced && pos && token && out_tokens && out_count, "ld_llm_decode_advance: null pointer");
  hipLaunchKernelGGL(ld_decode_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (const long*)sampled,
                     (const int*)forced, (int*)pos, (long*)token, (long*)out_tokens, (int*)out_count);
  return ld_check_launch("ld_llm_decode_advance");
}

// Flash-style fused attention for gfx950, head_dim 64, bf16 in / fp32 accumulate / bf16 out.
//
// Replaces:
//   sat attention_fn_default -> F.scaled_dot_product_attention on the joint text+video sequence
//   (landiff/diffusion/dit_video_concat.py:636-664; called from :587,:1308), and
//   flex_attention(block_mask=VideoDecoderMask) in the TiTok decoder
//   (landiff/tokenizer/modules/blocks.py:172-212; mask landiff/tokenizer/modules/flex_attention_mask.py:193-335,
//   whose closed form is  allowed(q,kv) <=> fid[kv] <= fid[q]  -- SURVEY.md Appendix B).
//
// MI355X-first structure:
//   * one workgroup = 4 wave64 = 128 query rows (32 per wave); K and V^T tiles of 64 keys are
//     brought HBM->LDS by LDS-DMA (global_load_lds_dwordx4), double buffered, one barrier/tile.
//   * S^T = K Q^T with v_mfma_f32_32x32x16_bf16 ("swapped" product): a lane then owns ONE query
//     column (lane&31) and 32 of the 64 keys, so the softmax row reductions are lane-local
//     plus a single exchange with lane^32 (v_permlane32_swap).
//   * the K rows fed to the MFMA are permuted (bits 2<->3 of the row index swapped in the LDS
//     read address -- free), which makes the 8 accumulator registers [8*ks .. 8*ks+7] exactly
//     the 8 consecutive keys the PV MFMA wants as its B fragment: no cross-lane traffic, no
//     P round trip through LDS.
//   * V is consumed as V^T [d][key] (written in that layout by the qkv-split kernel), so the
//     PV A-fragment is a plain ds_read_b128; O^T accumulates per (d-row, q-column) and the
//     online-softmax rescale is a per-lane scalar multiply.
//   * LDS tiles are lane-linear (DMA) with the 16-B chunk XOR-swizzled by ((row>>1)&7) on the
//     source address and on the read: conflict-free ds_read_b128.
//   * frame-block mask: per-KV-tile [min,max] frame ids let a workgroup skip fully masked
//     tiles (no load, no MFMA) and apply the element mask only on the few straddling tiles.
#include "ld_attn.h"

#ifndef LD_ATTN_PLAIN_WPS
#define LD_ATTN_PLAIN_WPS 3      // register budget (waves per SIMD) of the default instantiation
#endif
#define LD_STR_(x) #x
#define LD_STR(x) LD_STR_(x)

namespace {

template <int TPS, bool DEFER, bool PRIO, bool MFMASUM, int WPS>   // WPS = waves/SIMD of the register budget
__global__ __launch_bounds__(256, WPS) void ld_attn_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages x TPS x (K 8 KB + V^T 8 KB) + 64 B
  constexpr int STAGE = TPS * STAGE_BYTES;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;
  const int nqb = p.Npad / QB;
  const int nkt_all = (p.Nk + KT - 1) / KT;
  const int nst_all = (nkt_all + TPS - 1) / TPS;

  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = bid / nqb, qb = bid - bh * nqb;
  const int b = bh / p.H, h = bh - b * p.H;

  const bf16_t* Qb = p.Q + (long)bh * p.Npad * D;
  const bf16_t* Kb = p.K + (long)bh * p.Npad * D;
  const bf16_t* Vb = p.Vt + (long)bh * D * p.Npad;

  const int q = qb * QB + wave * 32 + (lane & 31);   // this lane's query row
  if (qb * QB >= p.Nq) return;                        // whole block past the end (uniform)

  // ---- frame ids of the queries; block/wave ranges ----
  const bool masked = p.fid_k != nullptr;
  int qfid = 0, wqmin = 0, wqmax = 0, bqmax = 0;
  if (masked) {
    const int qc = q < p.Nq ? q : p.Nq - 1;
    qfid = p.fid_q[qc];
    int mn = qfid, mx = qfid;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn = min(mn, __shfl_xor(mn, o, 64));
      mx = max(mx, __shfl_xor(mx, o, 64));
    }
    wqmin = mn; wqmax = mx;
    int* red = (int*)(smem + 2 * STAGE);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    bqmax = max(max(red[0], red[1]), max(red[2], red[3]));
  }

  // ---- Q fragments (B operand of K Q^T): lane = query column, 8 consecutive d per k-step ----
  // Q is pre-multiplied by softmax_scale*log2(e) (one extra bf16 rounding of q, 2^-9 relative) so that the MFMA
  // accumulator is directly the exp2 argument: started at -m (running max, log2 units) it needs no per-element FMA.
  bf16x8_t qf[4];
  {
    const bf16_t* qrow = Qb + (long)q * D + hi * 8;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const u32x4_t raw = *(const u32x4_t*)(qrow + kk * 16);
      u32x4_t sc;
#pragma unroll
      for (int e = 0; e < 4; ++e) sc[e] = pack_bf16x2(bf_lo(raw[e]) * p.c, bf_hi(raw[e]) * p.c);
      qf[kk] = __builtin_bit_cast(bf16x8_t, sc);
    }
  }

  // ---- LDS-DMA source offsets: waves 0,1 stage K (rows = keys), waves 2,3 stage V^T (rows = d) ----
  long goff[4];
  int ldsoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = wave * 4 + i;            // 0..15 (1 KB pieces of one 64-key tile)
    const int r = (idx & 7) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((r >> 1) & 7);
    if (idx < 8) goff[i] = (long)r * D + chunk * 8;          // + kv0 * D
    else goff[i] = (long)r * p.Npad + chunk * 8;             // + kv0
    ldsoff[i] = (idx < 8 ? 0 : KTILE_BYTES) + (idx & 7) * 1024;
  }
  auto stage = [&](int buf, int st) {
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int j = 0; j < TPS; ++j) {
      const int t = st * TPS + j;
      if (t < nkt_all) {
        const long kv0 = (long)t * KT;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int idx = wave * 4 + i;
          const bf16_t* src = (idx < 8) ? (Kb + kv0 * D + goff[i]) : (Vb + kv0 + goff[i]);
          glds16(src, base + j * STAGE_BYTES + ldsoff[i]);
        }
      }
    }
  };
  auto stage_needed = [&](int st) {
    if (!masked) return true;
    bool need = false;
#pragma unroll
    for (int j = 0; j < TPS; ++j) {
      const int t = st * TPS + j;
      if (t < nkt_all && p.kt_min[t] <= bqmax) need = true;
    }
    return need;
  };
  auto next_stage = [&](int st) {
    ++st;
    while (st < nst_all && !stage_needed(st)) ++st;
    return st;
  };

  // fragment read byte offsets within a 64-key tile, one per (row block i, k-step): loop invariant, so the
  // per-tile LDS addressing is just "register + immediate" (the stage/buffer offsets are compile-time constants)
  int kofs[4], vofs[4];       // row block i = 1 is +32 rows = +4096 B with the same swizzle key: an immediate offset
  {
    const int key = swap23(lane & 31);
    const int d = lane & 31;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int c = kk * 2 + hi;
      kofs[kk] = key * 128 + ((c ^ ((key >> 1) & 7)) << 4);
      vofs[kk] = KTILE_BYTES + d * 128 + ((c ^ ((d >> 1) & 7)) << 4);
    }
  }

  f32x16_t o[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
  // running max m (log2 units) lives negated in a 16-register block that is the C operand of the first QK^T MFMA;
  // it only changes in the rare rescale branch (deferred: P may grow to 2^THR before we rescale)
  f32x16_t negm;
#pragma unroll
  for (int r = 0; r < 16; ++r) negm[r] = 0.f;
  float negm0 = 0.f;                 // scalar copy used by the lean-register (WPS >= 4) form
  float lsum = 0.f;
  // MFMASUM: the softmax denominator comes out of the matrix pipe (an all-ones A fragment times P), not from 32 VALU adds
  f32x16_t lacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) lacc[r] = 0.f;
  const bf16x8_t ones = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
  bool unset = true;                 // no valid key seen yet for this query
  constexpr float THR = 8.0f;

  auto tile = [&](auto bufc, auto jc, int t) {
    constexpr int OFF = decltype(bufc)::value * STAGE + decltype(jc)::value * STAGE_BYTES;
    bool skip = false, need_mask = false;
    if (masked) {
      const int tmin = p.kt_min[t], tmax = p.kt_max[t];
      skip = tmin > wqmax;
      need_mask = tmax > wqmin;
    } else {
      need_mask = (t + 1) * KT > p.Nk;
    }
    if (skip) return;
    // ---- S^T = K Q^T: all 8 K fragments are requested up front (one exposed LDS latency per tile, not four);
    //      the first k-step uses the inline-constant 0 as C, so the accumulators are never zeroed by VALU moves ----
    f32x16_t sacc[2];
    if (WPS >= 4) {
      // lean-register form (<= 128 VGPRs, 4 waves/SIMD): C starts at the inline constant 0, K fragments in two halves
      const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        bf16x8_t kh[2][2];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
          for (int i = 0; i < 2; ++i) kh[i][k2] = *(const bf16x8_t*)(smem + kofs[half * 2 + k2] + OFF + i * 4096);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            sacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh[i][k2], qf[half * 2 + k2],
                                                              (half == 0 && k2 == 0) ? zero16 : sacc[i], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[i][r] += negm0;        // S' - m_old
    } else {
    bf16x8_t kf[2][4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i) kf[i][kk] = *(const bf16x8_t*)(smem + kofs[kk] + OFF + i * 4096);
    if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        sacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i][kk], qf[kk], kk == 0 ? negm : sacc[i], 0, 0, 0);
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    }
    // V^T fragments for the PV product: requested now so their LDS latency hides under the softmax VALU work
    bf16x8_t vf[2][4];
    if (WPS < 4) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i) vf[i][ks] = *(const bf16x8_t*)(smem + vofs[ks] + OFF + i * 4096);
    }
    // register r of sacc[i] holds key  t*64 + i*32 + (r>>3)*16 + hi*8 + (r&7)
    auto apply_mask = [&]() {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int key0 = t * KT + i * 32 + g * 16 + hi * 8;
          if (masked) {
            const int4 f0 = *(const int4*)(p.fid_k + key0);
            const int4 f1 = *(const int4*)(p.fid_k + key0 + 4);
            const int f[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (f[e] > qfid) sacc[i][g * 8 + e] = NEG_BIG;
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (key0 + e >= p.Nk) sacc[i][g * 8 + e] = NEG_BIG;
          }
        }
    };
    if (need_mask) apply_mask();
    // ---- online softmax (per query column; lane and lane^32 share the row) ----
    // max3f is an asm statement and its operands are MFMA results: hipcc pads the MFMA -> VALU read hazard only for instructions it
    // knows, so the wait states go here, in a statement that names the accumulators (round 6: the same helper right behind its
    // MFMAs in ld_attn_q64.hip's row-maximum pass read OLD accumulator values now and then -- run-to-run differences of one ulp)
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(sacc[0]), "+v"(sacc[1]));
    float mx = max3f(sacc[0][0], sacc[1][0], sacc[0][1]);
    mx = max3f(mx, sacc[1][1], sacc[0][2]);
#pragma unroll
    for (int r = 2; r < 15; ++r) mx = max3f(mx, sacc[1][r], sacc[0][r + 1]);
    mx = fmaxf(mx, sacc[1][15]);
    mx = lane32_max(mx);            // = max_k(S') - m_old for this query (S' in log2 units)
    const bool valid = mx > -1.0e29f;                       // at least one unmasked key in this tile
    if (!__all(!(valid && (unset || mx > (DEFER ? THR : 0.0f))))) {
      // rare: (re)base the running max.  d = how much m grows for this lane; everything held at the old max scales by 2^-d
      const float d = valid ? (unset ? mx : fmaxf(mx, 0.0f)) : 0.0f;
      unset = unset && !valid;
      const float alpha = __builtin_amdgcn_exp2f(-d);
      lsum *= alpha;
      if (MFMASUM) lacc[0] *= alpha;                       // only row 0 of the ones-product is ever read
      if (WPS >= 4) negm0 -= d;
      else {
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[r] -= d;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[i][r] *= alpha; sacc[i][r] -= d; }
    }
    float psum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(sacc[i][r]);
        sacc[i][r] = pv;
        if (!MFMASUM) psum += pv;
      }
    lsum += psum;
    // ---- O^T += V^T P^T ----
    if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4_t pw;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        pw[e] = pack_bf16x2(sacc[ks >> 1][(ks & 1) * 8 + 2 * e], sacc[ks >> 1][(ks & 1) * 8 + 2 * e + 1]);
      const bf16x8_t pb = __builtin_bit_cast(bf16x8_t, pw);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bf16x8_t va = (WPS >= 4) ? *(const bf16x8_t*)(smem + vofs[ks] + OFF + i * 4096) : vf[i][ks];
        o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pb, o[i], 0, 0, 0);
      }
      if (MFMASUM) lacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pb, lacc, 0, 0, 0);
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
  };
  auto stage_tiles = [&](auto bufc, int st) {
    tile(bufc, std::integral_constant<int, 0>{}, st * TPS);
    if (TPS > 1 && st * TPS + 1 < nkt_all) tile(bufc, std::integral_constant<int, TPS - 1>{}, st * TPS + 1);
  };

  // two stages per trip with compile-time buffer indices; the second half sits under an `if`, not behind a `break`
  // (a mid-loop exit makes hipcc keep two copies of the O accumulators and shuffle them every trip)
  int st = next_stage(-1);
  if (st < nst_all) stage(0, st);
  while (st < nst_all) {
    int stn = next_stage(st);
    __syncthreads();
    if (stn < nst_all) stage(1, stn);
    stage_tiles(std::integral_constant<int, 0>{}, st);
    st = stn;
    if (st < nst_all) {
      stn = next_stage(st);
      __syncthreads();
      if (stn < nst_all) stage(0, stn);
      stage_tiles(std::integral_constant<int, 1>{}, st);
      st = stn;
    }
  }

  // ---- finalize: O = O^T / l, write bf16 rows ----
  const float ltot = MFMASUM ? lacc[0] : lane32_sum(lsum);
  const float inv = ltot > 0.f ? 1.0f / ltot : 0.f;
  if (q < p.Nq) {
    bf16_t* orow = p.O + (long)b * p.o_bs + (long)q * p.o_rs + h * D;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = i * 32 + 8 * g + 4 * hi;
        u32x2_t w2;
        w2[0] = pack_bf16x2(o[i][4 * g + 0] * inv, o[i][4 * g + 1] * inv);
        w2[1] = pack_bf16x2(o[i][4 * g + 2] * inv, o[i][4 * g + 3] * inv);
        *(u32x2_t*)(orow + d0) = w2;
      }
  }
}

}  // namespace

int ld_attn_p16_launch(const AttnParams& p, hipStream_t st);     // ld_attn_p16.hip
int ld_attn_q64_launch(const AttnParams& p, hipStream_t st);     // ld_attn_q64.hip
#ifdef LD_VARIANTS   // measured alternatives, only in the variants build (build.sh: LD_BUILD_VARIANTS=1)
int ld_attn_pipe2_launch(const AttnParams& p, hipStream_t st);   // ld_attn_pipe.hip
int ld_attn_q128_launch(const AttnParams& p, hipStream_t st);    // ld_attn_q128.hip
#endif

// name of the kernel the calling thread's last ld_attn_fwd_bf16 launched (bench.py labels its roofline object with it)
static thread_local const char* g_attn_last_kernel = "";
void ld_attn_set_last_kernel(const char* name) { g_attn_last_kernel = name; }
LD_API const char* ld_attn_last_kernel(void) { return g_attn_last_kernel; }

int ld_attn_q64_exact_launch(const AttnParams& p, hipStream_t st);   // ld_attn_q64_exact.hip

// Where the calling thread's last launch counts the query blocks that left the fast pass's window (ld_attn_last_fallbacks):
// kind 0 = the kernel has no such window (running-max or two-pass softmax: nothing to count), 1 = counted at `src` (device address,
// valid until the next launch on that stream), 2 = a max-free fast pass that keeps no count (static dispatch, 32-row tile).
static thread_local const unsigned* g_attn_fb_src = nullptr;
static thread_local int g_attn_fb_kind = 0;
void ld_attn_set_fallback_source(const unsigned* src, int kind) { g_attn_fb_src = src; g_attn_fb_kind = kind; }

LD_API int ld_attn_last_fallbacks(int32_t* out, void* stream) {
  LD_REQUIRE(out, "ld_attn_last_fallbacks: null pointer");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e;
  if (g_attn_fb_kind == 1) e = hipMemcpyAsync(out, g_attn_fb_src, 4, hipMemcpyDeviceToDevice, st);
  else e = hipMemsetAsync(out, g_attn_fb_kind == 2 ? 0xFF : 0, 4, st);
  if (e != hipSuccess) return ld_set_error(LD_ERR_LAUNCH, "ld_attn_last_fallbacks: %s", hipGetErrorString(e));
  return LD_OK;
}

static int attn_fwd_impl(const void* Q, const void* K, const void* Vt, void* O,
                         int64_t B, int64_t H, int64_t Nq, int64_t Nk, int64_t Npad,
                         int64_t o_batch_stride, int64_t o_row_stride, float softmax_scale,
                         const int32_t* fid_q, const int32_t* fid_k,
                         const int32_t* kt_min, const int32_t* kt_max, void* stream, bool exact) {
  LD_REQUIRE(Q && K && Vt && O, "ld_attn_fwd_bf16: null pointer");
  LD_REQUIRE(B > 0 && H > 0 && Nq > 0 && Nk > 0, "ld_attn_fwd_bf16: empty problem");
  LD_REQUIRE(Npad % QB == 0 && Npad >= Nq && Npad >= Nk, "ld_attn_fwd_bf16: Npad=%ld must be a multiple of %d and >= Nq,Nk", (long)Npad, QB);
  LD_REQUIRE((fid_q == nullptr) == (fid_k == nullptr), "ld_attn_fwd_bf16: fid_q and fid_k go together");
  LD_REQUIRE(!fid_k || (kt_min && kt_max), "ld_attn_fwd_bf16: masked attention needs kt_min/kt_max");
  LD_REQUIRE(o_row_stride % 4 == 0 && ((uintptr_t)O & 7) == 0, "ld_attn_fwd_bf16: output alignment");
  AttnParams p{};
  p.Q = (const bf16_t*)Q; p.K = (const bf16_t*)K; p.Vt = (const bf16_t*)Vt; p.O = (bf16_t*)O;
  p.B = (int)B; p.H = (int)H; p.Nq = (int)Nq; p.Nk = (int)Nk; p.Npad = (int)Npad;
  p.o_bs = o_batch_stride; p.o_rs = o_row_stride;
  p.c = softmax_scale * 1.4426950408889634f;
  p.fid_q = fid_q; p.fid_k = fid_k; p.kt_min = kt_min; p.kt_max = kt_max;
  const int nqb = (int)(Npad / QB);
  dim3 grid((unsigned)(B * H * nqb)), block(256);
  static int var = -1;
  if (var < 0) {
    // tuning knob: 0 = default (pipelined 16x16x32 kernels of ld_attn_q64.hip / ld_attn_p16.hip for every unmasked problem of >= 6 key tiles, else the plain kernel),
    // 8 = the pipelined 32x32x16 kernel of ld_attn_pipe.hip (round-1 default; variants build only), 9 = plain kernel everywhere,
    // 1 / 4 = plain kernel with row sums on the matrix pipe / lean-register 4-waves form
    const char* e = getenv("LD_ATTN_VARIANT");
    var = e ? atoi(e) : 0;
  }
  hipStream_t st = (hipStream_t)stream;
  ld_attn_set_fallback_source(nullptr, 0);          // (the pipelined launchers below say otherwise)
  const size_t s1 = 2 * STAGE_BYTES + 64;
  const int64_t nkt = (Nk + KT - 1) / KT;
  // LD_ATTN_Q64=0 (tuning knob): the 32-query-row wave tile of ld_attn_p16.hip instead of the 64-row one of ld_attn_q64.hip
  static const int q64 = getenv("LD_ATTN_Q64") ? atoi(getenv("LD_ATTN_Q64")) : 1;
#ifdef LD_VARIANTS
  // LD_ATTN_Q128 (variants build only): 1 = the 128-query-row, one-wave-per-SIMD tile of ld_attn_q128.hip (512 rows per workgroup:
  // only for problems with enough query rows to fill the chip with such workgroups); 2 = every unmasked problem of >= 6 key tiles.
  // Like every knob it is read once, or per call under LD_TUNING=1 (ld_common.h) so that one process can time both tiles.
  static int k_q128 = LD_KNOB_UNSET;
  const int q128 = ld_knob("LD_ATTN_Q128", 0, &k_q128);
  if (var == 0 && !exact && !fid_k && nkt >= 6 && (q128 == 2 || (q128 == 1 && B * H * ((Npad + 511) / 512) >= 512)))
    return ld_attn_q128_launch(p, st);
#endif
  // exact form: the two-pass kernel where the pipelined tile applies; every other shape takes the plain kernel below, whose online
  // softmax is exact for any logit range already
  if (exact && !fid_k && nkt >= 6) return ld_attn_q64_exact_launch(p, st);
  if (var == 0 && !fid_k && nkt >= 6 && q64 && !exact) return ld_attn_q64_launch(p, st);
  if (var == 0 && !fid_k && nkt >= 6 && !exact) return ld_attn_p16_launch(p, st);        // any tile count
#ifdef LD_VARIANTS
  if (var == 8 && !exact && !fid_k && nkt >= 6 && (nkt - 2) % 4 == 0) return ld_attn_pipe2_launch(p, st);   // (the round-1 kernel: tile counts 4 m + 2)
#endif
  if (var == 1) {
    g_attn_last_kernel = "ld_attn_kernel<1,true,false,true,2>";
    hipLaunchKernelGGL((ld_attn_kernel<1, true, false, true, 2>), grid, block, s1, st, p);
  } else if (var == 4) {
    g_attn_last_kernel = "ld_attn_kernel<1,true,false,false,4>";
    hipLaunchKernelGGL((ld_attn_kernel<1, true, false, false, 4>), grid, block, s1, st, p);
  } else {
    g_attn_last_kernel = "ld_attn_kernel<1,true,false,false," LD_STR(LD_ATTN_PLAIN_WPS) ">";
    hipLaunchKernelGGL((ld_attn_kernel<1, true, false, false, LD_ATTN_PLAIN_WPS>), grid, block, s1, st, p);
  }
  return ld_check_launch("ld_attn_fwd_bf16");
}

LD_API int ld_attn_fwd_bf16(const void* Q, const void* K, const void* Vt, void* O,
                            int64_t B, int64_t H, int64_t Nq, int64_t Nk, int64_t Npad,
                            int64_t o_batch_stride, int64_t o_row_stride, float softmax_scale,
                            const int32_t* fid_q, const int32_t* fid_k,
                            const int32_t* kt_min, const int32_t* kt_max, void* stream) {
  return attn_fwd_impl(Q, K, Vt, O, B, H, Nq, Nk, Npad, o_batch_stride, o_row_stride, softmax_scale, fid_q, fid_k, kt_min, kt_max, stream, false);
}

LD_API int ld_attn_fwd_bf16_exact(const void* Q, const void* K, const void* Vt, void* O,
                                  int64_t B, int64_t H, int64_t Nq, int64_t Nk, int64_t Npad,
                                  int64_t o_batch_stride, int64_t o_row_stride, float softmax_scale,
                                  const int32_t* fid_q, const int32_t* fid_k,
                                  const int32_t* kt_min, const int32_t* kt_max, void* stream) {
  return attn_fwd_impl(Q, K, Vt, O, B, H, Nq, Nk, Npad, o_batch_stride, o_row_stride, softmax_scale, fid_q, fid_k, kt_min, kt_max, stream, true);
}

// All transformer blocks of one AR decode step as ONE persistent launch.
//
// Replaces the same reference code as the per-operation chain in ld_llm.hip (LlamaTransformerBlock.local_kvcache_inference,
// landiff/llm/modules/transformer_blocks.py:128-236; GPT.sample's layer loop, landiff/llm/models/transformer.py:91-119) and
// produces the same bits: every 256-thread sub-group runs the per-launch kernels' schedule (K split over 256 threads,
// v_dot2 partial sums, the same wave butterfly and 4-wave LDS sum; one (batch row, head, key split) per sub-group in the
// attention), only the assignment of rows to sub-groups differs.
//
// Why one launch: the chain is 6 dependent operations per block (qkv | split attention | combine | wo | w1.w3 | w2), 144
// per step, each streaming 8-90 MB of weights.  As separate launches every one pays the launch gap, a cold start (first weight
// request only after the launch has been dispatched everywhere) and a drain; the HBM pipe idles in between -- 3.2 TB/s
// over the step although each GEMV streams at 4.5-5 TB/s.  Here one workgroup per CU stays resident, the operations are
// separated by a grid barrier, and the WEIGHTS OF THE NEXT OPERATION ARE ALREADY IN FLIGHT WHILE THE BARRIER IS WAITED FOR:
// weight rows do not depend on the step's activations, so the first two batches of the next operation are requested before
// the arrival (batch A before the last batch of the current operation is consumed, batch B after this thread's outputs have
// drained), the K / V rows of the attention before the qkv barrier, the wo and w1.w3 rows during the attention.
//
// Hand-off between operations (different CUs, different XCDs, no coherent L2 between XCDs): outputs are written with sc1
// (write-through to the device coherence point), `s_waitcnt vmcnt(0)` drains them, one relaxed agent-scope add arrives on one
// of 8 counters (separate cache lines, workgroup b on counter b & 7), lanes 0-7 of wave 0 poll the 8 counters, consumers
// read with sc1.  No cache write-back / invalidate is involved (tools/probe/grid_barrier.hip: 3.4 us per barrier + read at
// 256 workgroups against 8.8 us with release / acquire fences; no stale value in 200 phases).
// All workgroups must be co-resident: the host launches min(CUs, occupancy) workgroups, and a poll gives up after ~1 s
// (error flag in the control block, every later launch returns at once) instead of hanging the GPU.
#include "ld_common.h"
#include "ld_llm_dev.h"
#include <stdlib.h>
#include "../../include/landiff_hip.h"

namespace {

constexpr int SGN = 1;                 // 256-thread sub-groups per workgroup (1: one wave per SIMD, the whole register file for four waves)
constexpr int NT = 256 * SGN;
constexpr int FB = 2;                  // batch rows: (cond, uncond)
constexpr int NCTR = 8, CSTRIDE = 32, CTR0 = 32;     // ctl[0] epoch, ctl[1] error, ctl[CTR0 + c * CSTRIDE] arrival counters
constexpr int VMAX = 16;
constexpr unsigned SPIN_LIMIT = 1u << 20;
constexpr int AUX_NT = 2, AUX_SC1 = 16;              // gfx940+ cache-policy bits of the buffer builtins
constexpr int BARRIERS_PER_LAYER = 6;

struct FusedParams {
  const ld_llm_layer* layers; int n_layers;
  bf16_t *x, *qkv, *att, *gate; float* ws;
  const float *cos_t, *sin_t; const int* pos;
  int hidden, heads, mlp, Lmax, nsplit; float rms_eps;
  KvSplitRule rule;         // splits in use at context length L: kv_eff_splits (ld_llm_dev.h), the same rule as the per-operation launch
  unsigned* ctl;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}
// agent-coherent scalar accesses (sc1) for the vectors one operation writes and the next one reads
__device__ __forceinline__ float ldc(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stc(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bf16_t ldc(const bf16_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stc(bf16_t* p, bf16_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- one batch of weight rows in registers: R rows (x2 gated), this thread's chunks j * 256 + t, + its epilogue operand ----
template <int R, int J, bool GATED>
struct WB {
  u32x4_t w[R][J];
  u32x4_t g[GATED ? R : 1][J];
  uint32_t res;                        // raw bf16 bits of the residual operand of output (row t / FB, batch row t % FB)
};

// rows [row0, row0 + R) below row_end (both uniform over the sub-group); chunks past K and dead rows read as zero without
// touching memory (bounds-checked buffer loads: branch-free, nothing to wait for before the next request)
template <int R, int J, bool GATED>
__device__ __forceinline__ void wb_request(WB<R, J, GATED>& q, const bf16_t* W, const bf16_t* W2, int K, int row0, int row_end,
                                           int t, const bf16_t* resid, int ldr) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int live = (row0 + r < row_end) ? K * 2 : 0;
    const __amdgpu_buffer_rsrc_t rw = rsrc_of(W + (long)(row0 + r) * K, live);
#pragma unroll
    for (int j = 0; j < J; ++j) q.w[r][j] = __builtin_amdgcn_raw_buffer_load_b128(rw, (j * 256 + t) * 16, 0, AUX_NT);
    if (GATED) {
      const __amdgpu_buffer_rsrc_t rg = rsrc_of(W2 + (long)(row0 + r) * K, live);
#pragma unroll
      for (int j = 0; j < J; ++j) q.g[r][j] = __builtin_amdgcn_raw_buffer_load_b128(rg, (j * 256 + t) * 16, 0, AUX_NT);
    }
  }
  q.res = 0u;
  const int er = t / FB, eb = t - er * FB;
  if (resid && t < R * FB && row0 + er < row_end) q.res = ldc(resid + (long)eb * ldr + row0 + er);
}

// dot products of one batch, wave butterfly, 4-wave LDS sum, epilogue (ld_gemv_reg_kernel's arithmetic), sc1 store
template <int R, int J, bool GATED>
__device__ __forceinline__ void wb_consume(const WB<R, J, GATED>& q, const u32x4_t (&xq)[FB][J], float (*red)[VMAX], int row0,
                                           int row_end, int t, int lane, int wave, int act, bool has_res, bf16_t* out, int ldo) {
  constexpr int NV = R * FB * (GATED ? 2 : 1), V = ceil_pow2(NV), LOGV = ilog2(V);
  static_assert(V <= VMAX, "reduction scratch too small");
  float v[V];
#pragma unroll
  for (int i = 0; i < V; ++i) v[i] = 0.f;
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int b = 0; b < FB; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[r * FB + b] = dot2_bf16(q.w[r][j][e], xq[b][j][e], v[r * FB + b]);
          if (GATED) v[R * FB + r * FB + b] = dot2_bf16(q.g[r][j][e], xq[b][j][e], v[R * FB + r * FB + b]);
        }
  wave_sum_multi<V>(v, lane);
  if ((lane & ((64 >> LOGV) - 1)) == 0) red[wave][lane >> (6 - LOGV)] = v[0];
  __syncthreads();
  const int er = t / FB, eb = t - er * FB;
  if (t < R * FB && row0 + er < row_end) {
    float a = red[0][t] + red[1][t] + red[2][t] + red[3][t];
    a = rbf(a);                                        // bf16 Linear output
    if (act) a = rbf(apply_act(act, a));
    if (GATED) a = rbf(a * rbf(red[0][R * FB + t] + red[1][R * FB + t] + red[2][R * FB + t] + red[3][R * FB + t]));
    if (has_res) a = rbf(bf2f((bf16_t)q.res) + a);
    stc(out + (long)eb * ldo + row0 + er, f2bf(a));
  }
}

// activation rows of the operation (sc1: written by other workgroups in the previous operation), optional fused RMSNorm
template <int J, bool NORM>
__device__ __forceinline__ void load_x(u32x4_t (&xq)[FB][J], const bf16_t* x, int ldx, int K, int t, int lane, int wave,
                                       const float* norm_w, float eps, float (*ssq)[FB]) {
#pragma unroll
  for (int b = 0; b < FB; ++b) {
    const __amdgpu_buffer_rsrc_t rx = rsrc_of(x + (long)b * ldx, K * 2);
#pragma unroll
    for (int j = 0; j < J; ++j) xq[b][j] = __builtin_amdgcn_raw_buffer_load_b128(rx, (j * 256 + t) * 16, 0, AUX_SC1);
  }
  if (NORM) {
    f32x4_t g0[J], g1[J];
    const __amdgpu_buffer_rsrc_t rg = rsrc_of(norm_w, K * 4);
#pragma unroll
    for (int j = 0; j < J; ++j) {
      g0[j] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, (j * 256 + t) * 32, 0, 0));
      g1[j] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, (j * 256 + t) * 32 + 16, 0, 0));
    }
#pragma unroll
    for (int b = 0; b < FB; ++b) {
      const float ss = wave_sum(chunks_sumsq<J>(xq[b]));
      if (lane == 0) ssq[wave][b] = ss;
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < FB; ++b) {
      const float r1 = rsqrtf((ssq[0][b] + ssq[1][b] + ssq[2][b] + ssq[3][b]) / (float)K + eps);
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const u32x4_t a = xq[b][j];
        xq[b][j] = (u32x4_t){pack_bf16x2(bf_lo(a[0]) * r1 * g0[j][0], bf_hi(a[0]) * r1 * g0[j][1]),
                             pack_bf16x2(bf_lo(a[1]) * r1 * g0[j][2], bf_hi(a[1]) * r1 * g0[j][3]),
                             pack_bf16x2(bf_lo(a[2]) * r1 * g1[j][0], bf_hi(a[2]) * r1 * g1[j][1]),
                             pack_bf16x2(bf_lo(a[3]) * r1 * g1[j][2], bf_hi(a[3]) * r1 * g1[j][3])};
      }
    }
  }
}

// One GEMV operation over this sub-group's rows [lo, hi): nb batches of R rows (nb uniform over the grid).  On entry b0 / b1
// hold batches 0 / 1 (requested during the previous operation / its barrier); a consumed buffer is re-requested at once (one
// batch in flight under a batch's arithmetic, two while waiting); before the last batch is consumed `pre_last()` requests the
// first batch of the NEXT operation.
template <int R, int J, bool GATED, class PreLast>
__device__ __forceinline__ void gemv_op(WB<R, J, GATED>& b0, WB<R, J, GATED>& b1, const u32x4_t (&xq)[FB][J], const bf16_t* W,
                                        const bf16_t* W2, int K, int lo, int hi, int nb, int act, const bf16_t* resid, int ldr,
                                        bf16_t* out, int ldo, float (*red)[4][VMAX], int& par, int t, int lane, int wave,
                                        PreLast pre_last) {
  int i = 0;
#pragma unroll 1
  while (true) {
    {
      if (i == nb - 1) pre_last();
      wb_consume(b0, xq, red[par], lo + i * R, hi, t, lane, wave, act, resid != nullptr, out, ldo);
      par ^= 1;
      if (++i == nb) break;
      if (i + 1 < nb) wb_request(b0, W, W2, K, lo + (i + 1) * R, hi, t, resid, ldr);
    }
    {
      if (i == nb - 1) pre_last();
      wb_consume(b1, xq, red[par], lo + i * R, hi, t, lane, wave, act, resid != nullptr, out, ldo);
      par ^= 1;
      if (++i == nb) break;
      if (i + 1 < nb) wb_request(b1, W, W2, K, lo + (i + 1) * R, hi, t, resid, ldr);
    }
  }
}

// this thread's stores of the operation have reached the coherence point (and every earlier load has landed)
__device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// arrive + wait; nbar = barriers passed since the control block was zeroed, this one included
__device__ __forceinline__ bool grid_sync(unsigned* ctl, unsigned nbar, int G, int tid, int* flag) {
  __syncthreads();
  if (tid < 64) {
    if (tid == 0) __hip_atomic_fetch_add(ctl + CTR0 + (blockIdx.x & (NCTR - 1)) * CSTRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned mine = tid < NCTR ? nbar * (unsigned)((G - tid + NCTR - 1) / NCTR) : 0u;     // counter c: workgroups b = c (mod 8)
    unsigned spins = 0;
    bool ok = true;
    while (true) {
      const unsigned v = tid < NCTR ? __hip_atomic_load(ctl + CTR0 + tid * CSTRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
      if (__builtin_amdgcn_ballot_w64(v < mine) == 0) break;
      __builtin_amdgcn_s_sleep(1);
      if (++spins > SPIN_LIMIT) { ok = false; break; }
    }
    if (tid == 0) {
      *flag = ok;
      if (!ok) __hip_atomic_store(ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  return *flag != 0;
}

// -DLD_FUSED_TRACE (tools/build_variant.sh): workgroup 0 records the 100 MHz wall clock at the phase boundaries of layer 1 into
// ctl[288 ...] (tools/llm_fused_trace.py prints the intervals)
#ifdef LD_FUSED_TRACE
#define TR(slot) do { if (blockIdx.x == LD_FUSED_TRACE && tid == 0 && layer == 1) ((unsigned long long*)(p.ctl + 288))[slot] = wall_clock64(); } while (0)
// every workgroup: arrival (even slot) / exit (odd slot) times of the six barriers of layer 1 into ctl[512 ...] (the trace tool
// allocates the larger control block)
#define TRB(k) do { if (tid == 0 && layer == 1) ((unsigned long long*)(p.ctl + 512))[blockIdx.x * 12 + (k)] = wall_clock64(); } while (0)
#else
#define TR(slot) do { } while (0)
#define TRB(k) do { } while (0)
#endif

// rows per batch of the four GEMV operations (K = hidden: one chunk per thread; K = mlp: J2 chunks per thread)
constexpr int R_QKV = 8, R_WO = 8, R_W13 = 4, R_W2 = 2;

template <int J2>
__global__ __launch_bounds__(NT) void ld_llm_blocks_fused_kernel(FusedParams p) {
  __shared__ float red[SGN][2][4][VMAX];
  __shared__ float ssq[SGN][4][FB];
  __shared__ float att_lds[SGN][8 + 4 * 128];
  __shared__ int flag;
  __shared__ unsigned epoch_s;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sg = wv >> 2, wave = wv & 3, t = tid & 255;
  const int sub = lane & 15, kq = lane >> 4;
  const int G = gridDim.x, NSG = G * SGN;
  const int gs = sg * G + blockIdx.x;                  // global sub-group id (neighbouring ids on different CUs)
  const int hidden = p.hidden, mlp = p.mlp, H = p.heads;

  if (tid == 0) {
    epoch_s = p.ctl[0];
    flag = p.ctl[1] == 0;
  }
  __syncthreads();
  if (!flag) return;                                   // an earlier launch gave up on a barrier: do nothing
  const unsigned nb_launch = (unsigned)(p.n_layers * BARRIERS_PER_LAYER - 1);
  unsigned nbar = epoch_s * nb_launch;

  // rows of the four operations owned by this sub-group, and the (uniform) batch counts
  auto share = [&](int N, int& lo, int& hi) { lo = (int)((long)N * gs / NSG); hi = (int)((long)N * (gs + 1) / NSG); };
  int lo_qkv, hi_qkv, lo_wo, hi_wo, lo_13, hi_13, lo_2, hi_2;
  share(3 * hidden, lo_qkv, hi_qkv); share(hidden, lo_wo, hi_wo); share(mlp, lo_13, hi_13); share(hidden, lo_2, hi_2);
  auto nbatch = [&](int N, int R) { const int per = (N + NSG - 1) / NSG; return (per + R - 1) / R; };
  const int nb_qkv = nbatch(3 * hidden, R_QKV), nb_wo = nbatch(hidden, R_WO), nb_13 = nbatch(mlp, R_W13), nb_2 = nbatch(hidden, R_W2);

  // attention work items: (batch row, head, key split); item gs (+ NSG per round)
  const int L = *p.pos + 1, pos = L - 1;
  const int nsplit = p.nsplit, ns = kv_eff_splits(L, nsplit, p.rule), chunk = (L + ns - 1) / ns;
  const int n_items = FB * H * ns, rounds = (n_items + NSG - 1) / NSG;      // partial results keep the [bh][nsplit] layout

  WB<R_QKV, 1, false> q0, q1;
  WB<R_WO, 1, false> o0, o1;
  WB<R_W13, 1, true> m0, m1;
  WB<R_W2, J2, false> d0, d1;
  KvRows kv;
  int par = 0;

  auto item_geom = [&](int item, int& b, int& h, int& k_begin, int& n) {
    const int bh = item / ns, sp = item - bh * ns;
    b = bh / H; h = bh - b * H;
    k_begin = sp * chunk;
    n = max(0, min(L, k_begin + chunk) - k_begin);
  };
  auto kv_request_item = [&](int item, const ld_llm_layer& w) {
    int b, h, k_begin, n;
    item_geom(item, b, h, k_begin, n);
    if (item >= n_items) n = 0;
    kv_rows_request(kv, (const bf16_t*)w.k_cache, (const bf16_t*)w.v_cache, (long)b * p.Lmax + k_begin, H, h, n, wave, kq, sub);
  };

  {
    const ld_llm_layer& w = p.layers[0];
    wb_request(q0, (const bf16_t*)w.wqkv, nullptr, hidden, lo_qkv, hi_qkv, t, nullptr, 0);
    wb_request(q1, (const bf16_t*)w.wqkv, nullptr, hidden, lo_qkv + R_QKV, hi_qkv, t, nullptr, 0);
  }

#pragma unroll 1
  for (int layer = 0; layer < p.n_layers; ++layer) {
    const ld_llm_layer w = p.layers[layer];
    const bool last_layer = layer + 1 == p.n_layers;

    // ---- qkv = Wqkv . rmsnorm(x) ----
    TR(0);
    {
      u32x4_t xq[FB][1];
      load_x<1, true>(xq, p.x, hidden, hidden, t, lane, wave, w.n0, p.rms_eps, ssq[sg]);
      TR(1);
      gemv_op(q0, q1, xq, (const bf16_t*)w.wqkv, nullptr, hidden, lo_qkv, hi_qkv, nb_qkv, 0, nullptr, 0, p.qkv, 3 * hidden,
              red[sg], par, t, lane, wave, [&]() { kv_request_item(gs, w); });
    }
    TR(2);
    drain();
    TR(3);
    wb_request(o0, (const bf16_t*)w.wo, nullptr, hidden, lo_wo, hi_wo, t, p.x, hidden);
    TRB(0); if (!grid_sync(p.ctl, ++nbar, G, tid, &flag)) return; TRB(1);
    TR(4);

    // ---- split attention with fused RoPE + KV append (m0: first w1.w3 batch requested under it) ----
    wb_request(m0, (const bf16_t*)w.w1, (const bf16_t*)w.w3, hidden, lo_13, hi_13, t, nullptr, 0);
#pragma unroll 1
    for (int rd = 0; rd < rounds; ++rd) {
      const int item = gs + rd * NSG;
      const bool active = item < n_items;
      int b, h, k_begin, n;
      item_geom(active ? item : 0, b, h, k_begin, n);
      if (!active) n = 0;
      if (rd > 0) kv_request_item(item, w);
      const bf16_t* src = p.qkv + ((long)b * 3 * H + h) * 128;       // [B][3][H][128]: q at +0, k at +H*128, v at +2*H*128
      const __amdgpu_buffer_rsrc_t rq = rsrc_of(src, (2 * H + 1) * 128 * 2);
      const u32x4_t a_q = __builtin_amdgcn_raw_buffer_load_b128(rq, sub * 16, 0, AUX_SC1);
      const u32x4_t a_k = __builtin_amdgcn_raw_buffer_load_b128(rq, (H * 128 + sub * 8) * 2, 0, AUX_SC1);
      const u32x4_t a_v = __builtin_amdgcn_raw_buffer_load_b128(rq, (2 * H * 128 + sub * 8) * 2, 0, AUX_SC1);
      float cs[4], sn[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { cs[e] = p.cos_t[pos * 64 + sub * 4 + e]; sn[e] = p.sin_t[pos * 64 + sub * 4 + e]; }
      const int item_a = active ? item : 0;
      float* out_ws = p.ws + ((long)(item_a / ns) * nsplit + (item_a % ns)) * 130;
      kv_attn_split_core(kv, a_q, a_k, a_v, cs, sn, true, (bf16_t*)w.k_cache, (bf16_t*)w.v_cache, (long)b * p.Lmax + pos, H, h,
                         pos - k_begin, n, active, out_ws, att_lds[sg], att_lds[sg] + 8, t, lane, wave,
                         [](float* q, float v) { stc(q, v); });
    }
    TR(5);
    drain();
    TR(6);
    wb_request(m1, (const bf16_t*)w.w1, (const bf16_t*)w.w3, hidden, lo_13 + R_W13, hi_13, t, nullptr, 0);
    TRB(2); if (!grid_sync(p.ctl, ++nbar, G, tid, &flag)) return; TRB(3);
    TR(7);

    // ---- combine the key splits: one (batch row, head) per sub-group ----
    for (int bh = gs; bh < FB * H; bh += NSG) {
      if (t < 128) {
        const float r = kv_attn_combine_core(p.ws + (long)bh * nsplit * 130, ns, t, [](const float* q) { return ldc(q); });
        stc(p.att + (long)bh * 128 + t, f2bf(r));
      }
    }
    TR(8);
    drain();
    TR(9);
    wb_request(o1, (const bf16_t*)w.wo, nullptr, hidden, lo_wo + R_WO, hi_wo, t, p.x, hidden);
    TRB(4); if (!grid_sync(p.ctl, ++nbar, G, tid, &flag)) return; TRB(5);
    TR(10);

    // ---- x += Wo . att ----
    {
      u32x4_t xq[FB][1];
      load_x<1, false>(xq, p.att, hidden, hidden, t, lane, wave, nullptr, 0.f, ssq[sg]);
      TR(11);
      gemv_op(o0, o1, xq, (const bf16_t*)w.wo, nullptr, hidden, lo_wo, hi_wo, nb_wo, 0, p.x, hidden, p.x, hidden,
              red[sg], par, t, lane, wave, [&]() {});
    }
    TR(12);
    drain();
    TR(13);
    TRB(6); if (!grid_sync(p.ctl, ++nbar, G, tid, &flag)) return; TRB(7);
    TR(14);

    // ---- gate = gelu(W1 . rmsnorm(x)) * (W3 . rmsnorm(x)) ----
    {
      u32x4_t xq[FB][1];
      load_x<1, true>(xq, p.x, hidden, hidden, t, lane, wave, w.n1, p.rms_eps, ssq[sg]);
      TR(15);
      gemv_op(m0, m1, xq, (const bf16_t*)w.w1, (const bf16_t*)w.w3, hidden, lo_13, hi_13, nb_13, LD_ACT_GELU_TANH, nullptr, 0,
              p.gate, mlp, red[sg], par, t, lane, wave,
              [&]() { wb_request(d0, (const bf16_t*)w.w2, nullptr, mlp, lo_2, hi_2, t, p.x, hidden); });
    }
    TR(16);
    drain();
    TR(17);
    wb_request(d1, (const bf16_t*)w.w2, nullptr, mlp, lo_2 + R_W2, hi_2, t, p.x, hidden);
    TRB(8); if (!grid_sync(p.ctl, ++nbar, G, tid, &flag)) return; TRB(9);
    TR(18);

    // ---- x += W2 . gate ----
    {
      u32x4_t xq[FB][J2];
      load_x<J2, false>(xq, p.gate, mlp, mlp, t, lane, wave, nullptr, 0.f, ssq[sg]);
      TR(19);
      const ld_llm_layer* nx = last_layer ? nullptr : p.layers + layer + 1;
      gemv_op(d0, d1, xq, (const bf16_t*)w.w2, nullptr, mlp, lo_2, hi_2, nb_2, 0, p.x, hidden, p.x, hidden,
              red[sg], par, t, lane, wave,
              [&]() { if (nx) wb_request(q0, (const bf16_t*)nx->wqkv, nullptr, hidden, lo_qkv, hi_qkv, t, nullptr, 0); });
      if (last_layer) break;
      TR(20);
      drain();
      TR(21);
      wb_request(q1, (const bf16_t*)nx->wqkv, nullptr, hidden, lo_qkv + R_QKV, hi_qkv, t, nullptr, 0);
    }
    TRB(10); if (!grid_sync(p.ctl, ++nbar, G, tid, &flag)) return; TRB(11);
    TR(22);
  }
  if (blockIdx.x == 0 && tid == 0) p.ctl[0] = epoch_s + 1;
}

}  // namespace

LD_API int ld_llm_decode_blocks_fused(const ld_llm_layer* layers_dev, int64_t n_layers, const int32_t* pos, void* x, void* qkv,
                                      void* att, void* gate, float* attn_ws, const float* cos_t, const float* sin_t, int64_t B,
                                      int64_t hidden, int64_t heads, int64_t mlp, int64_t Lmax, int64_t nsplit, float rms_eps,
                                      uint32_t* ctl, void* stream) {
  LD_REQUIRE(layers_dev && n_layers > 0 && pos && x && qkv && att && gate && attn_ws && cos_t && sin_t && ctl,
             "ld_llm_decode_blocks_fused: null pointer");
  if (B != FB || hidden != heads * 128 || hidden % 8 || hidden > 2048 || mlp % 8 || mlp > 6 * 2048 || nsplit < 2 ||
      (Lmax + nsplit - 1) / nsplit > 16 * KV_MAXIT)
    return ld_set_error(LD_ERR_UNSUPPORTED, "ld_llm_decode_blocks_fused: B=%ld hidden=%ld heads=%ld mlp=%ld Lmax=%ld nsplit=%ld outside "
                        "the fused form (B = 2, head_dim 128, hidden <= 2048, mlp <= 12288, <= 256 keys per split)",
                        (long)B, (long)hidden, (long)heads, (long)mlp, (long)Lmax, (long)nsplit);
  // every workgroup must be resident at once: one per CU, fewer if the device cannot hold that
  static thread_local int grid_cache[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return ld_set_error(LD_ERR_LAUNCH, "ld_llm_decode_blocks_fused: hipGetDevice failed");
  int grid = (dev >= 0 && dev < 16) ? grid_cache[dev] : 0;
  if (grid == 0) {
    hipDeviceProp_t prop;
    int per_cu = 0;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ld_llm_blocks_fused_kernel<6>, NT, 0) != hipSuccess || per_cu < 1)
      return ld_set_error(LD_ERR_LAUNCH, "ld_llm_decode_blocks_fused: no resident workgroup per CU");
    grid = prop.multiProcessorCount;
    if (const char* e = getenv("LD_LLM_FUSED_WGS")) { const int g = atoi(e); if (g > 0 && g < grid) grid = g; }
    if (dev >= 0 && dev < 16) grid_cache[dev] = grid;
  }
  FusedParams p{};
  p.layers = layers_dev; p.n_layers = (int)n_layers;
  p.x = (bf16_t*)x; p.qkv = (bf16_t*)qkv; p.att = (bf16_t*)att; p.gate = (bf16_t*)gate; p.ws = attn_ws;
  p.cos_t = cos_t; p.sin_t = sin_t; p.pos = (const int*)pos;
  p.hidden = (int)hidden; p.heads = (int)heads; p.mlp = (int)mlp; p.Lmax = (int)Lmax; p.nsplit = (int)nsplit; p.rms_eps = rms_eps; p.rule = ld_kv_split_rule();
  p.ctl = (unsigned*)ctl;
  hipLaunchKernelGGL(ld_llm_blocks_fused_kernel<6>, dim3((unsigned)grid), dim3(NT), 0, (hipStream_t)stream, p);
  return ld_check_launch("ld_llm_decode_blocks_fused");
}

LD_API int ld_llm_decode_forward_fused(const ld_llm_layer* layers_dev, int64_t n_layers, const float* emb_table,
                                       const int64_t* token, const int32_t* pos, void* x, void* qkv, void* att, void* gate,
                                       float* attn_ws, const float* cos_t, const float* sin_t, const float* lnf_w,
                                       const float* lnf_b, float* lnf_out, const float* head_w, float* logits, int64_t B,
                                       int64_t hidden, int64_t heads, int64_t mlp, int64_t vocab, int64_t Lmax, int64_t nsplit,
                                       float rms_eps, float ln_eps, uint32_t* ctl, void* stream) {
  LD_REQUIRE((emb_table == nullptr || token) && lnf_w && lnf_b && lnf_out && head_w && logits, "ld_llm_decode_forward_fused: null pointer");
  int rc = emb_table ? ld_llm_embed(emb_table, token, x, B, hidden, stream) : 0;
  if (rc) return rc;
  rc = ld_llm_decode_blocks_fused(layers_dev, n_layers, pos, x, qkv, att, gate, attn_ws, cos_t, sin_t, B, hidden, heads, mlp, Lmax,
                                  nsplit, rms_eps, ctl, stream);
  if (rc) return rc;
  rc = ld_layernorm_bf16_to_f32(x, hidden, lnf_w, lnf_b, lnf_out, B, hidden, ln_eps, stream);
  if (rc) return rc;
  return ld_gemv(lnf_out, hidden, 1, head_w, nullptr, 1, nullptr, nullptr, 0, logits, vocab, 1, B, vocab, hidden, 0, 0, nullptr, 0.f,
                 stream);
}

// Device helpers shared by the decode-step kernels (ld_llm.hip: one launch per operation; ld_llm_fused.hip: one persistent
// launch for all blocks of a step).  Both forms must produce the same bits, so the arithmetic they share lives here.
#pragma once
#include "ld_common.h"

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

__device__ __forceinline__ float dot2_bf16(uint32_t a, uint32_t b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}

// Sum V (power of two, <= 64) per-lane values over the 64 lanes; afterwards lane L holds the total of value
// L >> (6 - log2 V) (every lane of that group holds the same number).
template <int V, int N = V, int O = 32>
__device__ __forceinline__ void wave_sum_multi(float (&v)[V], int lane) {
  if constexpr (N > 1) {
    const bool up = (lane & O) != 0;
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
      const float keep = up ? v[i + N / 2] : v[i];
      const float send = up ? v[i] : v[i + N / 2];
      v[i] = keep + __shfl_xor(send, O, 64);
    }
    wave_sum_multi<V, N / 2, O / 2>(v, lane);
  } else if constexpr (O > 0) {
    v[0] += __shfl_xor(v[0], O, 64);
    wave_sum_multi<V, 1, O / 2>(v, lane);
  }
}
constexpr int ceil_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }
constexpr int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// sum of squares of the 8 * J bf16 values a thread holds of one activation row (RMSNorm partial)
template <int J>
__device__ __forceinline__ float chunks_sumsq(const u32x4_t (&x)[J]) {
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < J; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float lo = bf_lo(x[j][e]), hi = bf_hi(x[j][e]); ss += lo * lo + hi * hi; }
  return ss;
}

// ---------------------------------------------------------------------------------------------
// Decode-step (m = 1) KV attention of one (batch row, head, key split) by 256 threads: thread (wave, kq = lane >> 4,
// sub = lane & 15) holds 8 of the 128 dims of the keys wave * 4 + it * 16 + kq of the split.  The caller has requested the
// K / V rows (kv_rows_request) and the new token's q / k / v chunks; this is everything after the loads.
// ---------------------------------------------------------------------------------------------
constexpr int KV_MAXIT = 16;     // trips of 16 keys per 256 threads: a split covers at most 256 keys
struct KvRows { u32x4_t k[KV_MAXIT], v[KV_MAXIT]; };

// How many of the launch's `nsplit_max` key splits a decode step at context length L really uses: a short context is not worth a
// merge (one split: the workgroup writes the attention output itself), a medium one not worth eight arrivals.  A pure function of
// L, evaluated on the device by every form of the decode step (per-operation launches, the chained and the persistent form, a
// captured graph), so that all of them keep producing the same bits; workgroups of unused splits leave at once.
// t1 / t2 / t4: longest context served by 1 / 2 / 4 splits (LD_KV_SPLIT_T=t1,t2,t4; LD_KV_SPLIT_T=0,0,0: always nsplit_max).
struct KvSplitRule { int t1, t2, t4; };
// defaults from tools/kv_attn_sweep.py (profiles/r05_llm_kv_attn_split_sweep.txt): one split is fastest up to ~110 keys (no merge:
// -1.5 us), two are never the best, four win up to ~230 keys, beyond that the full eight (more would not fit one round of the chip)
constexpr int KV_SPLIT_T1 = 112, KV_SPLIT_T2 = 112, KV_SPLIT_T4 = 232;
__host__ __device__ inline int kv_eff_splits(int L, int nsplit_max, KvSplitRule r) {
  int ns = L <= r.t1 ? 1 : L <= r.t2 ? 2 : L <= r.t4 ? 4 : nsplit_max;
  if (ns > nsplit_max) ns = nsplit_max;
  while ((L + ns - 1) / ns > 16 * KV_MAXIT && ns < nsplit_max) ns = 2 * ns < nsplit_max ? 2 * ns : nsplit_max;
  return ns;
}
KvSplitRule ld_kv_split_rule();      // ld_llm.hip (environment knob, read once)

// rows k_begin .. k_begin + n of (b, h) in caches laid out [B][Lmax][H][128]; row0 = b * Lmax + k_begin
__device__ __forceinline__ void kv_rows_request(KvRows& r, const bf16_t* kc, const bf16_t* vc, long row0, int H, int h, int n,
                                                int wave, int kq, int sub) {
#pragma unroll
  for (int it = 0; it < KV_MAXIT; ++it) {
    const int kk = wave * 4 + it * 16 + kq;
    r.k[it] = (u32x4_t){0u, 0u, 0u, 0u}; r.v[it] = (u32x4_t){0u, 0u, 0u, 0u};
    if (it * 16 >= n) continue;
    if (kk < n) {
      const long off = ((row0 + kk) * H + h) * 128 + sub * 8;
      r.k[it] = __builtin_nontemporal_load((const u32x4_t*)(kc + off));
      r.v[it] = __builtin_nontemporal_load((const u32x4_t*)(vc + off));
    }
  }
}

// direct_out != nullptr: this split is the whole context -- the normalised bf16 output row is written there, no partial result.
// Store: functor (float* p, float v) -- plain or agent-coherent.  red: 8 floats, part: 4 * 128 floats of LDS owned by these
// 256 threads; contains three workgroup barriers (every thread of the workgroup must get here, active or not).
// rope: a_q / a_k / a_v are the new token's raw chunks, rotated here (apply_rope, pos_emb.py:16-46); the split that holds
// position pos takes the new key / value from registers and appends them to the cache for the following steps.
template <class Store>
__device__ __forceinline__ void kv_attn_split_core(KvRows& r, u32x4_t a_q, u32x4_t a_k, u32x4_t a_v, const float (&cs)[4],
                                                   const float (&sn)[4], bool rope, bf16_t* kc, bf16_t* vc, long cache_row,
                                                   int H, int h, int pk, int n, bool active, float* out_ws, float* red,
                                                   float* part, int t, int lane, int wave, Store store, bf16_t* direct_out = nullptr) {
  constexpr int D = 128;
  const int sub = lane & 15, kq = lane >> 4;
  float qreg[8];
  if (rope) {
    u32x4_t knew;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float qa = bf_lo(a_q[e]), qb = bf_hi(a_q[e]);
      qreg[2 * e] = rbf(qa * cs[e] - qb * sn[e]);
      qreg[2 * e + 1] = rbf(qa * sn[e] + qb * cs[e]);
      const float ka = bf_lo(a_k[e]), kb = bf_hi(a_k[e]);
      knew[e] = pack_bf16x2(ka * cs[e] - kb * sn[e], ka * sn[e] + kb * cs[e]);
    }
    if (active && pk >= 0 && pk < n) {                           // slot of the new key inside this split, if it is here
      if (wave == 0 && kq == 0) {                                // append for the following steps
        const long co = (cache_row * H + h) * D + sub * 8;
        *(u32x4_t*)(kc + co) = knew;
        *(u32x4_t*)(vc + co) = a_v;
      }
#pragma unroll
      for (int it = 0; it < KV_MAXIT; ++it) {
        if (wave * 4 + it * 16 + kq == pk) { r.k[it] = knew; r.v[it] = a_v; }
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) { qreg[2 * e] = bf_lo(a_q[e]); qreg[2 * e + 1] = bf_hi(a_q[e]); }
  }
  const float inv_sqrt_d = 0.08838834764831845f;
  float lmax = -3.0e38f;
  float sreg[KV_MAXIT];
#pragma unroll
  for (int it = 0; it < KV_MAXIT; ++it) {
    const int kk = wave * 4 + it * 16 + kq;
    sreg[it] = -3.0e38f;
    if (it * 16 >= n) continue;                       // uniform over the 256 threads: trips past this split's keys cost nothing
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { d = fmaf(qreg[2 * e], bf_lo(r.k[it][e]), d); d = fmaf(qreg[2 * e + 1], bf_hi(r.k[it][e]), d); }
    d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64); d += __shfl_xor(d, 8, 64);
    if (kk < n) {
      sreg[it] = rbf(rbf(d) * inv_sqrt_d);
      lmax = fmaxf(lmax, sreg[it]);
    }
  }
  lmax = wave_max(lmax);
  if (lane == 0) red[wave] = lmax;
  __syncthreads();
  const float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  float lsum = 0.f;
#pragma unroll
  for (int it = 0; it < KV_MAXIT; ++it) {
    const int kk = wave * 4 + it * 16 + kq;
    if (it * 16 >= n) continue;
    if (kk < n) {
      const float pk_ = __expf(sreg[it] - mx);
      if (sub == 0) lsum += pk_;
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[2 * e] = fmaf(pk_, bf_lo(r.v[it][e]), acc[2 * e]); acc[2 * e + 1] = fmaf(pk_, bf_hi(r.v[it][e]), acc[2 * e + 1]); }
    }
  }
  // reduce over the 4 key groups of a wave (lanes sub, sub+16, sub+32, sub+48), then over waves via LDS
#pragma unroll
  for (int e = 0; e < 8; ++e) { acc[e] += __shfl_xor(acc[e], 16, 64); acc[e] += __shfl_xor(acc[e], 32, 64); }
  lsum = wave_sum(lsum);
  __syncthreads();
  if (kq == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) part[wave * D + sub * 8 + e] = acc[e];
  }
  if (lane == 0) red[4 + wave] = lsum;
  __syncthreads();
  if (active) {
    if (direct_out) {            // the only split of its (batch row, head): what the merge of one partial result would give, o / l
      if (t < D) direct_out[t] = f2bf((part[t] + part[D + t] + part[2 * D + t] + part[3 * D + t]) / (red[4] + red[5] + red[6] + red[7]));
    } else {
      if (t < D) store(out_ws + 2 + t, part[t] + part[D + t] + part[2 * D + t] + part[3 * D + t]);
      if (t == 0) { store(out_ws, mx); store(out_ws + 1, red[4] + red[5] + red[6] + red[7]); }
    }
  }
}

// merge the nsplit partial results of one (batch row, head): thread d of 128.  Eight splits' loads are issued together (the
// loads may be device-coherent ones that travel to memory: one exposed latency per eight splits, not three per split).
template <class Load>
__device__ __forceinline__ float kv_attn_combine_core(const float* w, int nsplit, int d, Load load) {
  constexpr int D = 128, U = 8;
  float mx = -3.0e38f;
  for (int s0 = 0; s0 < nsplit; s0 += U) {
    float m[U];
#pragma unroll
    for (int i = 0; i < U; ++i) m[i] = (s0 + i < nsplit) ? load(w + (s0 + i) * (D + 2)) : -3.0e38f;
#pragma unroll
    for (int i = 0; i < U; ++i) mx = fmaxf(mx, m[i]);
  }
  float l = 0.f, o = 0.f;
  for (int s0 = 0; s0 < nsplit; s0 += U) {
    float m[U], ls[U], os[U];
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const bool live = s0 + i < nsplit;
      const float* ws = w + (s0 + i) * (D + 2);
      m[i] = live ? load(ws) : 0.f; ls[i] = live ? load(ws + 1) : 0.f; os[i] = live ? load(ws + 2 + d) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < U; ++i) {
      if (s0 + i < nsplit) {
        const float f = __expf(m[i] - mx);
        l += f * ls[i];
        o += f * os[i];
      }
    }
  }
  return o / l;
}

// ---------------------------------------------------------------------------------------------
// Dependent launches on two streams (ld_llm_decode_blocks_chained): operation k of a decode step is launched on stream k & 1
// with NO stream dependency on operation k - 1; instead its workgroups request their first weight rows, then wait until every
// workgroup of operation k - 1 has arrived on that operation's counters (its outputs written sc1 and drained), and arrive on
// their own counters when done.  A launch is therefore dispatched, resident and streaming weights while its predecessor is
// still running.  ctl: word 0 = error flag; slot s, counter c at word 64 + (s * 8 + c) * 16; workgroup w arrives on counter
// w & 7; counters are never reset inside a decode: the target of epoch e (1-based step count) is e * (workgroups on it).
// ---------------------------------------------------------------------------------------------
struct ChainSync {
  unsigned* ctl;         // null: plain launch
  int slot;              // this operation's counter slot
  int prev_grid;         // workgroups of the operation waited for (0: none)
  unsigned epoch1;       // 1-based step count since the control block was zeroed
};
constexpr int CHAIN_CTR0 = 64, CHAIN_CSTRIDE = 16, CHAIN_NCTR = 8;
constexpr unsigned CHAIN_SPIN_LIMIT = 1u << 20;

// every thread of the workgroup calls it; one workgroup barrier inside
__device__ __forceinline__ void chain_wait(const ChainSync& cs, int tid) {
  if (cs.prev_grid > 0 && tid < 64) {
    const unsigned* ctr = cs.ctl + CHAIN_CTR0 + (cs.slot - 1) * CHAIN_NCTR * CHAIN_CSTRIDE;
    const unsigned mine = tid < CHAIN_NCTR ? cs.epoch1 * (unsigned)((cs.prev_grid - tid + CHAIN_NCTR - 1) / CHAIN_NCTR) : 0u;
    unsigned spins = 0;
    while (true) {
      const unsigned v = tid < CHAIN_NCTR ? __hip_atomic_load(ctr + tid * CHAIN_CSTRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
      if (__builtin_amdgcn_ballot_w64(v < mine) == 0) break;
      if ((spins & 63) == 63 && __hip_atomic_load(cs.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;   // somebody gave up: run through
      __builtin_amdgcn_s_sleep(1);
      if (++spins > CHAIN_SPIN_LIMIT) {
        if (tid == 0) __hip_atomic_store(cs.ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  __syncthreads();
}

// after the operation's last (sc1) store of this thread; wg = linear workgroup id
__device__ __forceinline__ void chain_arrive(const ChainSync& cs, int tid, int wg) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0)
    __hip_atomic_fetch_add(cs.ctl + CHAIN_CTR0 + (cs.slot * CHAIN_NCTR + (wg & (CHAIN_NCTR - 1))) * CHAIN_CSTRIDE, 1u, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}

// Unmasked DiT attention (joint text+video, head_dim 64) on v_mfma_f32_16x16x32_bf16.  Same reference op as ld_attn.hip /
// ld_attn_pipe.hip (sat attention_fn_default -> F.scaled_dot_product_attention, landiff/diffusion/dit_video_concat.py:636-664),
// same pipeline as ld_attn_pipe.hip (four K and four V^T slots of 8 KB fed by buffer_load ... lds, one workgroup barrier per
// two tiles, QK^T of tile j+1 issued over the softmax of tile j, max-free fast pass with an in-kernel safe fallback); what
// changes is the matrix instruction.
//
// Why: these kernels are bound by the chip's power / current governor, not by issue slots (removing 7 VALU instructions
// and a branch per tile from the 32x32x16 kernel changed nothing; the same stream runs 31 % faster on zero operands), and an
// MFMA-only loop on random operands sustains 2105 TFLOP/s on 16x16x32 against 1837 on 32x32x16 (tools/probe/mfma_power.hip,
// profiles/r02_mfma_power_probe.txt): the 16x16x32 form costs less energy per FLOP.  Own translation unit, built with
// -fno-slp-vectorize like ld_attn_pipe.hip.
//
// Register-level dataflow of one wave (32 query rows = two 16-row blocks qb, one 64-key tile = four 16-key blocks kb):
//   S^T[kb][qb] (16 keys x 16 q) += K[kb] (16 keys x 32 d, A operand) . Q^T[qb] (32 d x 16 q, B operand)      2 k-steps over d
//       C layout: lane l holds q = qb*16 + (l & 15), S^T rows 4*(l >> 4) + r, r = 0..3
//   O^T[db][qb] (16 d x 16 q)  += V^T[db] (16 d x 32 keys, A) . P[kg][qb] (32 keys x 16 q, B)                 2 key groups kg
//       B layout: lane l holds q = qb*16 + (l & 15), keys kg*32 + 8*(l >> 4) + e, e = 0..7
// The accumulator registers of S^T become the B fragment of the PV product without any cross-lane traffic when row
// rho = 4*h + r of block kb = 2*kg + b holds the key kg*32 + 8*h + 4*b + r: the K fragment reads simply permute the rows they
// fetch (bits [b][h][r] of the row index -> key bits [h][b][r]).  Row sums stay per lane (four lanes share a query row)
// and are combined once at the end.
//
// LDS images (lane-linear LDS-DMA destination, swizzle on the source address and on the read): K tile [64 keys][64 d],
// V^T tile [64 d][64 keys], 128-byte rows, 16-byte chunk index XOR f(row).  V^T fragments read rows db*16 + (l & 15) with
// chunk kg*4 + (l >> 4): the usual f = (row >> 1) & 7 is conflict-free for the four 16-lane groups of a ds_read_b128.  The
// permuted K rows are not, with that f; f_K(row) = bit 1 | bits 4:3 << 1 is (each 16-lane group then covers all 16 slots).
#include "ld_attn.h"

namespace {

#define FENCE() __builtin_amdgcn_sched_barrier(0)
// timing experiments only (tools/attn_ablate.sh): bit 0 no exp2, 1 no fragment LDS reads (5: no V^T reads only, 6: no K reads only), 2 no LDS-DMA (7: every DMA reads tile 0, 8: no vmcnt wait at the period end, 9: no barrier there), 3 no P
// packing, 10 / 11 / 12: every other K fragment read / V^T fragment read / DMA piece only (the traffic of a 64-query-row wave), 4 no row-sum MFMAs in the main loop.  Results are wrong when any bit is set; the shipped library is built without the macro.
#ifndef LD_ATTN_ABLATE
#define LD_ATTN_ABLATE 0
#endif

__device__ __forceinline__ int swz_k(int r) { return ((r >> 1) & 1) | (((r >> 3) & 3) << 1); }

// MSUM: softmax denominators from the matrix pipe -- l[qb] += ones(16 x 32) . P[kg][qb], 4 more MFMAs per tile (+12.5 %)
// instead of 32 v_add_f32: the kernel's cycle count follows its VALU-class issue count (SQ counters, profiles/
// r02_attn_p16_pmc_sq.txt), and this takes it from 112 to 84 per tile.  Every lane then holds the complete denominator of
// its two query rows (all 16 result rows of the all-ones product are equal): no cross-lane reduction at the end.  The sum
// runs over the bf16-rounded probabilities, i.e. exactly the weights the PV product uses.  (Measured on one box: 3.99 ms with
// the matrix-pipe sums, 4.10 ms with v_dot2_f32_bf16 on the packed P words, 4.29 ms with v_add_f32; on another, 4.05 ms
// against 4.16 ms for key group 0 on the matrix pipe and key group 1 by v_add_f32: VALU issue slots are the scarcer resource
// even with the matrix pipe 83 % busy.)
template <int NW, bool MSUM>     // NW: waves per workgroup (32 query rows each)
__device__ __forceinline__ void attn_p16_body(const AttnParams& p, int force_safe, char* smem) {   // smem: K slots 0..3 | V^T slots 0..3 | NW flag words
  constexpr int VBASE = 4 * KTILE_BYTES;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, h4 = lane >> 4;
  constexpr int QBW = NW * 32;                    // query rows per workgroup
  constexpr int NPW = 16 / NW;                    // LDS-DMA pieces per wave and tile (K: 8 pieces, V^T: 8 pieces)
  const int nqb = (p.Npad + QBW - 1) / QBW;
  const int n = (p.Nk + KT - 1) / KT;             // >= 6 (launcher)

  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = bid / nqb, qblk = bid - bh * nqb;
  const int b = bh / p.H, h = bh - b * p.H;
  const bf16_t* Qb = p.Q + (long)bh * p.Npad * D;
  const bf16_t* Kb = p.K + (long)bh * p.Npad * D;
  const bf16_t* Vb = p.Vt + (long)bh * D * p.Npad;
  const int q0 = qblk * QBW + wave * 32;
  if (qblk * QBW >= p.Nq) return;

  // Q^T fragments (B operand): rows q0 + qb*16 + l16, d = ks*32 + h4*8 .. + 8, pre-multiplied by scale * log2(e)
  bf16x8_t qf[2][2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int q = q0 + qb * 16 + l16;
    const bf16_t* qrow = Qb + (long)(q < p.Npad ? q : p.Npad - 1) * D + h4 * 8;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const u32x4_t raw = *(const u32x4_t*)(qrow + ks * 32);
      u32x4_t sc;
#pragma unroll
      for (int e = 0; e < 4; ++e) sc[e] = pack_bf16x2(bf_lo(raw[e]) * p.c, bf_hi(raw[e]) * p.c);
      qf[qb][ks] = __builtin_bit_cast(bf16x8_t, sc);
    }
  }

  // LDS-DMA: the first half of the waves brings K tiles (rows = keys), the second half V^T tiles (rows = d)
  const bool kwave = wave < NW / 2;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(kwave ? Kb : Vb), 0, 0x7fffffff, 0x00020000);   // raw buffer, wave-uniform
  const int tstride = kwave ? KT * D * 2 : KT * 2;               // bytes per tile step in the source
  const int rstride = kwave ? D : p.Npad;
  uint32_t goff[NPW];
  int ldsoff[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int piece = (wave % (NW / 2)) * NPW + i;
    const int r = piece * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ (kwave ? swz_k(r) : ((r >> 1) & 7));
    goff[i] = (uint32_t)(r * rstride + chunk * 8) * 2u;
    ldsoff[i] = (kwave ? 0 : VBASE) + piece * 1024;
  }
  auto dma_piece = [&](int i, int slot, int t) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(smem + slot * KTILE_BYTES + ldsoff[i]),
                                             16, goff[i], t * tstride, 0, 0);
  };
  auto dma = [&](int slot, int t) {
#pragma unroll
    for (int i = 0; i < NPW; ++i) dma_piece(i, slot, t);
  };

  // fragment read offsets: K block kb = 2*kg + b, k-step ks: kofs[ks] + kg*4096 + b*512;  V^T block db, key group kg: vofs[kg] + db*2048
  int kofs[2], vofs[2];
  {
    const int key = 8 * (l16 >> 2) + (l16 & 3);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = ks * 4 + h4;
      kofs[ks] = key * 128 + ((c ^ swz_k(key)) << 4);
      vofs[ks] = VBASE + l16 * 128 + ((c ^ ((l16 >> 1) & 7)) << 4);
    }
  }

  f32x4_t o[4][2];
  using T = std::true_type; using F = std::false_type;
  using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>; using S3 = std::integral_constant<int, 3>;
  const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};

  auto mask_tail = [&](f32x4_t (&s)[4][2], int t) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int key0 = t * KT + (kb >> 1) * 32 + h4 * 8 + (kb & 1) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (key0 + r >= p.Nk) { s[kb][0][r] = NEG_BIG; s[kb][1][r] = NEG_BIG; }
    }
  };
  auto load_kf = [&](bf16x8_t (&kf)[4][2], int slot) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        kf[kb][ks] = *(const bf16x8_t*)(smem + kofs[ks] + slot * KTILE_BYTES + (kb >> 1) * 4096 + (kb & 1) * 512);
  };
  auto load_vf = [&](bf16x8_t (&vf)[4][2], int slot) {
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int kg = 0; kg < 2; ++kg)
        vf[db][kg] = *(const bf16x8_t*)(smem + vofs[kg] + slot * KTILE_BYTES + db * 2048);
  };
  auto qk_tile = [&](f32x4_t (&s)[4][2], const bf16x8_t (&kf)[4][2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
          s[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kb][ks], qf[qb][ks], ks == 0 ? zero4 : s[kb][qb], 0, 0, 0);
  };

  // ---------------- fast pass: no running maximum (see ld_attn_pipe.hip for the argument and the window test) ----------------
  // returns the two softmax denominators of this lane's query rows (complete: summed over the four lanes of a row)
  float ltot[2] = {0.f, 0.f};
  auto fast_pass = [&]() {
    float ls[2][2] = {{0.f, 0.f}, {0.f, 0.f}};      // !MSUM: [qb][two partial sums]
    f32x4_t lacc[2] = {zero4, zero4};               // MSUM: row sums from the matrix pipe
    const bf16x8_t ones = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
#pragma unroll
    for (int db = 0; db < 4; ++db) { o[db][0] = zero4; o[db][1] = zero4; }
    f32x4_t sA[4][2], sB[4][2];
    bf16x8_t kf[4][2], vf[4][2];
    // the 32 scores of a lane in a tile are numbered v = kb*8 + qb*4 + r (v < 16: key group 0, v >= 16: key group 1)
    auto EXPV = [&](f32x4_t (&s)[4][2], int v) { const int kb = v >> 3, qb = (v >> 2) & 1, r = v & 3; s[kb][qb][r] = __builtin_amdgcn_exp2f(s[kb][qb][r]); };
    constexpr int NPRE = MSUM ? 18 : 0;             // MSUM: scores 0..NPRE-1 of tile j+1 are exponentiated under the PV MFMAs of tile j

    // One pipelined iteration.  sc = S_j (finished scores; MSUM: the first NPRE already probabilities), sn receives S_{j+1}.
    // HAS_QK: tile j+1 exists; HAS_K2: tile j+2 exists (a main-loop iteration); MASK: tile j+1 is the ragged one.  Even
    // main-loop iterations issue the period's DMA pieces (past the end the tile index is clamped: a re-fetch into a slot nobody
    // reads); odd iterations end the period.
    auto iter = [&](f32x4_t (&sc)[4][2], f32x4_t (&sn)[4][2], int j, auto vslot_c, auto has_qk_c, auto has_k2_c, auto mask_c) {
      constexpr int vslot = decltype(vslot_c)::value;            // slot of V_j; the others follow from it
      constexpr int k2slot = (vslot + 2) & 3;
      constexpr bool HAS_QK = decltype(has_qk_c)::value, HAS_K2 = decltype(has_k2_c)::value, MASK = decltype(mask_c)::value;
      constexpr bool EVEN = (vslot & 1) == 0;
      constexpr bool do_dma = HAS_K2 && EVEN;
      int dt0 = kwave ? j + 4 : j + 2, dt1 = dt0 + 1;
      dt0 = dt0 < n ? dt0 : n - 1; dt1 = dt1 < n ? dt1 : n - 1;
      const int dslot0 = kwave ? vslot : (vslot + 2) & 3, dslot1 = kwave ? (vslot + 1) & 3 : (vslot + 3) & 3;
      u32x4_t pw[2][2];                                          // P fragments [kg][qb]
      // E(v): score -> probability (tile j); N(v): the same on tile j+1 (MSUM); A(v): row-sum add (!MSUM); C(kg, qb, half): two
      // packed words of P fragment [kg][qb]
      auto E = [&](int v) { if (!(LD_ATTN_ABLATE & 1)) EXPV(sc, v); };
      auto N = [&](int v) { if (HAS_QK && !(LD_ATTN_ABLATE & 1)) EXPV(sn, v); };
      auto A = [&](int v) { const int kb = v >> 3, qb = (v >> 2) & 1, r = v & 3; ls[qb][r & 1] += sc[kb][qb][r]; };
      auto C = [&](int kg, int qb, int half) {
        if (LD_ATTN_ABLATE & 8) { pw[kg][qb][2 * half] = __float_as_uint(sc[2 * kg + half][qb][0]); pw[kg][qb][2 * half + 1] = __float_as_uint(sc[2 * kg + half][qb][2]); return; }
        pw[kg][qb][2 * half] = pack_bf16x2(sc[2 * kg + half][qb][0], sc[2 * kg + half][qb][1]);
        pw[kg][qb][2 * half + 1] = pack_bf16x2(sc[2 * kg + half][qb][2], sc[2 * kg + half][qb][3]);
      };
      auto VF = [&](int g) { if (!(LD_ATTN_ABLATE & (2 | 32)) && !((LD_ATTN_ABLATE & 2048) && (g & 1))) vf[g >> 1][g & 1] = *(const bf16x8_t*)(smem + vofs[g & 1] + vslot * KTILE_BYTES + (g >> 1) * 2048); };
      auto KF = [&](int g) {                                      // g = kb*2 + ks
        if ((LD_ATTN_ABLATE & 1024) && (g & 1)) { asm volatile("" : "+v"(kf[g >> 1][g & 1])); return; }      // not re-read, but opaque: keeps the QK^T MFMAs in the loop
        if (HAS_K2 && !(LD_ATTN_ABLATE & (2 | 64))) kf[g >> 1][g & 1] = *(const bf16x8_t*)(smem + kofs[g & 1] + k2slot * KTILE_BYTES + (g >> 2) * 4096 + ((g >> 1) & 1) * 512);
      };
      auto QK = [&](int g) {                                      // g = ks*8 + kb*2 + qb
        const int ks = g >> 3, kb = (g >> 1) & 3, qb = g & 1;
        if (HAS_QK) sn[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kb][ks], qf[qb][ks], ks == 0 ? zero4 : sn[kb][qb], 0, 0, 0);
      };
      auto PV = [&](int g) {                                      // g = kg*8 + db*2 + qb
        const int kg = g >> 3, db = (g >> 1) & 3, qb = g & 1;
        o[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[db][kg], __builtin_bit_cast(bf16x8_t, pw[kg][qb]), o[db][qb], 0, 0, 0);
      };
      auto SUM = [&](int kg, int qb) {
        if (!(LD_ATTN_ABLATE & 16)) lacc[qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, __builtin_bit_cast(bf16x8_t, pw[kg][qb]), lacc[qb], 0, 0, 0);
      };
      auto DMA = [&](int g) {                 // piece g of the period's 2 * NPW
        if (do_dma && g < 2 * NPW && !(LD_ATTN_ABLATE & 4) && !((LD_ATTN_ABLATE & 4096) && (g & 1))) dma_piece(g % NPW, g < NPW ? dslot0 : dslot1, (LD_ATTN_ABLATE & 128) ? 0 : (g < NPW ? dt0 : dt1));
      };
      if constexpr (MSUM) {
        // ---- phase 1: QK^T of tile j+1 (16 MFMAs) over exp2 of scores 18..31 of tile j, the packing of P[0][*], the V_j
        //      fragment reads and the period's DMA pieces ----
        QK(0);  E(18); C(0, 0, 0); VF(0); DMA(0); FENCE();
        QK(1);  E(19); FENCE();
        QK(2);  E(20); C(0, 1, 0); VF(1); DMA(1); FENCE();
        QK(3);  E(21); FENCE();
        QK(4);  E(22); C(0, 0, 1); VF(2); DMA(2); FENCE();
        QK(5);  E(23); FENCE();
        QK(6);  E(24); C(0, 1, 1); VF(3); DMA(3); FENCE();
        QK(7);  E(25); FENCE();
        QK(8);  E(26); VF(4); DMA(4); FENCE();
        QK(9);  E(27); FENCE();
        QK(10); E(28); VF(5); DMA(5); FENCE();
        QK(11); E(29); FENCE();
        QK(12); E(30); VF(6); DMA(6); FENCE();
        QK(13); E(31); FENCE();
        QK(14); C(1, 0, 0); VF(7); DMA(7); FENCE();
        QK(15); C(1, 1, 0); FENCE();
        if (MASK) mask_tail(sn, j + 1);
        // ---- phase 2a: PV and row sums over key group 0 (10 MFMAs) over the packing of P[1][*] and the first exp2 of tile j+1 ----
        PV(0);  C(1, 0, 1); FENCE();
        PV(1);  C(1, 1, 1); FENCE();
        PV(2);  N(0);  KF(0); FENCE();
        PV(3);  N(1);  FENCE();
        PV(4);  N(2);  KF(1); FENCE();
        PV(5);  N(3);  FENCE();
        PV(6);  N(4);  KF(2); FENCE();
        PV(7);  N(5);  FENCE();
        SUM(0, 0); N(6); KF(3); FENCE();
        SUM(0, 1); N(7); FENCE();
        // ---- phase 2b: PV and row sums over key group 1 (10 MFMAs) over exp2 of scores 8..17 of tile j+1 and the K_{j+2} reads ----
        PV(8);  N(8);  KF(4); FENCE();
        PV(9);  N(9);  FENCE();
        PV(10); N(10); KF(5); FENCE();
        PV(11); N(11); FENCE();
        PV(12); N(12); KF(6); FENCE();
        PV(13); N(13); FENCE();
        PV(14); N(14); KF(7); FENCE();
        PV(15); N(15); FENCE();
        SUM(1, 0); N(16); FENCE();
        SUM(1, 1); N(17); FENCE();
      } else {
        // ---- phase 1: QK^T of tile j+1 (16 MFMAs) over exp2 of the key groups kg = 0 (all 16 scores: v = 0..15) and half of
        //      kg = 1 (v = 16..23), the packing of P[0][*], the V_j fragment reads and the period's DMA pieces ----
        QK(0);  E(0);  E(1);  VF(0); DMA(0); FENCE();
        QK(1);  E(2);  E(3);  A(0);  FENCE();
        QK(2);  E(4);  E(5);  VF(1); DMA(1); FENCE();
        QK(3);  E(6);  E(7);  A(1);  FENCE();
        QK(4);  E(8);  E(9);  VF(2); DMA(2); FENCE();
        QK(5);  E(10); E(11); C(0, 0, 0); FENCE();
        QK(6);  E(12); E(13); VF(3); DMA(3); FENCE();
        QK(7);  E(14); E(15); C(0, 1, 0); FENCE();
        QK(8);  E(16); A(2);  VF(4); DMA(4); FENCE();
        QK(9);  E(17); C(0, 0, 1); FENCE();
        QK(10); E(18); A(3);  VF(5); DMA(5); FENCE();
        QK(11); E(19); C(0, 1, 1); FENCE();
        QK(12); E(20); A(4);  VF(6); DMA(6); FENCE();
        QK(13); E(21); A(5);  FENCE();
        QK(14); E(22); A(6);  VF(7); DMA(7); FENCE();
        QK(15); E(23); A(7);  FENCE();
        if (MASK) mask_tail(sn, j + 1);
        // ---- phase 2a: PV over key group 0 (8 MFMAs) over the rest of exp2 (v = 24..31) and the packing of P[1][*] ----
        PV(0);  E(24); E(25); A(8);  FENCE();
        PV(1);  E(26); E(27); C(1, 0, 0); FENCE();
        PV(2);  E(28); E(29); A(9);  FENCE();
        PV(3);  E(30); E(31); C(1, 1, 0); FENCE();
        PV(4);  A(10); A(11); KF(0); FENCE();
        PV(5);  C(1, 0, 1);   A(12); FENCE();
        PV(6);  A(13); A(14); KF(1); FENCE();
        PV(7);  C(1, 1, 1);   A(15); FENCE();
        // ---- phase 2b: PV over key group 1 (8 MFMAs) over the remaining row-sum adds and the K_{j+2} fragment reads ----
        PV(8);  A(16); A(17); KF(2); FENCE();
        PV(9);  A(18); A(19); FENCE();
        PV(10); A(20); A(21); KF(3); FENCE();
        PV(11); A(22); A(23); FENCE();
        PV(12); A(24); A(25); KF(4); FENCE();
        PV(13); A(26); A(27); KF(5); FENCE();
        PV(14); A(28); A(29); KF(6); FENCE();
        PV(15); A(30); A(31); KF(7); FENCE();
      }
      // ---- end of a period (odd iteration): retire this wave's LDS reads and DMA pieces, then the barrier ----
      if (!EVEN) {
        if (LD_ATTN_ABLATE & 256) __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0) only
        else __builtin_amdgcn_s_waitcnt(0x0070);     // vmcnt(0) lgkmcnt(0)
        if (!(LD_ATTN_ABLATE & 512)) __builtin_amdgcn_s_barrier();
      }
      FENCE();
    };

    // ---- prologue: K0..K3, V0, V1 land; S_0 from K0; K_1 fragments ----
    if (kwave) { dma(0, 0); dma(1, 1); dma(2, 2); dma(3, 3); }
    else { dma(0, 0); dma(1, 1); }
    __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
    FENCE();
    load_kf(kf, 0);
    qk_tile(sA, kf);
    FENCE();
    load_kf(kf, 1);
    if (LD_ATTN_ABLATE & (2 | 32)) load_vf(vf, 0);
#pragma unroll
    for (int v = 0; v < NPRE; ++v) EXPV(sA, v);
    __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();        // every wave has K0 / K1 in registers: period 0 may refill their slots
    FENCE();

    // ---- main loop: iterations j = 0 .. n-3 (tiles j+1 and j+2 exist, tile j+1 is never the ragged one) ----
    int j = 0;
    for (; j + 4 <= n - 2; j += 4) {
      iter(sA, sB, j,     S0{}, T{}, T{}, F{});
      iter(sB, sA, j + 1, S1{}, T{}, T{}, F{});
      iter(sA, sB, j + 2, S2{}, T{}, T{}, F{});
      iter(sB, sA, j + 3, S3{}, T{}, T{}, F{});
    }
    // the final n - j = 2 .. 5 iterations (j is a multiple of 4): full iterations while tile i+2 exists, then iteration n-2
    // (masks tile n-1 if it is ragged, no K fragments to fetch) and iteration n-1 (only finishes tile n-1)
    const bool ragged = n * KT > p.Nk;
    const int rem = n - j;
    if (rem == 2) {
      if (ragged) iter(sA, sB, j, S0{}, T{}, F{}, T{}); else iter(sA, sB, j, S0{}, T{}, F{}, F{});
      iter(sB, sA, j + 1, S1{}, F{}, F{}, F{});
    } else if (rem == 3) {
      iter(sA, sB, j, S0{}, T{}, T{}, F{});
      if (ragged) iter(sB, sA, j + 1, S1{}, T{}, F{}, T{}); else iter(sB, sA, j + 1, S1{}, T{}, F{}, F{});
      iter(sA, sB, j + 2, S2{}, F{}, F{}, F{});
    } else if (rem == 4) {
      iter(sA, sB, j, S0{}, T{}, T{}, F{});
      iter(sB, sA, j + 1, S1{}, T{}, T{}, F{});
      if (ragged) iter(sA, sB, j + 2, S2{}, T{}, F{}, T{}); else iter(sA, sB, j + 2, S2{}, T{}, F{}, F{});
      iter(sB, sA, j + 3, S3{}, F{}, F{}, F{});
    } else {
      iter(sA, sB, j, S0{}, T{}, T{}, F{});
      iter(sB, sA, j + 1, S1{}, T{}, T{}, F{});
      iter(sA, sB, j + 2, S2{}, T{}, T{}, F{});
      if (ragged) iter(sB, sA, j + 3, S3{}, T{}, F{}, T{}); else iter(sB, sA, j + 3, S3{}, T{}, F{}, F{});
      iter(sA, sB, j + 4, S0{}, F{}, F{}, F{});
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      if constexpr (MSUM) {
        ltot[qb] = lacc[qb][0];
      } else {
        float l = ls[qb][0] + ls[qb][1];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        ltot[qb] = l;
      }
    }
  };

  // ---------------- safe pass: plain online softmax with a running maximum, one tile at a time (the fallback; not tuned) ----------------
  auto safe_pass = [&]() {
    float m[2] = {NEG_BIG, NEG_BIG}, ls[2] = {0.f, 0.f};
#pragma unroll
    for (int db = 0; db < 4; ++db) { o[db][0] = zero4; o[db][1] = zero4; }
    for (int t = 0; t < n; ++t) {
      __syncthreads();                                 // every wave is done with slot 0 of the previous tile
      dma(0, t);                                       // K waves: K_t -> K slot 0; V waves: V_t -> V slot 0
      __builtin_amdgcn_s_waitcnt(0x0070);
      __syncthreads();
      bf16x8_t kf[4][2], vf[4][2];
      f32x4_t s[4][2];
      load_kf(kf, 0); load_vf(vf, 0);
      qk_tile(s, kf);
      if ((t + 1) * KT > p.Nk) mask_tail(s, t);
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        float mx = NEG_BIG;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kb][qb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m[qb], mx);
        const float alpha = __builtin_amdgcn_exp2f(m[qb] - mn);
        m[qb] = mn;
        ls[qb] *= alpha;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[db][qb][r] *= alpha;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) { s[kb][qb][r] = __builtin_amdgcn_exp2f(s[kb][qb][r] - mn); ls[qb] += s[kb][qb][r]; }
      }
#pragma unroll
      for (int kg = 0; kg < 2; ++kg)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
          u32x4_t pw;
          pw[0] = pack_bf16x2(s[2 * kg][qb][0], s[2 * kg][qb][1]);     pw[1] = pack_bf16x2(s[2 * kg][qb][2], s[2 * kg][qb][3]);
          pw[2] = pack_bf16x2(s[2 * kg + 1][qb][0], s[2 * kg + 1][qb][1]); pw[3] = pack_bf16x2(s[2 * kg + 1][qb][2], s[2 * kg + 1][qb][3]);
#pragma unroll
          for (int db = 0; db < 4; ++db)
            o[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[db][kg], __builtin_bit_cast(bf16x8_t, pw), o[db][qb], 0, 0, 0);
        }
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float l = ls[qb];
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
      ltot[qb] = l;
    }
  };

  bool redo = force_safe != 0;
  if (!redo) {
    fast_pass();
    // 2^-80 <= l <= 2^110 (NaN fails): see the header comment of ld_attn_pipe.hip
    bool bad = false;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
      bad = bad || (!(ltot[qb] >= 8.2718061e-25f && ltot[qb] <= 1.2980742e33f) && (q0 + qb * 16 + l16 < p.Nq));
    int* flags = (int*)(smem + 8 * KTILE_BYTES);
    const bool wbad = __any(bad);
    if (lane == 0) flags[wave] = wbad ? 1 : 0;
    __syncthreads();
    redo = false;
#pragma unroll
    for (int w = 0; w < NW; ++w) redo = redo || flags[w] != 0;
    __syncthreads();
  }
  if (redo) safe_pass();
  // Every LDS-DMA of this workgroup has LANDED before the workgroup ends: the loops above run ahead of the tiles they consume;
  // a wave that ended with buffer_load ... lds in flight would let the data
  // arrive in LDS that may by then belong to the next workgroup on this CU.  (Round 5: added while hunting the co-residency bug
  // that turned out to be the packed-fp32 one -- csrc/build.sh -- and kept: it measures at 0 us of a 3.6 ms launch.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int q = q0 + qb * 16 + l16;
    const float inv = ltot[qb] > 0.f ? 1.0f / ltot[qb] : 0.f;
    if (q < p.Nq) {
      bf16_t* orow = p.O + (long)b * p.o_bs + (long)q * p.o_rs + h * D + h4 * 4;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        u32x2_t w2;
        w2[0] = pack_bf16x2(o[db][qb][0] * inv, o[db][qb][1] * inv);
        w2[1] = pack_bf16x2(o[db][qb][2] * inv, o[db][qb][3] * inv);
        *(u32x2_t*)(orow + db * 16) = w2;
      }
    }
  }
}
#undef FENCE

__global__ __launch_bounds__(256, 2) void ld_attn_p16_w4_kernel(AttnParams p, int force_safe) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  attn_p16_body<4, true>(p, force_safe, smem);
}
__global__ __launch_bounds__(512, 2) void ld_attn_p16_w8_kernel(AttnParams p, int force_safe) {       // 256 query rows per workgroup
  extern __shared__ __attribute__((aligned(16))) char smem[];
  attn_p16_body<8, true>(p, force_safe, smem);
}
__global__ __launch_bounds__(256, 2) void ld_attn_p16a_w4_kernel(AttnParams p, int force_safe) {      // row sums by v_add_f32 (A/B timing)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  attn_p16_body<4, false>(p, force_safe, smem);
}

}  // namespace

void ld_attn_set_last_kernel(const char* name);   // ld_attn.hip
void ld_attn_set_fallback_source(const unsigned* src, int kind);   // ld_attn.hip

// LD_ATTN_SAFE=1 forces the running-max pass (testing); LD_ATTN_MSUM=0 takes the row sums by v_add_f32 (A/B timing).
int ld_attn_p16_launch(const AttnParams& p, hipStream_t st) {
  constexpr int SMEM = 8 * KTILE_BYTES + 64;
  static int safe = -1, msum = 1, nw = 4;
  if (safe < 0) {
    const char* e = getenv("LD_ATTN_SAFE"); safe = e ? atoi(e) : 0;
    const char* m = getenv("LD_ATTN_MSUM"); if (m) msum = atoi(m);
    const char* w = getenv("LD_ATTN_NW"); if (w && atoi(w) == 8) nw = 8;
  }
  static thread_local LdSmemCache c4{}, c4a{}, c8{};
  ld_attn_set_fallback_source(nullptr, safe ? 0 : 2);       // a fast pass with a window, no count kept
  if (nw == 8 && msum) {
    if (int rc = ld_ensure_dyn_smem((const void*)ld_attn_p16_w8_kernel, SMEM, &c8)) return rc;
    ld_attn_set_last_kernel(safe ? "ld_attn_p16_w8_kernel[safe pass forced]" : "ld_attn_p16_w8_kernel");
    hipLaunchKernelGGL(ld_attn_p16_w8_kernel, dim3((unsigned)((long)p.B * p.H * ((p.Npad + 255) / 256))), dim3(512), SMEM, st, p, safe);
    return ld_check_launch("ld_attn_fwd_bf16(p16 w8)");
  }
  if (int rc = msum ? ld_ensure_dyn_smem((const void*)ld_attn_p16_w4_kernel, SMEM, &c4)
                    : ld_ensure_dyn_smem((const void*)ld_attn_p16a_w4_kernel, SMEM, &c4a)) return rc;
  dim3 grid((unsigned)((long)p.B * p.H * ((p.Npad + 127) / 128)));
  if (msum) {
    ld_attn_set_last_kernel(safe ? "ld_attn_p16_w4_kernel[safe pass forced]" : "ld_attn_p16_w4_kernel");
    hipLaunchKernelGGL(ld_attn_p16_w4_kernel, grid, dim3(256), SMEM, st, p, safe);
  } else {
    ld_attn_set_last_kernel(safe ? "ld_attn_p16a_w4_kernel[safe pass forced]" : "ld_attn_p16a_w4_kernel");
    hipLaunchKernelGGL(ld_attn_p16a_w4_kernel, grid, dim3(256), SMEM, st, p, safe);
  }
  return ld_check_launch("ld_attn_fwd_bf16(p16)");
}

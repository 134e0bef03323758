// The query-block body of the 64-query-row attention tile (see ld_attn_q64.hip for the design), shared by the headline kernels
// (ld_attn_q64.hip: Q64_FAST) and the exact two-pass form (ld_attn_q64_exact.hip: Q64_EXACT).  Include inside an anonymous namespace,
// after ld_attn.h.
#define FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ int q64_swz_k(int r) { return ((r >> 1) & 1) | (((r >> 3) & 3) << 1); }

constexpr int Q64_NW = 4;               // waves per workgroup
constexpr int Q64_ROWS = Q64_NW * 64;   // query rows per workgroup

// Two forms of one query block's work:
//   Q64_FAST   the max-free fast pass; if a row's denominator left its window, the plain running-max pass below (in the same
//              workgroup, one tile at a time) -- ld_attn_q64_kernel / ld_attn_q64_dyn_kernel (ld_attn_q64.hip), the headline path;
//   Q64_EXACT  (round 6, ld_attn_q64_exact.hip: ld_attn_fwd_bf16_exact) pass A = the row maxima of the scaled scores over all keys
//              (QK^T only, K tiles through all eight LDS slots), pass B = the SAME pipelined loop as the fast pass with -max[q] as
//              the C operand of every score block's first MFMA, so every probability is 2^(s - max) <= 1 and every denominator lies
//              in [1, Nk]: exact safe softmax for any logit range at ~1.55 x the fast pass, with no data-dependent re-run.
enum { Q64_FAST = 0, Q64_EXACT = 1 };

template <int MODE>
__device__ __forceinline__ void attn_q64_body(const AttnParams& p, int force_safe, char* smem, const int bid) {   // smem: K slots 0..3 | V^T slots 0..3 | flag words; bid: remapped block index
  constexpr int NW = Q64_NW;
  constexpr int VBASE = 4 * KTILE_BYTES;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, h4 = lane >> 4;
  constexpr int NPW = 16 / NW;                    // LDS-DMA pieces per wave and tile
  const int nqb = (p.Npad + Q64_ROWS - 1) / Q64_ROWS;
  const int n = (p.Nk + KT - 1) / KT;             // >= 6 (launcher)
  const int NH = 2 * n;                           // halves

  const int bh = bid / nqb, qblk = bid - bh * nqb;
  const int b = bh / p.H, h = bh - b * p.H;
  const bf16_t* Qb = p.Q + (long)bh * p.Npad * D;
  const bf16_t* Kb = p.K + (long)bh * p.Npad * D;
  const bf16_t* Vb = p.Vt + (long)bh * D * p.Npad;
  const int q0 = qblk * Q64_ROWS + wave * 64;
  if (qblk * Q64_ROWS >= p.Nq) return;
#ifdef LD_Q64_TRACE   // timing builds (tools/attn_q64_trace.py): when, where and for how many cycles every workgroup ran, into kt_min
  const unsigned long long t_real0 = __builtin_amdgcn_s_memrealtime(), t_cyc0 = __builtin_amdgcn_s_memtime();
#endif

  // Q^T fragments (B operand): rows q0 + qb*16 + l16, d = ks*32 + h4*8 .. + 8, pre-multiplied by scale * log2(e)
  bf16x8_t qf[4][2];
#pragma unroll
  for (int qb = 0; qb < 4; ++qb) {
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

  // LDS-DMA: waves 0, 1 bring K tiles (rows = keys), waves 2, 3 V^T tiles (rows = d)
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
    const int chunk = (lane & 7) ^ (kwave ? q64_swz_k(r) : ((r >> 1) & 7));
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

  // fragment read offsets: K block 2*kg + b, k-step ks: kofs[ks] + slot*8K + kg*4096 + b*512;  V^T block db, key group kg: vofs[kg] + slot*8K + db*2048
  int kofs[2], vofs[2];
  {
    const int key = 8 * (l16 >> 2) + (l16 & 3);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = ks * 4 + h4;
      kofs[ks] = key * 128 + ((c ^ q64_swz_k(key)) << 4);
      vofs[ks] = VBASE + l16 * 128 + ((c ^ ((l16 >> 1) & 7)) << 4);
    }
  }

  f32x4_t o[4][4];                                 // O^T [db][qb]
  const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4_t cinit[4] = {zero4, zero4, zero4, zero4};  // Q64_EXACT: -max of query l16 of block qb, the score blocks' initial accumulator
  auto C0 = [&](int qb) -> f32x4_t { if constexpr (MODE == Q64_EXACT) return cinit[qb]; else return zero4; };

  // keys of half hh (tile hh >> 1, key group hh & 1) past Nk -> -inf-like scores
  auto mask_half = [&](f32x4_t (&s)[2][4], int hh) {
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
      const int key0 = (hh >> 1) * KT + (hh & 1) * 32 + h4 * 8 + bb * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (key0 + r >= p.Nk) {
#pragma unroll
          for (int qb = 0; qb < 4; ++qb) s[bb][qb][r] = NEG_BIG;
        }
    }
  };
  auto load_kh = [&](bf16x8_t (&kf)[2][2], int slot, int kg) {
#pragma unroll
    for (int bb = 0; bb < 2; ++bb)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        kf[bb][ks] = *(const bf16x8_t*)(smem + kofs[ks] + slot * KTILE_BYTES + kg * 4096 + bb * 512);
  };
  auto load_vh = [&](bf16x8_t (&vf)[4], int slot, int kg) {
#pragma unroll
    for (int db = 0; db < 4; ++db) vf[db] = *(const bf16x8_t*)(smem + vofs[kg] + slot * KTILE_BYTES + db * 2048);
  };
  auto qk_half = [&](f32x4_t (&s)[2][4], const bf16x8_t (&kf)[2][2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int bb = 0; bb < 2; ++bb)
#pragma unroll
        for (int qb = 0; qb < 4; ++qb)
          s[bb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[bb][ks], qf[qb][ks], ks == 0 ? C0(qb) : s[bb][qb], 0, 0, 0);
  };
  auto pack_p = [&](u32x4_t& pw, const f32x4_t (&s)[2][4], int qb) {
    pw[0] = pack_bf16x2(s[0][qb][0], s[0][qb][1]); pw[1] = pack_bf16x2(s[0][qb][2], s[0][qb][3]);
    pw[2] = pack_bf16x2(s[1][qb][0], s[1][qb][1]); pw[3] = pack_bf16x2(s[1][qb][2], s[1][qb][3]);
  };

  // ---------------- fast pass: no running maximum (see ld_attn_pipe.hip for the argument and the window test) ----------------
  float ltot[4] = {0.f, 0.f, 0.f, 0.f};
  auto fast_pass = [&]() {
    f32x4_t lacc[4] = {zero4, zero4, zero4, zero4};           // softmax denominators from the matrix pipe: ones(16 x 32) . P[qb]
    const bf16x8_t ones = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int qb = 0; qb < 4; ++qb) o[db][qb] = zero4;
    f32x4_t sA[2][4], sB[2][4];
    bf16x8_t kf[2][2], vf[4];
    // the 32 scores of a lane in a half are numbered v = b*16 + qb*4 + r
    auto EXPV = [&](f32x4_t (&s)[2][4], int v) { const int bb = v >> 4, qb = (v >> 2) & 3, r = v & 3; s[bb][qb][r] = __builtin_amdgcn_exp2f(s[bb][qb][r]); };
    constexpr int NPRE = 18;                        // scores 0..NPRE-1 of half h+1 are exponentiated under the PV MFMAs of half h

    // One pipelined iteration on half hh (phase PH = hh & 7 fixes every slot).  sc = S_hh (the first NPRE already probabilities),
    // sn receives S_{hh+1}; the K fragments of half hh+2 are fetched and this half's share of the period's DMA pieces issued
    // (past the end the tile index is clamped: a re-fetch of the last tile); mask: half hh+1 holds keys past Nk (possibly all).
    auto iter = [&](f32x4_t (&sc)[2][4], f32x4_t (&sn)[2][4], int hh, auto phase_c, bool mask) {
      constexpr int PH = decltype(phase_c)::value;
      constexpr int kg = PH & 1;
      constexpr int vslot = (PH >> 1) & 3;                       // slot of this half's tile
      constexpr int k2slot = ((PH + 2) >> 1) & 3;                // slot of the tile of half hh + 2 (same key group kg)
      constexpr int PER = PH & 3;                                // position inside the period (two tiles)
      constexpr int pslot = vslot & 2;                           // slot of the period's first tile
      const int tp = (hh >> 1) - ((PH >> 1) & 1);                // the period's first tile
      int dt0 = kwave ? tp + 3 : tp + 2, dt1 = dt0 + 1;
      dt0 = dt0 < n ? dt0 : n - 1; dt1 = dt1 < n ? dt1 : n - 1;
      constexpr int kslot0 = (pslot + 3) & 3, kslot1 = pslot, vslot0 = (pslot + 2) & 3, vslot1 = (pslot + 3) & 3;
      const int dslot0 = kwave ? kslot0 : vslot0, dslot1 = kwave ? kslot1 : vslot1;
      u32x4_t pw[4];                                              // P fragments [qb]
      auto E = [&](int v) { EXPV(sc, v); };
      auto N = [&](int v) { EXPV(sn, v); };
      auto C = [&](int qb, int bb) {                              // two packed words of P fragment qb
        pw[qb][2 * bb] = pack_bf16x2(sc[bb][qb][0], sc[bb][qb][1]);
        pw[qb][2 * bb + 1] = pack_bf16x2(sc[bb][qb][2], sc[bb][qb][3]);
      };
      auto VF = [&](int db) { vf[db] = *(const bf16x8_t*)(smem + vofs[kg] + vslot * KTILE_BYTES + db * 2048); };
      auto KF = [&](int g) {                                      // g = b*2 + ks
        kf[g >> 1][g & 1] = *(const bf16x8_t*)(smem + kofs[g & 1] + k2slot * KTILE_BYTES + kg * 4096 + (g >> 1) * 512);
      };
      auto QK = [&](int g) {                                      // g = ks*8 + b*4 + qb
        const int ks = g >> 3, bb = (g >> 2) & 1, qb = g & 3;
        sn[bb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[bb][ks], qf[qb][ks], ks == 0 ? C0(qb) : sn[bb][qb], 0, 0, 0);
      };
      auto PV = [&](int g) {                                      // g = qb*4 + db
        const int qb = g >> 2, db = g & 3;
        o[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[db], __builtin_bit_cast(bf16x8_t, pw[qb]), o[db][qb], 0, 0, 0);
      };
      auto SUM = [&](int qb) {
        lacc[qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, __builtin_bit_cast(bf16x8_t, pw[qb]), lacc[qb], 0, 0, 0);
      };
      auto DMA = [&](int i) {                 // piece 2 * PER + i of the period's 2 * NPW
        constexpr int g0 = 2 * PER;
        const int g = g0 + i;
        dma_piece(g % NPW, g < NPW ? dslot0 : dslot1, g < NPW ? dt0 : dt1);
      };
      // ---- phase 1: QK^T of half hh+1 (16 MFMAs) over exp2 of scores 18..31 of half hh, the packing of P, the V^T fragment
      //      reads of half hh and this half's two DMA pieces ----
      QK(0);  E(18); C(0, 0); VF(0); DMA(0); FENCE();
      QK(1);  E(19); FENCE();
      QK(2);  E(20); C(1, 0); VF(1); DMA(1); FENCE();
      QK(3);  E(21); C(0, 1); FENCE();
      QK(4);  E(22); C(2, 0); VF(2); FENCE();
      QK(5);  E(23); FENCE();
      QK(6);  E(24); C(3, 0); VF(3); FENCE();
      QK(7);  E(25); C(1, 1); FENCE();
      QK(8);  E(26); FENCE();
      QK(9);  E(27); FENCE();
      QK(10); E(28); C(2, 1); FENCE();
      QK(11); E(29); FENCE();
      QK(12); E(30); FENCE();
      QK(13); E(31); FENCE();
      QK(14); C(3, 1); FENCE();
      QK(15); FENCE();
      if (mask) mask_half(sn, hh + 1);
      // ---- phase 2: PV and row sums of half hh (20 MFMAs) over exp2 of scores 0..17 of half hh+1 and the K fragment reads
      //      of half hh+2 ----
      PV(0);  N(0);  KF(0); FENCE();
      PV(1);  N(1);  FENCE();
      PV(2);  N(2);  KF(1); FENCE();
      PV(3);  N(3);  FENCE();
      SUM(0); N(4);  KF(2); FENCE();
      PV(4);  N(5);  FENCE();
      PV(5);  N(6);  KF(3); FENCE();
      PV(6);  N(7);  FENCE();
      PV(7);  N(8);  FENCE();
      SUM(1); N(9);  FENCE();
      PV(8);  N(10); FENCE();
      PV(9);  N(11); FENCE();
      PV(10); N(12); FENCE();
      PV(11); N(13); FENCE();
      SUM(2); N(14); FENCE();
      PV(12); N(15); FENCE();
      PV(13); N(16); FENCE();
      PV(14); N(17); FENCE();
      PV(15); FENCE();
      SUM(3); FENCE();
      // ---- end of a period: retire this wave's LDS reads and DMA pieces, then the barrier ----
      if (PER == 3) {
        __builtin_amdgcn_s_waitcnt(0x0070);          // vmcnt(0) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
      }
      FENCE();
    };
    using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
    using P2 = std::integral_constant<int, 2>; using P3 = std::integral_constant<int, 3>;
    using P4 = std::integral_constant<int, 4>; using P5 = std::integral_constant<int, 5>;
    using P6 = std::integral_constant<int, 6>; using P7 = std::integral_constant<int, 7>;

    // ---- prologue: K0..K2, V0, V1 land; S_0 from K0 key group 0; the K fragments of half 1 ----
    if (kwave) { dma(0, 0); dma(1, 1); dma(2, 2); }
    else { dma(0, 0); dma(1, 1); }
    __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
    FENCE();
    load_kh(kf, 0, 0);
    qk_half(sA, kf);
    FENCE();
    load_kh(kf, 0, 1);
#pragma unroll
    for (int v = 0; v < NPRE; ++v) EXPV(sA, v);
    __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();        // every wave has K0's fragments in registers: period 0 may refill its slot
    FENCE();

    // ---- the loop: eight halves (four tiles) per trip, over ALL halves rounded up to a multiple of eight.  There is no
    //      peeled tail: halves past the last one (and the keys of the last tile past Nk) are masked to -inf-like scores, i.e.
    //      exactly zero probabilities, their DMA re-fetches the last tile (finite data), and the QK^T issued for the half after
    //      the last is never used.  At most six extra halves per 2 * n (0.7 % at the DiT shape; since round 5 the last trip ends after four halves when the
    //      other four hold no key) instead of 28 peeled iteration
    //      bodies whose register pressure spilled (640 MB of scratch writes per launch).  The trips that need no mask run
    //      from a copy of the loop without the mask tests: a uniform branch per half costs the whole step 2.5 %. ----
    const int hmask = p.Nk / 32;                    // first half that holds a key >= Nk
    const int hend = (p.Nk + 31) / 32;              // first half that holds no key at all
    int hh = 0;
    for (; hh + 8 < hmask; hh += 8) {               // trips that compute no half >= hmask: no mask tests in the instruction stream
      iter(sA, sB, hh,     P0{}, false);
      iter(sB, sA, hh + 1, P1{}, false);
      iter(sA, sB, hh + 2, P2{}, false);
      iter(sB, sA, hh + 3, P3{}, false);
      iter(sA, sB, hh + 4, P4{}, false);
      iter(sB, sA, hh + 5, P5{}, false);
      iter(sA, sB, hh + 6, P6{}, false);
      iter(sB, sA, hh + 7, P7{}, false);
    }
    for (; hh < NH; hh += 8) {                      // the last one or two trips
      iter(sA, sB, hh,     P0{}, hh + 1 >= hmask);
      iter(sB, sA, hh + 1, P1{}, hh + 2 >= hmask);
      iter(sA, sB, hh + 2, P2{}, hh + 3 >= hmask);
      iter(sB, sA, hh + 3, P3{}, hh + 4 >= hmask);
      if (hh + 4 >= hend) break;                    // round 5: the second half of the last trip holds no key (a period boundary: every
                                                    // wave agrees, the barrier of period 3 has been passed): at the DiT shape 556 of
                                                    // the 560 rounded-up halves are left, -0.9 % per launch, the same bits
                                                    // (profiles/r05_attn_half_trip_exit_ab.txt)
      iter(sA, sB, hh + 4, P4{}, hh + 5 >= hmask);
      iter(sB, sA, hh + 5, P5{}, hh + 6 >= hmask);
      iter(sA, sB, hh + 6, P6{}, hh + 7 >= hmask);
      iter(sB, sA, hh + 7, P7{}, hh + 8 >= hmask);
    }
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) ltot[qb] = lacc[qb][0];
  };

  // ---------------- safe pass: plain online softmax with a running maximum, one tile at a time (the fallback; not tuned) ----------------
  auto safe_pass = [&]() {
    float m[4], ls[4];
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) { m[qb] = NEG_BIG; ls[qb] = 0.f; }
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int qb = 0; qb < 4; ++qb) o[db][qb] = zero4;
    for (int t = 0; t < n; ++t) {
      __syncthreads();                                 // every wave is done with slot 0 of the previous tile
      dma(0, t);                                       // K waves: K_t -> K slot 0; V waves: V_t -> V slot 0
      __builtin_amdgcn_s_waitcnt(0x0070);
      __syncthreads();
#pragma unroll
      for (int kg = 0; kg < 2; ++kg) {
        bf16x8_t kf[2][2], vf[4];
        f32x4_t s[2][4];
        load_kh(kf, 0, kg); load_vh(vf, 0, kg);
        qk_half(s, kf);
        if ((t + 1) * KT > p.Nk) mask_half(s, 2 * t + kg);
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
          float mx = NEG_BIG;
#pragma unroll
          for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[bb][qb][r]);
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
          for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int r = 0; r < 4; ++r) { s[bb][qb][r] = __builtin_amdgcn_exp2f(s[bb][qb][r] - mn); ls[qb] += s[bb][qb][r]; }
          u32x4_t pw;
          pack_p(pw, s, qb);
#pragma unroll
          for (int db = 0; db < 4; ++db)
            o[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[db], __builtin_bit_cast(bf16x8_t, pw), o[db][qb], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
      float l = ls[qb];
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
      ltot[qb] = l;
    }
  };

  // ---------------- Q64_EXACT pass A: row maxima of the scaled scores (log2 domain) over all keys ----------------
  // QK^T only: all four waves bring K tiles (two 1 KB pieces per wave and tile) through all eight LDS slots, four tiles per group,
  // group g + 1 in flight under the MFMAs of group g, one barrier per group.
  auto max_pass = [&]() {
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, 0x7fffffff, 0x00020000);
    const int ng = (n + 3) / 4;
    float m[4] = {NEG_BIG, NEG_BIG, NEG_BIG, NEG_BIG};
    // Groups [g0, g1) with or without the key mask.  Each call computes its per-lane offsets from its own OPAQUE copy of the lane
    // id, so that they are values of this loop only.  Shared -- with pass B, whose loop has no register to spare in this mode, or
    // with the masked form, which keeps its scores in scratch -- they were spilled for their whole lifetime and reloaded in every
    // trip of the unmasked loop, each reload behind an s_waitcnt vmcnt(0) that waited for the NEXT group's LDS-DMA
    // (tools/audit_spills.py found it).
    auto max_part = [&](int g0, int g1, auto maskc) {
      int lane_m = lane;
      asm volatile("" : "+v"(lane_m));
      uint32_t ga[2];                               // a K tile = 64 keys x 128 B = eight 1 KB pieces: two per wave
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 8 + (lane_m >> 3);
        ga[i] = (uint32_t)(r * D + ((lane_m & 7) ^ q64_swz_k(r)) * 8) * 2u;
      }
      const int l16m = lane_m & 15, h4m = lane_m >> 4;
      int kofs_m[2];
      {
        const int key = 8 * (l16m >> 2) + (l16m & 3);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) kofs_m[ks] = key * 128 + (((ks * 4 + h4m) ^ q64_swz_k(key)) << 4);
      }
      auto dma_group = [&](int set, int g) {      // tiles 4g .. 4g+3 (clamped: a re-fetch of the last tile) -> slots 4 set .. 4 set + 3
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          int tile = 4 * g + t; tile = tile < n ? tile : n - 1;
#pragma unroll
          for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (__attribute__((address_space(3))) void*)(smem + (4 * set + t) * KTILE_BYTES + (wave * 2 + i) * 1024),
                                                     16, ga[i], tile * (KT * D * 2), 0, 0);
        }
      };
      if (g0 == 0 && g1 > 0) dma_group(0, 0);
      for (int g = g0; g < g1; ++g) {
        // this wave's pieces of group g have landed, its fragment reads of group g - 1 are done; then every wave's.  As ONE asm
        // statement with a memory clobber + a scheduling barrier: the builtins alone do not keep hipcc from moving the fragment
        // reads below (plain LDS loads -- it does not know that the LDS-DMA above writes what they read) across the barrier
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        FENCE();
        if (g + 1 < ng) dma_group((g + 1) & 1, g + 1);
        FENCE();
        const int set = g & 1;
        // eight halves, straight-line: hipcc pipelines the fragment reads under the MFMAs
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int kg = 0; kg < 2; ++kg) {
            bf16x8_t kf[2][2];
            f32x4_t sc[2][4];
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
              for (int ks = 0; ks < 2; ++ks)
                kf[bb][ks] = *(const bf16x8_t*)(smem + kofs_m[ks] + (4 * set + t) * KTILE_BYTES + kg * 4096 + bb * 512);
            qk_half(sc, kf);                      // (C0 = 0 here: cinit is still zero)
            if constexpr (decltype(maskc)::value) {
              const int hh = 2 * (4 * g + t) + kg;      // (mask_half on this part's own lane values)
#pragma unroll
              for (int bb = 0; bb < 2; ++bb) {
                const int key0 = (hh >> 1) * KT + (hh & 1) * 32 + h4m * 8 + bb * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const bool dead = key0 + r >= p.Nk;
#pragma unroll
                  for (int qb = 0; qb < 4; ++qb) sc[bb][qb][r] = dead ? NEG_BIG : sc[bb][qb][r];
                }
              }
            }
            // plain fmaxf, not the v_max3_f32 asm helper: the operands are MFMA results, and nothing pads the MFMA -> VALU read
            // hazard for an instruction inside an asm statement (a max3 scheduled right behind its MFMA read the accumulator's
            // OLD value now and then: row maxima that missed a key, different from run to run)
#pragma unroll
            for (int qb = 0; qb < 4; ++qb) {
              const float a = fmaxf(fmaxf(sc[0][qb][0], sc[0][qb][1]), fmaxf(sc[0][qb][2], sc[0][qb][3]));
              const float c2 = fmaxf(fmaxf(sc[1][qb][0], sc[1][qb][1]), fmaxf(sc[1][qb][2], sc[1][qb][3]));
              m[qb] = fmaxf(m[qb], fmaxf(a, c2));
            }
          }
      }
    };
    // Two loops, not one loop with a branch: only the last group can hold keys past Nk (tiles past the last one are fetched as
    // copies of the last tile and masked too: harmless for a maximum)
    const int ng_full = p.Nk / (4 * KT) < ng ? p.Nk / (4 * KT) : ng;
    max_part(0, ng_full, std::false_type{});
    max_part(ng_full, ng, std::true_type{});
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");     // every wave is done with the slots before pass B's prologue refills them
    FENCE();
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
      float v = m[qb];
      v = fmaxf(v, __shfl_xor(v, 16, 64));
      v = fmaxf(v, __shfl_xor(v, 32, 64));
      cinit[qb] = (f32x4_t){-v, -v, -v, -v};
    }
  };

  if constexpr (MODE == Q64_EXACT) {
    max_pass();
    fast_pass();
  } else {
    bool redo = force_safe != 0;
    if (!redo) {
      fast_pass();
      // 2^-80 <= l <= 2^110 (NaN fails): see the header comment of ld_attn_pipe.hip
      bool bad = false;
#pragma unroll
      for (int qb = 0; qb < 4; ++qb)
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
  }
  // Every LDS-DMA of this workgroup has LANDED before the workgroup ends: the loops above run ahead of the tiles they consume (and,
  // having no peeled tail, request tiles nobody reads); a wave that ended with buffer_load ... lds in flight would let the data
  // arrive in LDS that may by then belong to the next workgroup on this CU.  (Round 5: added while hunting the co-residency bug
  // that turned out to be the packed-fp32 one -- csrc/build.sh -- and kept: it measures at 0 us of a 3.6 ms launch.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

#pragma unroll
  for (int qb = 0; qb < 4; ++qb) {
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
#ifdef LD_Q64_TRACE
  if (p.kt_min && tid == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* rec = (unsigned long long*)p.kt_min + (long)bid * 4;
    rec[0] = t_real0; rec[1] = __builtin_amdgcn_s_memrealtime(); rec[2] = __builtin_amdgcn_s_memtime() - t_cyc0;
    rec[3] = ((unsigned long long)xcc << 32) | hw;
  }
#endif
}
#undef FENCE

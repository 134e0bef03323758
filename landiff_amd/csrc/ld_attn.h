// Shared declarations of the gfx950 attention kernels (ld_attn.hip, ld_attn_pipe.hip).
#pragma once
#include "ld_common.h"
#include "../../include/landiff_hip.h"
#include <stdlib.h>
#include <type_traits>

struct AttnParams {
  const bf16_t* Q;    // [BH][Npad][64]
  const bf16_t* K;    // [BH][Npad][64]
  const bf16_t* Vt;   // [BH][64][Npad]
  bf16_t* O;          // [B][Nq][H*64] (row stride o_rs, batch stride o_bs)
  int B, H, Nq, Nk, Npad;
  long o_bs, o_rs;
  float c;            // softmax_scale * log2(e)
  const int* fid_q;   // [Npad] or null
  const int* fid_k;   // [Npad] or null (padding keys must carry INT_MAX)
  const int* kt_min;  // [Npad/64]
  const int* kt_max;
};

namespace {

constexpr int QB = 128;   // query rows per workgroup
constexpr int KT = 64;    // keys per tile
constexpr int D = 64;
constexpr int KTILE_BYTES = KT * D * 2;       // 8 KB
constexpr int STAGE_BYTES = 2 * KTILE_BYTES;  // K + V^T
constexpr float NEG_BIG = -1.0e30f;


__device__ __forceinline__ void glds16(const bf16_t* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void*)g,
      (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ float lane32_max(float x) {
  // max(x, value of lane^32)
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float lane32_sum(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// v_max3_f32 by hand: plain fmaxf on MFMA outputs makes hipcc insert a canonicalising v_max per operand.  CAUTION: an asm statement
// gets no hazard padding -- when the operands are MFMA results the caller provides the MFMA -> VALU wait states (ld_attn.hip)
__device__ __forceinline__ float max3f(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

__device__ __forceinline__ int swap23(int i) {
  return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1);
}

}  // namespace

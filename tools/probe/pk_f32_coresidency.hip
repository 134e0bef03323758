// Minimal reproducer for the round-5 finding (profiles/r05_llm_packed_f32_under_coresidency.txt): is a packed-fp32 VALU result that
// a dependent packed-fp32 instruction consumes a few issue slots later always the value just written -- also when the SIMD is shared
// with another kernel's MFMA-heavy waves?  Each thread runs, `iters` times, the instruction sequence hipcc's SLP vectoriser emitted for
// RoPE in ld_rope_append_kernel (as ONE asm block, so that nothing is re-scheduled):
//     v_pk_mul_f32 P,  S, AB op_sel_hi:[0,1]        ; (s*a, s*b)
//     v_pk_mul_f32 AB, C, AB op_sel_hi:[0,1]        ; (c*a, c*b)   -- in place
//     <FILL independent v_pk_mul_f32>
//     v_pk_add_f32 R, AB, P  op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]     ; lo = c*a - s*b, hi = c*b - s*a
// and compares R.lo bit for bit with the same value from scalar v_mul / v_mul / v_sub.  Mismatches are counted per LANE.
// Variants: FILL = 0..3 (issue distance producer -> consumer), NOSEL (no operand swizzle: consumer reads AB and P straight),
// NOP (an s_nop 1 in front of the consumer).  Build: hipcc --offload-arch=gfx950 -O2 -fno-slp-vectorize -shared -fPIC -o libpkprobe.so pk_f32_coresidency.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f2 __attribute__((ext_vector_type(2)));

#define PK_SEQ(FILLS, NOPS, ADD)                                                                        \
  asm volatile("v_pk_mul_f32 %[p], %[ss], %[ab] op_sel_hi:[0,1]\n"                                       \
               "v_pk_mul_f32 %[ab], %[cc], %[ab] op_sel_hi:[0,1]\n" FILLS NOPS ADD                        \
               : [p] "=&v"(p), [ab] "+v"(ab), [r] "=&v"(r), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3) \
               : [ss] "v"(ss), [cc] "v"(cc), [d0] "v"(d0))
#define F1 "v_pk_mul_f32 %[d1], %[d0], %[d0]\n"
#define F2 F1 "v_pk_mul_f32 %[d2], %[d0], %[d0]\n"
#define F3 F2 "v_pk_mul_f32 %[d3], %[d0], %[d0]\n"
#define ADD_SEL "v_pk_add_f32 %[r], %[ab], %[p] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
#define ADD_STRAIGHT "v_pk_add_f32 %[r], %[ab], %[p] neg_lo:[0,1] neg_hi:[0,1]\n"      /* lo = c*a - s*a, hi = c*b - s*b */

template <int VARIANT>
__global__ void pk_victim(const float* A, const float* B, const float* C, const float* S, unsigned* bad, unsigned* first_bits, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
  float a = A[tid], b = B[tid];
  const float c = C[tid], s = S[tid];
  unsigned nbad = 0;
  for (int it = 0; it < iters; ++it) {
    f2 ab = {a, b}, p, r, d1, d2, d3;
    const f2 cc = {c, c}, ss = {s, s}, d0 = {a, c};
    if (VARIANT == 0) PK_SEQ("", "", ADD_SEL);
    else if (VARIANT == 1) PK_SEQ(F1, "", ADD_SEL);
    else if (VARIANT == 2) PK_SEQ(F2, "", ADD_SEL);                    // the compiled kernel's distance
    else if (VARIANT == 3) PK_SEQ(F3, "", ADD_SEL);
    else if (VARIANT == 4) PK_SEQ(F2, "", ADD_STRAIGHT);
    else if (VARIANT == 5) PK_SEQ(F2, "s_nop 1\n", ADD_SEL);
    else PK_SEQ(F2, "s_nop 7\n", ADD_SEL);
    const float ca = __fmul_rn(c, a), sx = __fmul_rn(s, VARIANT == 4 ? a : b);
    const float ref = __fsub_rn(ca, sx);
    if (__float_as_uint(r[0]) != __float_as_uint(ref)) {
      if (nbad == 0) { first_bits[tid * 4 + 0] = __float_as_uint(r[0]); first_bits[tid * 4 + 1] = __float_as_uint(ref);
                       first_bits[tid * 4 + 2] = __float_as_uint(a); first_bits[tid * 4 + 3] = __float_as_uint(c); }
      ++nbad;
    }
    a = __uint_as_float(__float_as_uint(a) ^ ((unsigned)(it & 7) << 3));        // keep the operands moving
    b = __uint_as_float(__float_as_uint(b) ^ ((unsigned)(it & 3) << 5));
  }
  if (nbad) atomicAdd(&bad[lane], nbad);
}

extern "C" __attribute__((visibility("default")))
int pk_probe_run(int variant, const float* A, const float* B, const float* C, const float* S, unsigned* bad, unsigned* first_bits,
                 int nblocks, int iters, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  dim3 g(nblocks), blk(64);
  switch (variant) {
    case 0: hipLaunchKernelGGL(pk_victim<0>, g, blk, 0, st, A, B, C, S, bad, first_bits, iters); break;
    case 1: hipLaunchKernelGGL(pk_victim<1>, g, blk, 0, st, A, B, C, S, bad, first_bits, iters); break;
    case 2: hipLaunchKernelGGL(pk_victim<2>, g, blk, 0, st, A, B, C, S, bad, first_bits, iters); break;
    case 3: hipLaunchKernelGGL(pk_victim<3>, g, blk, 0, st, A, B, C, S, bad, first_bits, iters); break;
    case 4: hipLaunchKernelGGL(pk_victim<4>, g, blk, 0, st, A, B, C, S, bad, first_bits, iters); break;
    case 5: hipLaunchKernelGGL(pk_victim<5>, g, blk, 0, st, A, B, C, S, bad, first_bits, iters); break;
    default: hipLaunchKernelGGL(pk_victim<6>, g, blk, 0, st, A, B, C, S, bad, first_bits, iters); break;
  }
  return (int)hipGetLastError();
}

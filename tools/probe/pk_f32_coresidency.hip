// Minimal reproducer for the round-5 finding (profiles/r05_llm_packed_f32_under_coresidency.txt): is a packed-fp32 VALU result that
// a dependent packed-fp32 instruction consumes a few issue slots later always the value just written -- also when the SIMD is shared
// with another kernel's MFMA-heavy waves?  Each thread runs, `iters` times, the instruction sequence hipcc's SLP vectoriser emitted for
// RoPE in ld_rope_append_kernel (as ONE asm block, so that nothing is re-scheduled):
//     v_pk_mul_f32 P,  S, AB op_sel_hi:[0,1]        ; (s*a, s*b)
//     v_pk_mul_f32 AB, C, AB op_sel_hi:[0,1]        ; (c*a, c*b)   -- in place
//     <FILL independent v_pk_mul_f32>
//     v_pk_add_f32 R, AB, P  op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]     ; lo = c*a - s*b, hi = c*b - s*a
// and compares R.lo bit for bit with the same value from scalar v_mul / v_mul / v_sub.  Mismatches are counted per LANE.
// Variants: 0-3 FILL = 0..3 (issue distance producer -> consumer); 4 no operand swizzle in the consumer; 5 / 6 an s_nop 1 / 7 in front
// of the consumer; 7 the consumer ALONE (products made by scalar v_mul long before, s_nop 7, then only the swizzled v_pk_add);
// 8 producers without the op_sel_hi broadcast (real {c, c} / {s, s} pairs), swizzled consumer; 9 v_pk_mov_b32 op_sel:[1,0] alone
// (low lane <- high register: is it the operand path or the fp32 pipe?).
// Build: hipcc --offload-arch=gfx950 -O2 -fno-slp-vectorize -ffp-contract=off -shared -fPIC -o libpkprobe.so pk_f32_coresidency.hip
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
    else if (VARIANT == 6) PK_SEQ(F2, "s_nop 7\n", ADD_SEL);
    else if (VARIANT == 7) {
      ab = (f2){c * a, c * b}; p = (f2){s * a, s * b};
      asm volatile("s_nop 7\n" ADD_SEL : [r] "=&v"(r) : [ab] "v"(ab), [p] "v"(p));
    } else if (VARIANT == 9) {                        // v_pk_mov_b32 with the same swizzle: low lane <- high register
      asm volatile("s_nop 7\nv_pk_mov_b32 %[r], %[ab], %[ab] op_sel:[1,0]\n" : [r] "=&v"(r) : [ab] "v"(ab));
      if (__float_as_uint(r[0]) != __float_as_uint(b)) ++nbad;
      a = __uint_as_float(__float_as_uint(a) ^ ((unsigned)(it & 7) << 3)); b = __uint_as_float(__float_as_uint(b) ^ ((unsigned)(it & 3) << 5));
      continue;
    } else {
      asm volatile("v_pk_mul_f32 %[p], %[ss], %[ab]\n"
                   "v_pk_mul_f32 %[ab], %[cc], %[ab]\n" F2 ADD_SEL
                   : [p] "=&v"(p), [ab] "+v"(ab), [r] "=&v"(r), [d1] "=&v"(d1), [d2] "=&v"(d2) : [ss] "v"(ss), [cc] "v"(cc), [d0] "v"(d0));
    }
    const float ca = __fmul_rn(c, a), sx = __fmul_rn(s, VARIANT == 4 ? a : b);
    const float ref = __fsub_rn(ca, sx);
    if (__float_as_uint(r[0]) != __float_as_uint(ref)) {
      if (nbad == 0) { first_bits[tid * 8 + 0] = __float_as_uint(r[0]); first_bits[tid * 8 + 1] = __float_as_uint(ref);
                       first_bits[tid * 8 + 2] = __float_as_uint(a); first_bits[tid * 8 + 3] = __float_as_uint(b);
                       first_bits[tid * 8 + 4] = __float_as_uint(c); first_bits[tid * 8 + 5] = __float_as_uint(s);
                       first_bits[tid * 8 + 6] = __float_as_uint(r[1]); first_bits[tid * 8 + 7] = (unsigned)it; }
      ++nbad;
    }
    a = __uint_as_float(__float_as_uint(a) ^ ((unsigned)(it & 7) << 3));        // keep the operands moving
    b = __uint_as_float(__float_as_uint(b) ^ ((unsigned)(it & 3) << 5));
  }
  if (nbad) atomicAdd(&bad[lane], nbad);
}

// ---- synthetic co-resident load: 256-thread workgroups looping over ONE kind of instruction (8 independent ones per trip) ----
// 0 v_fma_f32   1 v_exp_f32   2 v_cvt_pk_bf16_f32   3 ds_read_b128   4 v_permlane32_swap   5 v_pk_fma_f32 (no modifiers)
// 6 v_pk_mul_f32 op_sel_hi:[0,1]   7 v_mfma_f32_16x16x32_bf16   8 v_mfma_f32_32x32x16_bf16   9 ds_write_b128   10 v_mov_b32 dpp row_shr
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef short s8 __attribute__((ext_vector_type(8)));
template <int KIND>
__global__ __launch_bounds__(256) void pk_aggressor(float* sink, int trips) {
  __shared__ __attribute__((aligned(16))) float lds[2048];
  const int tid = threadIdx.x;
  for (int i = tid; i < 2048; i += 256) lds[i] = (float)i;
  __syncthreads();
  float x0 = tid * 0.001f + 1.0f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
  f2 p0 = {x0, x1}, p1 = {x2, x3};
  f4 acc4 = {0.f, 0.f, 0.f, 0.f}; f16v acc16 = {};
  s8 fa = {1, 2, 3, 4, 5, 6, 7, 8}, fb = {8, 7, 6, 5, 4, 3, 2, 1};
  f4 l4 = {0.f, 0.f, 0.f, 0.f};
  const unsigned laddr = (unsigned)(tid * 16) & 8176u;
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x0) : "v"(x1));
      else if (KIND == 1) asm volatile("v_exp_f32 %0, %1" : "=v"(x2) : "v"(x0));
      else if (KIND == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x3) : "v"(x0), "v"(x1));
      else if (KIND == 3) asm volatile("ds_read_b128 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(l4) : "v"(laddr));
      else if (KIND == 4) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x0), "+v"(x1));
      else if (KIND == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p0) : "v"(p1));
      else if (KIND == 6) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(p0) : "v"(p1), "v"(p1));
      else if (KIND == 7) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc4) : "v"(fa), "v"(fb));
      else if (KIND == 8) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc16) : "v"(fa), "v"(fb));
      else if (KIND == 9) asm volatile("ds_write_b128 %0, %1\ns_waitcnt lgkmcnt(0)" : : "v"(laddr), "v"(l4) : "memory");
      else asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(x2) : "v"(x0));
    }
  }
  if (x0 + x1 + x2 + x3 + p0[0] + p0[1] + acc4[0] + acc16[0] + l4[0] == 12345.678f) sink[0] = x0;
}

extern "C" __attribute__((visibility("default")))
int pk_aggressor_run(int kind, float* sink, int nblocks, int trips, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  dim3 g(nblocks), blk(256);
#define AGG(K) case K: hipLaunchKernelGGL(pk_aggressor<K>, g, blk, 0, st, sink, trips); break;
  switch (kind) { AGG(0) AGG(1) AGG(2) AGG(3) AGG(4) AGG(5) AGG(6) AGG(7) AGG(8) AGG(9) default: hipLaunchKernelGGL(pk_aggressor<10>, g, blk, 0, st, sink, trips); }
#undef AGG
  return (int)hipGetLastError();
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
    case 6: hipLaunchKernelGGL(pk_victim<6>, g, blk, 0, st, A, B, C, S, bad, first_bits, iters); break;
    case 7: hipLaunchKernelGGL(pk_victim<7>, g, blk, 0, st, A, B, C, S, bad, first_bits, iters); break;
    case 8: hipLaunchKernelGGL(pk_victim<8>, g, blk, 0, st, A, B, C, S, bad, first_bits, iters); break;
    default: hipLaunchKernelGGL(pk_victim<9>, g, blk, 0, st, A, B, C, S, bad, first_bits, iters); break;
  }
  return (int)hipGetLastError();
}

// Shared device/host helpers for the LanDiff gfx950 kernels.
// Everything here is CDNA4-only (wave64, MFMA 32x32x16 bf16, LDS-DMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LD_API extern "C" __attribute__((visibility("default")))

// ---- error plumbing (thread-local message, negative return codes) ----
enum {
  LD_OK = 0,
  LD_ERR_INVALID = -1,   // bad argument (shape/alignment/null pointer)
  LD_ERR_LAUNCH = -2,    // hipLaunch error
  LD_ERR_UNSUPPORTED = -3
};
int ld_set_error(int code, const char* fmt, ...);
int ld_check_launch(const char* what);

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute: the memo of "already raised to N bytes" is kept
// per device and per host thread (declare the cache `static thread_local` at the call site, one per kernel), and the
// return code of the runtime call is checked.
struct LdSmemCache { size_t bytes[16]; };     // indexed by device ordinal; devices >= 16 are never cached
int ld_ensure_dyn_smem(const void* kernel, size_t bytes, LdSmemCache* cache);

// Tuning knobs (LD_* environment variables) are read ONCE per process -- getenv races with setenv / os.environ writes from other
// host threads -- unless LD_TUNING=1 was set when the library first looked: then every call re-reads them, so that one process
// (tests, tools/*_ab.py) can alternate two forms.  `cache` is a static int initialised to LD_KNOB_UNSET at the call site.
#define LD_KNOB_UNSET (-0x7fffffff)
int ld_knob(const char* name, int dflt, int* cache);

#define LD_REQUIRE(cond, ...)                                  \
  do {                                                         \
    if (!(cond)) return ld_set_error(LD_ERR_INVALID, __VA_ARGS__); \
  } while (0)

typedef uint16_t bf16_t;  // raw bf16 bits

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // MFMA A/B fragment (4 VGPR)
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;  // 32x32 accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

// ---- bf16 <-> f32 (round-to-nearest-even, same as torch's c10::BFloat16) ----
__device__ __forceinline__ float bf2f(bf16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}
// gfx950 has a native RNE pack-convert (v_cvt_pk_bf16_f32); clang emits it for float -> __bf16 conversions.
typedef __attribute__((ext_vector_type(2))) float ld_f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 ld_bf16x2_t;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  ld_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, ld_bf16x2_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf16x2(f, 0.0f) & 0xffffu); }
// round an f32 value to the nearest bf16 and return it as f32 (emulates a bf16 op output)
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// Element pairs: a packed bf16 pair unpacks into an adjacent register pair (v_pk_{add,mul,fma}_f32 operands), and one
// v_cvt_pk_bf16_f32 rounds both -- 1.5 VALU operations per bf16 rounding of an element instead of 3.
__device__ __forceinline__ ld_f32x2_t unpack_bf16x2(uint32_t w) { return (ld_f32x2_t){bf_lo(w), bf_hi(w)}; }
__device__ __forceinline__ uint32_t pack_bf16x2(ld_f32x2_t v) { return pack_bf16x2(v[0], v[1]); }
__device__ __forceinline__ ld_f32x2_t rbf2(ld_f32x2_t v) { return unpack_bf16x2(pack_bf16x2(v)); }

// ---- activations (fp32 math, matching torch's CPU/CUDA formulas) ----
// 0.5*x*(1+tanh(u)) == x / (1 + exp(-2u)): one v_exp_f32 + one v_rcp_f32 instead of the tanhf call (the epilogue of the
// 4h GEMM is VALU-bound otherwise); agrees with the tanh form to a few fp32 ulps, far inside the bf16 output rounding.
__device__ __forceinline__ float act_gelu_tanh(float x) {
  const float kBeta = 0.7978845608028654f;  // sqrt(2/pi)
  const float kKappa = 0.044715f;
  const float u = kBeta * (x + kKappa * x * x * x);
  const float e = __builtin_amdgcn_exp2f(-2.8853900817779268f * u);   // exp(-2u)
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}
// the same operations on an element pair (v_pk_mul / v_pk_fma for the polynomial; the two transcendentals stay scalar)
__device__ __forceinline__ ld_f32x2_t act_gelu_tanh2(ld_f32x2_t x) {
  const ld_f32x2_t kBeta = {0.7978845608028654f, 0.7978845608028654f}, kKappa = {0.044715f, 0.044715f};
  const ld_f32x2_t kExp = {-2.8853900817779268f, -2.8853900817779268f}, one = {1.0f, 1.0f};
  const ld_f32x2_t u = kBeta * (x + kKappa * x * x * x);
  const ld_f32x2_t a = kExp * u;
  const ld_f32x2_t d = one + (ld_f32x2_t){__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
  return x * (ld_f32x2_t){__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}
__device__ __forceinline__ float act_gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f));
}
__device__ __forceinline__ float act_silu(float x) { return x / (1.0f + __expf(-x)); }

enum { LD_ACT_NONE = 0, LD_ACT_GELU_TANH = 1, LD_ACT_GELU_ERF = 2, LD_ACT_SILU = 3, LD_ACT_TANH = 4 };

__device__ __forceinline__ float apply_act(int act, float x) {
  switch (act) {
    case LD_ACT_GELU_TANH: return act_gelu_tanh(x);
    case LD_ACT_GELU_ERF: return act_gelu_erf(x);
    case LD_ACT_SILU: return act_silu(x);
    case LD_ACT_TANH: return tanhf(x);
    default: return x;
  }
}

// ---- wave-level reductions (64 lanes) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// XCD-aware bijective remap of a linear workgroup id (8 XCDs; block b runs on XCD b % 8).
// Consecutive logical ids land on the same XCD so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int nx = 8;
  int q = nwg / nx, r = nwg % nx;
  int xcd = bid % nx, idx = bid / nx;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

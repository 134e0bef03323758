// T5 encoder pieces that the shared kernels do not cover (FLAN-T5-XXL for the LLM condition,
// landiff/llm/modules/text_encoder.py:16-146; T5-v1.1-XXL for the DiT context,
// landiff/diffusion/sgm/modules/encoders/modules.py:246-295 -- both HF transformers T5EncoderModel in bf16).
// The projections and the gated-GELU feed-forward run on ld_gemm_bf16; here: T5LayerNorm with its two bf16 roundings and
// the self-attention with the bucketed relative-position bias (no 1/sqrt(d) scaling, softmax in fp32, bf16 rounding points
// of the bf16 HF module: scores, scores + bias, probabilities, output).  Sequences are short (<= 512 tokens, once per
// prompt): a plain VALU kernel, one wave per query row, HBM/L2-resident K and V.
#include "ld_common.h"
#include "../../include/landiff_hip.h"

namespace {

// T5LayerNorm (modeling_t5.py): variance in fp32, x * rsqrt(var + eps) rounded to bf16, then weight * (that) rounded again
__global__ __launch_bounds__(256) void ld_t5_rmsnorm_kernel(const bf16_t* x, const bf16_t* w, bf16_t* out, int rows, int D, float eps) {
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
    const u32x4_t ww = *(const u32x4_t*)(w + c * 8);
    u32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      o[e] = pack_bf16x2(bf_lo(ww[e]) * rbf(bf_lo(a[e]) * rs), bf_hi(ww[e]) * rbf(bf_hi(a[e]) * rs));
    *(u32x4_t*)(out + (long)r * D + c * 8) = o;
  }
}

constexpr int T5_MAXN = 512;

// q, k, v, out: [N][ld] bf16 with head h at columns h*64 .. h*64+63.  bucket[j - i + N - 1] is the relative-position bucket
// of key j seen from query i; bias_table [num_buckets][H] bf16.  One workgroup = 4 waves = 4 query rows of one head.
__global__ __launch_bounds__(256) void ld_t5_attn_kernel(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* out, long ld,
                                                        const bf16_t* bias_table, const int* bucket, int N, int H) {
  __shared__ float prob[4][T5_MAXN];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = blockIdx.y;
  const int i = blockIdx.x * 4 + wave;
  if (i >= N) return;
  float qv[64];
  {
    const bf16_t* qr = q + (long)i * ld + h * 64;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const u32x4_t a = *(const u32x4_t*)(qr + c * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) { qv[c * 8 + 2 * e] = bf_lo(a[e]); qv[c * 8 + 2 * e + 1] = bf_hi(a[e]); }
    }
  }
  // scores for keys lane, lane + 64, ...
  float s[T5_MAXN / 64];
  float mx = -3.0e38f;
#pragma unroll
  for (int t = 0; t < T5_MAXN / 64; ++t) {
    const int j = lane + 64 * t;
    s[t] = -3.0e38f;
    if (j < N) {
      const bf16_t* kr = k + (long)j * ld + h * 64;
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const u32x4_t a = *(const u32x4_t*)(kr + c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc += qv[c * 8 + 2 * e] * bf_lo(a[e]) + qv[c * 8 + 2 * e + 1] * bf_hi(a[e]);
      }
      const float b = bf2f(bias_table[(long)bucket[j - i + N - 1] * H + h]);
      s[t] = rbf(rbf(acc) + b);                 // bf16 matmul output, bf16 add of the position bias
      mx = fmaxf(mx, s[t]);
    }
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < T5_MAXN / 64; ++t) {
    const int j = lane + 64 * t;
    if (j < N) { s[t] = __expf(s[t] - mx); sum += s[t]; }
  }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int t = 0; t < T5_MAXN / 64; ++t) {
    const int j = lane + 64 * t;
    if (j < N) prob[wave][j] = rbf(s[t] * inv);   // softmax(fp32).type_as(bf16)
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // out[d = lane] = sum_j p_j v[j][d]
  float o = 0.f;
  const bf16_t* vc = v + h * 64 + lane;
  for (int j = 0; j < N; ++j) o += prob[wave][j] * bf2f(vc[(long)j * ld]);
  out[(long)i * ld + h * 64 + lane] = f2bf(o);
}

}  // namespace

LD_API int ld_t5_rmsnorm(const void* x, const void* w, void* out, int64_t rows, int64_t D, float eps, void* stream) {
  LD_REQUIRE(x && w && out && D % 8 == 0 && rows > 0, "ld_t5_rmsnorm: bad args");
  hipLaunchKernelGGL(ld_t5_rmsnorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)out, (int)rows, (int)D, eps);
  return ld_check_launch("ld_t5_rmsnorm");
}

LD_API int ld_t5_attn(const void* q, const void* k, const void* v, void* out, int64_t ld, const void* bias_table,
                      const int32_t* bucket, int64_t N, int64_t H, void* stream) {
  LD_REQUIRE(q && k && v && out && bias_table && bucket, "ld_t5_attn: null pointer");
  LD_REQUIRE(N > 0 && N <= T5_MAXN, "ld_t5_attn: N=%ld outside [1,%d]", (long)N, T5_MAXN);
  LD_REQUIRE(H > 0 && ld >= H * 64 && ld % 8 == 0, "ld_t5_attn: bad head count / leading dimension");
  hipLaunchKernelGGL(ld_t5_attn_kernel, dim3((unsigned)((N + 3) / 4), (unsigned)H), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out, (long)ld, (const bf16_t*)bias_table,
                     (const int*)bucket, (int)N, (int)H);
  return ld_check_launch("ld_t5_attn");
}

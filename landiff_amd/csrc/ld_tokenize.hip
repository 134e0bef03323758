// Tokenizer-encoder side kernels: feature normalisation into the channels-last GEMM layout, and the
// nearest-code search of the vector quantiser.
//
// Replaces VideoVQ.norm_features + the b t c h w -> b (t h w) c rearrange in front of TiTokEncoder.patch_embed
// (landiff/tokenizer/models/video_titok_vq.py:226-231, landiff/tokenizer/modules/blocks.py:598-600) and
// EuclideanCodebook.forward in eval (vector-quantize-pytorch 1.19.2: argmax of -cdist(x, embed), fp32).
// Both are HBM-bound and tiny next to the encoder's GEMMs; the transpose goes through a padded LDS tile so that the
// reads (contiguous in h*w) and the writes (contiguous in c) are both full 128-byte lines.
#include "ld_common.h"
#include "../../include/landiff_hip.h"

namespace {

// in [T][C][P] (fp32 or bf16) -> out [T*P][C] bf16 = bf16((x - mean[c]) / (std[c] + 1e-8)), fp32 math.
template <bool IN_F32>
__global__ __launch_bounds__(256) void ld_feature_norm_cl_kernel(const void* in, const float* mean, const float* stdv,
                                                                 bf16_t* out, int C, int P) {
  __shared__ float tile[64][65];
  const int t = blockIdx.z, c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {                      // rows = channels, columns = positions (contiguous in memory)
    const int c = c0 + r, p = p0 + tx;
    float v = 0.f;
    if (c < C && p < P) {
      const long off = ((long)t * C + c) * P + p;
      v = IN_F32 ? ((const float*)in)[off] : bf2f(((const bf16_t*)in)[off]);
      v = (v - mean[c]) / (stdv[c] + 1e-8f);
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {                      // rows = positions, columns = channels
    const int p = p0 + r, c = c0 + tx;
    if (p < P && c < C) out[((long)t * P + p) * C + c] = f2bf(tile[tx][r]);
  }
}

// x [rows][C] bf16 (channels-last decoder features) -> out bf16 = bf16(float(x) * (std[c] + 1e-8) + mean[c]): the fp32
// promotion of VideoVQ.denorm_features followed by SemanticCond's cast back to the module dtype.
__global__ __launch_bounds__(256) void ld_feature_denorm_kernel(const bf16_t* x, const float* mean, const float* stdv,
                                                                bf16_t* out, long n, int C) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % C);
  out[i] = f2bf(bf2f(x[i]) * (stdv[c] + 1e-8f) + mean[c]);
}

// One wave per row: idx[row] = first code minimising |x|^2 + |e|^2 - 2 x.e (clamped at 0), all fp32.
__global__ __launch_bounds__(256) void ld_vq_nearest_kernel(const bf16_t* x, long ldx, const float* codebook, long* idx,
                                                            int rows, int V, int dim) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float xv[64];
  float x2 = 0.f;
#pragma unroll
  for (int d = 0; d < 64; ++d) {
    xv[d] = d < dim ? bf2f(x[(long)row * ldx + d]) : 0.f;
    x2 = fmaf(xv[d], xv[d], x2);
  }
  float best = 3.0e38f;
  int besti = 0x7fffffff;
  for (int j = lane; j < V; j += 64) {
    const float* e = codebook + (long)j * dim;
    float e2 = 0.f, dot = 0.f;
    for (int d = 0; d < dim; ++d) { e2 = fmaf(e[d], e[d], e2); dot = fmaf(xv[d], e[d], dot); }
    const float d2 = fmaxf(x2 + e2 - 2.0f * dot, 0.f);
    if (d2 < best) { best = d2; besti = j; }              // ascending j: the first minimum of this lane's codes
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(besti, o, 64);
    if (ob < best || (ob == best && oi < besti)) { best = ob; besti = oi; }
  }
  if (lane == 0) idx[row] = besti;
}

}  // namespace

LD_API int ld_feature_norm_cl(const void* features, int32_t in_f32, const float* mean, const float* stdv, void* out,
                              int64_t T, int64_t C, int64_t P, void* stream) {
  LD_REQUIRE(features && mean && stdv && out && T > 0 && C > 0 && P > 0, "ld_feature_norm_cl: bad args");
  dim3 grid((unsigned)((P + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)T);
  if (in_f32) hipLaunchKernelGGL(ld_feature_norm_cl_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, features, mean, stdv,
                                 (bf16_t*)out, (int)C, (int)P);
  else hipLaunchKernelGGL(ld_feature_norm_cl_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, features, mean, stdv,
                          (bf16_t*)out, (int)C, (int)P);
  return ld_check_launch("ld_feature_norm_cl");
}

LD_API int ld_feature_denorm(const void* x, const float* mean, const float* stdv, void* out, int64_t rows, int64_t C,
                             void* stream) {
  LD_REQUIRE(x && mean && stdv && out && rows > 0 && C > 0, "ld_feature_denorm: bad args");
  const long n = (long)rows * C;
  hipLaunchKernelGGL(ld_feature_denorm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, mean, stdv, (bf16_t*)out, n, (int)C);
  return ld_check_launch("ld_feature_denorm");
}

LD_API int ld_vq_nearest(const void* x, int64_t ldx, const float* codebook, int64_t* idx, int64_t rows, int64_t V,
                         int64_t dim, void* stream) {
  LD_REQUIRE(x && codebook && idx && rows > 0 && V > 0, "ld_vq_nearest: bad args");
  LD_REQUIRE(dim >= 1 && dim <= 64 && ldx >= dim, "ld_vq_nearest: dim=%ld must be in [1,64] and <= ldx", (long)dim);
  hipLaunchKernelGGL(ld_vq_nearest_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (long)ldx, codebook, (long*)idx, (int)rows, (int)V, (int)dim);
  return ld_check_launch("ld_vq_nearest");
}

// Layout / elementwise kernels around the GEMM, attention and conv kernels (all HBM-bound streaming passes).
//
// Replaces: the patchify half of ImagePatchEmbeddingMixin.word_embedding_forward
// and `x + semantic_feature` (landiff/diffusion/dit_video_concat.py:47-62,991), unpatchify (:392-410) fused with the
// denoiser preconditioning + CFG combine (landiff/diffusion/sgm/modules/diffusionmodules/denoiser.py:25-41,
// guiders.py:75-79, sampling_utils.py:8-13), timestep_embedding (util.py:207-233), the sampler's elementwise
// updates (sampling.py:613-644,750-783), Upsample3D's nearest interpolation (vae_modules/cp_enc_dec.py:605-627),
// PixelShuffle (vq_gan_blocks.py:41-66), F.pad / halo placement (cp_enc_dec.py:467-468), and the
// [-1,1] -> uint8 post-process (dif_infer.py:37-49, landiff/utils.py:327-331).
#include "ld_common.h"
#include "../../include/landiff_hip.h"

namespace {

// x [B][T][C][H][W] f32 (+ optional sem [1][T][C][H][W] bf16 broadcast over B, added in bf16)
//   -> patches [B][T*(H/p)*(W/p)][C*p*p] bf16, K index = c*p*p + i*p + j  (Conv2d weight [D][C][p][p] flattened)
__global__ void ld_patchify_kernel(const float* x, const bf16_t* sem, bf16_t* out, int B, int T, int C, int H, int W, int p) {
  const int hp = H / p, wp = W / p, K = C * p * p;
  const long total = (long)B * T * hp * wp * K;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % K);
    long r = i / K;
    const int pw = (int)(r % wp); r /= wp;
    const int ph = (int)(r % hp); r /= hp;
    const int t = (int)(r % T);
    const int b = (int)(r / T);
    const int c = k / (p * p), ij = k % (p * p);
    const int hh = ph * p + ij / p, ww = pw * p + ij % p;
    const long src = (((long)(b * T + t) * C + c) * H + hh) * W + ww;
    float v = rbf(x[src]);
    if (sem) v = rbf(v + bf2f(sem[(((long)t * C + c) * H + hh) * W + ww]));
    out[i] = f2bf(v);
  }
}

// lin [2][T*hp*wp][C*p*p] bf16 (rows 0: uncond, 1: cond), x [1][T][C][H][W] f32
//   den_b = lin_b * c_out + x * c_skip ; out = den_u + scale * (den_c - den_u)           (fp32, unfused ops)
__global__ void ld_unpatchify_cfg_kernel(const bf16_t* lin, const float* x, float* out, int T, int C, int H, int W, int p,
                                         float c_out, float c_skip, float scale) {
  const int hp = H / p, wp = W / p, K = C * p * p;
  const long n_img = (long)T * hp * wp;
  const long total = (long)T * C * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ww = (int)(i % W);
    long r = i / W;
    const int hh = (int)(r % H); r /= H;
    const int c = (int)(r % C);
    const int t = (int)(r / C);
    const long row = ((long)t * hp + hh / p) * wp + ww / p;
    const int k = c * p * p + (hh % p) * p + (ww % p);
    const float eu = bf2f(lin[row * K + k]), ec = bf2f(lin[(n_img + row) * K + k]);
    const float xs = __fmul_rn(x[i], c_skip);
    const float du = __fadd_rn(__fmul_rn(eu, c_out), xs);
    const float dc = __fadd_rn(__fmul_rn(ec, c_out), xs);
    out[i] = __fadd_rn(du, __fmul_rn(scale, __fsub_rn(dc, du)));
  }
}

// out = a*x + b*y + c*z evaluated left to right with separately rounded products (matches eager torch)
__global__ void ld_axpbypcz_kernel(float* out, const float* x, float a, const float* y, float b, const float* z, float c, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = __fmul_rn(a, x[i]);
    if (y) v = __fadd_rn(v, __fmul_rn(b, y[i]));
    if (z) v = __fadd_rn(v, __fmul_rn(c, z[i]));
    out[i] = v;
  }
}

// timestep_embedding: t [B] f32 -> [B][dim] bf16 (cos || sin, fp32 math then cast)
__global__ void ld_timestep_embedding_kernel(const float* t, bf16_t* out, int B, int dim, float max_period) {
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  const int b = i / half, j = i % half;
  const float freq = expf(-logf(max_period) * (float)j / (float)half);
  const float arg = t[b] * freq;
  out[(long)b * dim + j] = f2bf(cosf(arg));
  out[(long)b * dim + half + j] = f2bf(sinf(arg));
  if ((dim & 1) && j == 0) out[(long)b * dim + dim - 1] = 0;
}

// generic channels-last placement:  in [F][Ti][Hi][Wi][Cin]  ->  out [F][To + tpad][Ho + 2][Wo + 2][Cout] interior
//   mode 0: copy (Ho = Hi, Wo = Wi, To = Ti), channels zero-padded Cin -> Cout
//   mode 1: nearest x2 in space; time x2 per Upsample3D's rule when time_up (first frame kept single if Ti odd)
//   mode 2: PixelShuffle(2): Cin = 4*Cout, out[2h+i][2w+j][c] = in[h][w][c*4 + i*2 + j]
struct PlaceParams {
  const bf16_t* in; bf16_t* out;
  int F, Ti, Hi, Wi, Cin, To, Ho, Wo, Cout, tpad, hpad, wpad, mode, time_up;
};

__global__ void ld_place_kernel(PlaceParams p) {
  const int chunks = p.Cout >> 3;
  const long total = (long)p.F * p.To * p.Ho * p.Wo * chunks;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % chunks);
    long r = i / chunks;
    const int w = (int)(r % p.Wo); r /= p.Wo;
    const int h = (int)(r % p.Ho); r /= p.Ho;
    const int t = (int)(r % p.To);
    const int f = (int)(r / p.To);
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (p.mode == 0) {
      if (ch * 8 < p.Cin) {
        const bf16_t* src = p.in + ((((long)f * p.Ti + t) * p.Hi + h) * p.Wi + w) * p.Cin + ch * 8;
        if (ch * 8 + 8 <= p.Cin) v = *(const u32x4_t*)src;
        else {
          bf16_t tmp[8];
          for (int e = 0; e < 8; ++e) tmp[e] = (ch * 8 + e < p.Cin) ? src[e] : (bf16_t)0;
          for (int e = 0; e < 4; ++e) v[e] = (uint32_t)tmp[2 * e] | ((uint32_t)tmp[2 * e + 1] << 16);
        }
      }
    } else if (p.mode == 1) {
      int ts = t;
      if (p.time_up && p.Ti > 1) ts = (p.Ti & 1) ? (t == 0 ? 0 : 1 + (t - 1) / 2) : t / 2;
      v = *(const u32x4_t*)(p.in + ((((long)f * p.Ti + ts) * p.Hi + (h >> 1)) * p.Wi + (w >> 1)) * p.Cin + ch * 8);
    } else {
      const bf16_t* src = p.in + ((((long)f * p.Ti + t) * p.Hi + (h >> 1)) * p.Wi + (w >> 1)) * p.Cin;
      const int sub = (h & 1) * 2 + (w & 1);
      bf16_t tmp[8];
      for (int e = 0; e < 8; ++e) tmp[e] = src[(ch * 8 + e) * 4 + sub];
      for (int e = 0; e < 4; ++e) v[e] = (uint32_t)tmp[2 * e] | ((uint32_t)tmp[2 * e + 1] << 16);
    }
    const long Tp = p.To + p.tpad, Hp = p.Ho + 2 * p.hpad, Wp = p.Wo + 2 * p.wpad;
    *(u32x4_t*)(p.out + ((((long)f * Tp + t + p.tpad) * Hp + h + p.hpad) * Wp + w + p.wpad) * p.Cout + ch * 8) = v;
  }
}

// VAE output [P][3] bf16 in [-1,1] -> uint8 [P][3]: ((v + 1) / 2) clamp [0,1], * 255, clip, truncate.
// Also optionally the float video [3][P] (CogOutput.video layout, fp32 in [0,1]).
__global__ void ld_to_uint8_kernel(const bf16_t* x, int ldx, uint8_t* out, float* video, long P) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < P * 3; i += (long)gridDim.x * blockDim.x) {
    const long pos = i / 3;
    const int c = (int)(i % 3);
    float v = bf2f(x[pos * ldx + c]);
    v = __fdiv_rn(__fadd_rn(v, 1.0f), 2.0f);
    v = fminf(fmaxf(v, 0.0f), 1.0f);
    if (video) video[(long)c * P + pos] = v;
    float u = __fmul_rn(v, 255.0f);
    u = fminf(fmaxf(u, 0.0f), 255.0f);
    out[i] = (uint8_t)u;
  }
}

// f32 -> bf16 with an optional scale (latent / scale_factor) and layout change [C][T][H][W] or [T][C][H][W] -> [T][H][W][Cpad]
__global__ void ld_latent_to_cl_kernel(const float* x, bf16_t* out, int T, int C, int H, int W, int Cpad, float mul, int src_tchw) {
  const long total = (long)T * H * W * Cpad;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cpad);
    long r = i / Cpad;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const int t = (int)(r / H);
    float v = 0.f;
    if (c < C) {
      const long src = src_tchw ? ((((long)t * C + c) * H + h) * W + w) : ((((long)c * T + t) * H + h) * W + w);
      v = rbf(rbf(x[src]) * mul);     // samples.to(bf16) then `1/scale_factor * latent` in bf16
    }
    out[i] = f2bf(v);
  }
}

inline dim3 grid_for(long total, int block = 256) {
  long b = (total + block - 1) / block;
  return dim3((unsigned)(b < 16384 ? (b > 0 ? b : 1) : 16384));
}

}  // namespace

LD_API int ld_patchify(const float* x, const void* sem, void* out, int64_t B, int64_t T, int64_t C, int64_t H, int64_t W,
                       int64_t p, void* stream) {
  LD_REQUIRE(x && out && H % p == 0 && W % p == 0, "ld_patchify: bad args");
  const long total = B * T * C * H * W;
  hipLaunchKernelGGL(ld_patchify_kernel, grid_for(total), dim3(256), 0, (hipStream_t)stream, x, (const bf16_t*)sem,
                     (bf16_t*)out, (int)B, (int)T, (int)C, (int)H, (int)W, (int)p);
  return ld_check_launch("ld_patchify");
}

LD_API int ld_unpatchify_cfg(const void* lin, const float* x, float* out, int64_t T, int64_t C, int64_t H, int64_t W,
                             int64_t p, float c_out, float c_skip, float scale, void* stream) {
  LD_REQUIRE(lin && x && out, "ld_unpatchify_cfg: null pointer");
  hipLaunchKernelGGL(ld_unpatchify_cfg_kernel, grid_for(T * C * H * W), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)lin, x, out, (int)T, (int)C, (int)H, (int)W, (int)p, c_out, c_skip, scale);
  return ld_check_launch("ld_unpatchify_cfg");
}

LD_API int ld_axpbypcz(float* out, const float* x, float a, const float* y, float b, const float* z, float c, int64_t n,
                       void* stream) {
  LD_REQUIRE(out && x && n > 0, "ld_axpbypcz: bad args");
  hipLaunchKernelGGL(ld_axpbypcz_kernel, grid_for(n), dim3(256), 0, (hipStream_t)stream, out, x, a, y, b, z, c, (long)n);
  return ld_check_launch("ld_axpbypcz");
}

LD_API int ld_timestep_embedding(const float* t, void* out, int64_t B, int64_t dim, float max_period, void* stream) {
  LD_REQUIRE(t && out && dim >= 2, "ld_timestep_embedding: bad args");
  hipLaunchKernelGGL(ld_timestep_embedding_kernel, dim3((unsigned)((B * (dim / 2) + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, t, (bf16_t*)out, (int)B, (int)dim, max_period);
  return ld_check_launch("ld_timestep_embedding");
}

LD_API int ld_place_cl(const void* in, void* out_padded, int64_t F, int64_t Ti, int64_t Hi, int64_t Wi, int64_t Cin,
                       int64_t Cout, int32_t mode, int32_t time_up, int64_t tpad, int64_t hpad, int64_t wpad, void* stream) {
  LD_REQUIRE(in && out_padded, "ld_place_cl: null pointer");
  LD_REQUIRE(Cout % 8 == 0, "ld_place_cl: Cout must be a multiple of 8");
  LD_REQUIRE(mode >= 0 && mode <= 2, "ld_place_cl: bad mode");
  LD_REQUIRE(mode != 1 || (Cin == Cout), "ld_place_cl: upsample keeps channels");
  LD_REQUIRE(mode != 2 || (Cin == 4 * Cout), "ld_place_cl: pixel shuffle needs Cin = 4*Cout");
  LD_REQUIRE(mode != 0 || Cin <= Cout, "ld_place_cl: copy cannot drop channels");
  PlaceParams p{};
  p.in = (const bf16_t*)in; p.out = (bf16_t*)out_padded;
  p.F = (int)F; p.Ti = (int)Ti; p.Hi = (int)Hi; p.Wi = (int)Wi; p.Cin = (int)Cin; p.Cout = (int)Cout;
  p.mode = mode; p.time_up = time_up; p.tpad = (int)tpad; p.hpad = (int)hpad; p.wpad = (int)wpad;
  p.To = (int)Ti; p.Ho = (int)Hi; p.Wo = (int)Wi;
  if (mode == 1) {
    p.Ho = 2 * p.Hi; p.Wo = 2 * p.Wi;
    if (time_up && Ti > 1) p.To = (Ti & 1) ? (int)(1 + 2 * (Ti - 1)) : (int)(2 * Ti);
  } else if (mode == 2) {
    p.Ho = 2 * p.Hi; p.Wo = 2 * p.Wi;
  }
  const long total = (long)p.F * p.To * p.Ho * p.Wo * (p.Cout / 8);
  hipLaunchKernelGGL(ld_place_kernel, grid_for(total), dim3(256), 0, (hipStream_t)stream, p);
  return ld_check_launch("ld_place_cl");
}

LD_API int ld_to_uint8(const void* x, int64_t ldx, uint8_t* out, float* video, int64_t P, void* stream) {
  LD_REQUIRE(x && out && P > 0, "ld_to_uint8: bad args");
  hipLaunchKernelGGL(ld_to_uint8_kernel, grid_for(P * 3), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (int)ldx,
                     out, video, (long)P);
  return ld_check_launch("ld_to_uint8");
}

LD_API int ld_latent_to_cl(const float* x, void* out, int64_t T, int64_t C, int64_t H, int64_t W, int64_t Cpad, float mul,
                           int32_t src_tchw, void* stream) {
  LD_REQUIRE(x && out && Cpad >= C, "ld_latent_to_cl: bad args");
  hipLaunchKernelGGL(ld_latent_to_cl_kernel, grid_for(T * H * W * Cpad), dim3(256), 0, (hipStream_t)stream, x,
                     (bf16_t*)out, (int)T, (int)C, (int)H, (int)W, (int)Cpad, mul, src_tchw);
  return ld_check_launch("ld_latent_to_cl");
}

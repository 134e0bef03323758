/* liblandiff_hip.so -- C ABI of the MI355X (gfx950) LanDiff inference kernels.
 *
 * The reference (LanDiff/LanDiff) has no FFI: its operator seams are Python methods that take
 * torch tensors.  Each entry point below names the reference op site it
 * replaces (paths relative to the reference repo root).  Conventions:
 *   - raw device pointers (tensor.data_ptr()), shapes/strides as int64_t in ELEMENTS,
 *   - the caller owns every buffer including workspaces; the library never allocates,
 *   - `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream); all work is
 *     asynchronous on it, no internal synchronisation, re-entrant per stream,
 *   - return 0 on success, negative on error; ld_last_error() gives the thread-local message,
 *   - bf16 tensors are raw uint16 bit patterns; "f32" means IEEE float.
 */
#ifndef LANDIFF_HIP_H
#define LANDIFF_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LD_ABI_VERSION 1

int ld_version(void);
const char* ld_last_error(void);

/* activation codes for ld_epilogue_t.act */
#define LD_ACT_NONE_ 0
#define LD_ACT_GELU_TANH_ 1
#define LD_ACT_GELU_ERF_ 2
#define LD_ACT_SILU_ 3
#define LD_ACT_TANH_ 4

/* Fused GEMM/conv epilogue, applied in this order on the fp32 accumulator x of out[m][n]:
 *   x += bias[n]; x = bf16(x); x = bf16(act(x)); x = bf16(x * mul[m][n]);
 *   x = bf16(x * gate[b(m)][region(m)][n]); x = resid[m][n] + x; x = x + add2[m][n]
 * (null pointers skip a step; the last two round to bf16 unless out_f32).
 * b(m) = m / rows_per_batch, region = text if (m % rows_per_batch) < text_len else image,
 * gate row = gate + b*gate_bstride + (gate_off_txt | gate_off_img).
 * Mirrors: bias+GELU-tanh of sat's MLP, `h + gate * y` of AdaLNMixin.layer_forward
 * (landiff/diffusion/dit_video_concat.py:593-598,619-624), the control add (:1357-1370),
 * `x + attn` / `x + mlp` of ResidualAttentionBlock (landiff/tokenizer/modules/blocks.py:292-304),
 * `x + h` of the VAE/upsampler ResnetBlocks (cp_enc_dec.py:782, vq_gan_blocks.py:148). */
typedef struct ld_epilogue_t {
  const void* bias;    /* bf16 [N] */
  int32_t act;
  const void* mul;     /* bf16 [M][ldmul] */
  int64_t ldmul;
  const void* resid;   /* bf16 or f32 [M][ldr] */
  int64_t ldr;
  int32_t resid_f32;
  const void* gate;    /* bf16 */
  int64_t gate_bstride, gate_off_img, gate_off_txt;
  int32_t rows_per_batch, text_len;
  const void* add2;    /* bf16 [M][ldadd] */
  int64_t ldadd;
  int32_t out_f32;
} ld_epilogue_t;

/* out[M][N] = epi(A[M][K] @ W[N][K]^T), bf16 operands, fp32 MFMA accumulation.
 * K % 64 == 0, lda % 8 == 0, 16-byte aligned pointers.
 * Replaces sat ColumnParallelLinear/RowParallelLinear (DiT qkv/dense/mlp), nn.Linear in the
 * TiTok decoder, the control zero-linears (dit_video_concat.py:1234-1237), 1x1x1 convs. */
int ld_gemm_bf16(const void* A, int64_t lda, const void* W, void* out, int64_t ldo,
                 int64_t M, int64_t N, int64_t K, const ld_epilogue_t* epi, void* stream);

/* Channels-last implicit-GEMM convolution, stride 1.
 * in_padded: bf16 [T+kT-1][H+kH-1][W+kW-1][Cin] with the zero spatial border and the causal time
 * halo already in place (the producer kernels write that layout); Wt: bf16 [Cout][kT][kH][kW][Cin];
 * out: [T*H*W][ldo].  Cin % 64 == 0.
 * Replaces ContextParallelCausalConv3d.forward (landiff/diffusion/vae_modules/cp_enc_dec.py:416-473)
 * and the Conv2d sites of Upsample3D (:605-633) and vq_gan_blocks (ResnetBlock :90-148). */
int ld_conv_cl_bf16(const void* in_padded, const void* Wt, void* out, int64_t ldo,
                    int64_t T, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                    int64_t kT, int64_t kH, int64_t kW, const ld_epilogue_t* epi, void* stream);

/* Fused attention, head_dim 64: O = softmax(scale * Q K^T [+ frame mask]) V.
 * Q, K: bf16 [B*H][Npad][64]; Vt: bf16 [B*H][64][Npad] (V transposed, keys contiguous);
 * O: bf16, element (b, n, h*64 + d) at O + b*o_batch_stride + n*o_row_stride + h*64 + d.
 * Npad % 128 == 0; rows/keys in [N, Npad) must be zero-filled.  fid_q/fid_k (int32 [Npad], or
 * both NULL) select the frame-block mask allowed(q,kv) <=> fid_k[kv] <= fid_q[q]; padding keys
 * carry INT32_MAX; kt_min/kt_max are the per-64-key-tile min/max of fid_k.
 * Replaces sat attention_fn_default/F.scaled_dot_product_attention behind AdaLNMixin.attention_fn
 * (landiff/diffusion/dit_video_concat.py:636-664) and flex_attention with VideoDecoderMask
 * (landiff/tokenizer/modules/blocks.py:172-212, flex_attention_mask.py:193-335). */
int ld_attn_fwd_bf16(const void* Q, const void* K, const void* Vt, void* O,
                     int64_t B, int64_t H, int64_t Nq, int64_t Nk, int64_t Npad,
                     int64_t o_batch_stride, int64_t o_row_stride, float softmax_scale,
                     const int32_t* fid_q, const int32_t* fid_k,
                     const int32_t* kt_min, const int32_t* kt_max, void* stream);

#ifdef __cplusplus
}
#endif
#endif

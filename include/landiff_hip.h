/* liblandiff_hip.so -- C ABI of the MI355X (gfx950) LanDiff inference kernels.
 *
 * The reference (LanDiff/LanDiff) has no FFI: its operator seams are Python methods that take
 * torch tensors.  Each entry point below names the reference op site it
 * replaces (paths relative to the reference repo root).  Conventions:
 *   - raw device pointers (tensor.data_ptr()), shapes/strides as int64_t in ELEMENTS,
 *   - the caller owns every buffer including workspaces; the library never allocates,
 *   - `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream); all work is
 *     asynchronous on it, no internal synchronisation, re-entrant per stream,
 *   - the library keeps ONE piece of state: the work-queue counters of the dynamic attention launch (64 sets of
 *     64 bytes in static device memory, one copy per device; see ld_attn_fwd_bf16 and ld_reset).  A set belongs to one
 *     (device, stream) pair and is zeroed on the launch stream before every use, so launches on different streams never
 *     share one and a failed launch leaves nothing behind; everything else is stateless,
 *   - return 0 on success, negative on error; ld_last_error() gives the thread-local message,
 *   - bf16 tensors are raw uint16 bit patterns; "f32" means IEEE float.
 */
#ifndef LANDIFF_HIP_H
#define LANDIFF_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a signature changes or an entry point is added / removed (2: ld_groupnorm_stats took its `partials`
 * argument, ld_gemm_qkv_heads / ld_llm_sample_advance / ld_groupnorm_stats_blocks / ld_attn_last_kernel were added).  A caller
 * compares ld_version() with the LD_ABI_VERSION it was built against before anything else (landiff_amd/_lib.py does).
 * 6: ld_reset and ld_attn_queue_poke were added.  7: ld_conv_cl_bf16_gn, ld_conv_gn_partials_size and
 * ld_groupnorm_stats_from_conv were added.  8: ld_gemm_qkv_heads_mxfp8 was added.
 * 9: ld_attn_fwd_bf16_exact was added.  10: ld_attn_last_fallbacks was added. */
#define LD_ABI_VERSION 10

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

/* The DiT's qkv Linear with the head split fused into its epilogue: ld_gemm_bf16 (bias) followed by ld_qkv_split mode 0,
 * without the [M][3*heads*64] round trip through HBM.  Replaces attention.query_key_value + sat's _transpose_for_scores +
 * query/key_layernorm of AdaLNMixin.attention_fn (landiff/diffusion/dit_video_concat.py:636-653).
 * A [M = B*Ntok][K] bf16 (row stride lda), W [3*heads*64][K] (thirds q | k | v), bias [3*heads*64];
 * Q, Kh [B][heads][Npad][64] = LayerNorm(64, eps) of the bf16 Linear output per head; Vt [B][heads][64][Npad] = V transposed.
 * Rows [Ntok, Npad) of the three outputs are NOT written: zero-fill them once.  K % 64 == 0, Ntok % 8 == 0, Ntok >= 256. */
int ld_gemm_qkv_heads(const void* A, int64_t lda, const void* W, const void* bias, int64_t M, int64_t K,
                      void* Q, void* Kh, void* Vt, int64_t B, int64_t Ntok, int64_t heads, int64_t Npad,
                      const void* q_w, const void* q_b, const void* k_w, const void* k_b, float eps, void* stream);

/* Channels-last implicit-GEMM convolution, stride 1.
 * in_padded: bf16 [T+kT-1][H+kH-1][W+kW-1][Cin] with the zero spatial border and the causal time
 * halo already in place (the producer kernels write that layout); Wt: bf16 [Cout][kT][kH][kW][Cin];
 * out: [T*H*W][ldo].  Cin % 64 == 0.
 * Replaces ContextParallelCausalConv3d.forward (landiff/diffusion/vae_modules/cp_enc_dec.py:416-473)
 * and the Conv2d sites of Upsample3D (:605-633) and vq_gan_blocks (ResnetBlock :90-148). */
int ld_conv_cl_bf16(const void* in_padded, const void* Wt, void* out, int64_t ldo,
                    int64_t T, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                    int64_t kT, int64_t kH, int64_t kW, const ld_epilogue_t* epi, void* stream);

/* ld_conv_cl_bf16 whose epilogue also leaves GroupNorm partial sums of the bf16 OUTPUT it stores: gn_partials, caller-owned,
 * ld_conv_gn_partials_size(T*H*W, Cout) floats = [ceil(T*H*W / 64)][Cout / 4][2] (sum, sum of squares of every 64-row x
 * 4-channel patch), every entry written exactly once by a fixed sequence of fp32 additions (deterministic; no atomics).
 * ld_groupnorm_stats_from_conv turns them into the statistics ld_groupnorm_stats would compute from a second read of the
 * activation -- every GroupNorm of the VAE decoder normalises a convolution's output (SpatialNorm3D, cp_enc_dec.py:546-569,
 * inside the resblock :745-782).  Needs Cout % 8 == 0, ldo % 8 == 0, bf16 output, epilogue operands with leading dimensions
 * % 8 == 0 (LD_ERR_INVALID otherwise); always the GEMM routes 0-2. */
int64_t ld_conv_gn_partials_size(int64_t M, int64_t Cout);
int ld_conv_cl_bf16_gn(const void* in_padded, const void* Wt, void* out, int64_t ldo,
                       int64_t T, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                       int64_t kT, int64_t kH, int64_t kW, const ld_epilogue_t* epi, float* gn_partials, void* stream);

/* Which kernel ld_conv_cl_bf16 runs for a shape, without launching anything (no GPU needed): 0 = 128x128 two-stage,
 * 1 = 256x256 two-stage (32-bit element offsets: padded inputs up to 8 GiB), 2 = 256x256 8-phase (one raw buffer descriptor
 * over the padded input: inputs below 2 GiB only -- larger ones are routed to 1), negative = the shape is refused
 * (e.g. a padded input of 8 GiB or more), 3 = never returned for a convolution (the register-staged 4-wave GEMM loop of the
 * variants build), 5 = the narrow-output kernel (3x3x3, Cin 128, <= 4 output channels, H % 4 == 0,
 * W % 16 == 0 and a bias-only epilogue -- the VAE's conv_out: ld_conv_narrow.hip reads the input once instead of once per tap),
 * 4 = 512x128 8-phase (VARIANTS build with LD_GEMM_M512=1 only: 64 < Cout <= 128, 2048 <= K <= 4096, at least 256 such tiles,
 * padded input below 2 GiB -- a measured alternative for the VAE's 480x720 level).  All GEMM routes (0, 1, 2, 4) give bit-identical
 * outputs.  Host logic only; lets the size guard be tested on CPU. */
int ld_conv_route(int64_t T, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t kT, int64_t kH, int64_t kW);

/* ---- calibration loops for bench.py (not on the product path; ld_calib.hip) ----
 * ld_calib_mfma_bf16: n_workgroups x 256 threads (one wave per SIMD), `iters` trips of 128 v_mfma_f32_16x16x32_bf16 per wave on
 * operands taken once from `operands` (random bf16: the matrix pipe's power depends on the operand bits); *flops_out = FLOP of
 * the launch.  ld_calib_stream_read: one pass of 16-byte non-temporal loads over `bytes`. */
int ld_calib_mfma_bf16(const void* operands, int64_t operand_bytes, float* sink, int64_t n_workgroups, int64_t iters,
                       double* flops_out, void* stream);
int ld_calib_stream_read(const void* buf, int64_t bytes, uint32_t* sink, void* stream);

/* ---- optional fp8 (OCP e4m3) form of the DiT's large linear layers (BASELINE.json configs[4]; the headline metric and
 * every parity claim of the bf16 path stay on ld_gemm_bf16) ---- */

/* Row-wise dynamic quantisation of the activations in front of a Linear: x bf16 [rows][K] (row stride ldx) ->
 * q e4m3 [rows][K] (row stride ldq bytes) and scale[r] = amax(row r) / 448 (1 for an all-zero row); q = rne(x / scale). */
int ld_quantize_fp8(const void* x, int64_t ldx, void* q, int64_t ldq, float* scale, int64_t rows, int64_t K, void* stream);

/* out[M][N] = epilogue(scale_a[m] * scale_w[n] * sum_k A8[m][k] * W8[n][k]): the same nn.Linear sites and the same
 * epilogues as ld_gemm_bf16 (dit_video_concat.py:540-629), operands e4m3 (A8 row stride lda bytes, W8 [N][K] contiguous),
 * fp32 accumulation on v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales.  K % 128 == 0. */
int ld_gemm_fp8(const void* A8, int64_t lda, const float* scale_a, const void* W8, const float* scale_w, void* out,
                int64_t ldo, int64_t M, int64_t N, int64_t K, const ld_epilogue_t* epi, void* stream);

/* MXFP8 form of the same (OCP Microscaling v1.0 container: blocks of 32 consecutive K elements share one E8M0 scale;
 * here the smallest power of two >= amax / 448, so nothing saturates; elements are e4m3 casts of x / scale).
 * scales: uint8, K-tile-major [K/128][lds >= rows][4] (byte = exponent + 127; 0 for an all-zero block): the 256 rows of
 * a GEMM tile and K-tile are 1 KB contiguous for the LDS-DMA.  K % 128 == 0. */
int ld_quantize_mxfp8(const void* x, int64_t ldx, void* q, int64_t ldq, void* scales, int64_t lds, int64_t rows,
                      int64_t K, void* stream);

/* out = epilogue(sum over blocks of 2^(sa-127) 2^(sw-127) sum_{k in block} A8[m][k] W8[n][k]): the block scales are
 * applied by the MFMA itself (v_mfma_scale_f32_16x16x128_f8f6f4 on the persistent two-phase loop since round 6).  scales_a [K/128][M][4], scales_w [K/128][N][4] contiguous. */
int ld_gemm_mxfp8(const void* A8, int64_t lda, const void* scales_a, const void* W8, const void* scales_w, void* out,
                  int64_t ldo, void* out_scales, int64_t ldos, int64_t M, int64_t N, int64_t K,
                  const ld_epilogue_t* epi, void* stream);
/* (out_scales != NULL: the output is itself MXFP8 -- out = e4m3 [M][ldo bytes], out_scales [N/128][ldos >= M][4] -- for the
 * bias + GELU-tanh epilogue of dense_h_to_4h, whose result only feeds the next MXFP8 GEMM.) */

/* ld_gemm_qkv_heads on MXFP8 operands (BASELINE configs[4]; round 6): the DiT's qkv Linear with QK-LayerNorm, head split and V
 * transpose in its epilogue, A8 [M][lda bytes] / W8 [3*heads*64][K] e4m3 codes with scales_a [K/128][M][4], scales_w
 * [K/128][3*heads*64][4].  Same outputs and shape rules as ld_gemm_qkv_heads; K % 128 == 0.  Reference sites: as ld_gemm_qkv_heads. */
int ld_gemm_qkv_heads_mxfp8(const void* A8, int64_t lda, const void* scales_a, const void* W8, const void* scales_w,
                            const void* bias, int64_t M, int64_t K, void* Q, void* Kh, void* Vt, int64_t B, int64_t Ntok,
                            int64_t heads, int64_t Npad, const void* q_w, const void* q_b, const void* k_w, const void* k_b,
                            float eps, void* stream);

/* ld_layernorm (+ AdaLN modulate) with MXFP8 output: exactly ld_layernorm's bf16 result, quantised where it is produced
 * (q e4m3 [rows][ldq bytes], scales [D/128][lds >= rows][4]) for the MXFP8 GEMM that consumes it.  bf16 input, D % 128 == 0. */
int ld_layernorm_mxfp8(const void* x, int64_t ldx, const void* w, const void* b, void* q, int64_t ldq, void* scales,
                       int64_t lds, int64_t rows, int64_t D, float eps, const void* mod, int64_t mod_bstride,
                       int64_t shift_img, int64_t scale_img, int64_t shift_txt, int64_t scale_txt,
                       int64_t rows_per_batch, int64_t text_len, void* stream);

/* Fused attention, head_dim 64: O = softmax(scale * Q K^T [+ frame mask]) V.
 * Q, K: bf16 [B*H][Npad][64]; Vt: bf16 [B*H][64][Npad] (V transposed, keys contiguous);
 * O: bf16, element (b, n, h*64 + d) at O + b*o_batch_stride + n*o_row_stride + h*64 + d.
 * Npad % 128 == 0; rows/keys in [N, Npad) must be zero-filled.  fid_q/fid_k (int32 [Npad], or
 * both NULL) select the frame-block mask allowed(q,kv) <=> fid_k[kv] <= fid_q[q]; padding keys
 * carry INT32_MAX; kt_min/kt_max are the per-64-key-tile min/max of fid_k.
 * Replaces sat attention_fn_default/F.scaled_dot_product_attention behind AdaLNMixin.attention_fn
 * (landiff/diffusion/dit_video_concat.py:636-664) and flex_attention with VideoDecoderMask
 * (landiff/tokenizer/modules/blocks.py:172-212, flex_attention_mask.py:193-335).
 * Large unmasked problems (>= 4 rounds of 2 workgroups per CU) run as a dynamic launch whose workgroups pull query blocks
 * through per-XCD counters.  The counter set is chosen by (current device, stream): the first 64 streams of a device that
 * launch attention own one each, further streams -- and any stream that is being captured into a graph -- take the static
 * launch of the same kernel body (bit-identical output).  The set is zeroed on `stream` right before the kernel. */
int ld_attn_fwd_bf16(const void* Q, const void* K, const void* Vt, void* O,
                     int64_t B, int64_t H, int64_t Nq, int64_t Nk, int64_t Npad,
                     int64_t o_batch_stride, int64_t o_row_stride, float softmax_scale,
                     const int32_t* fid_q, const int32_t* fid_k,
                     const int32_t* kt_min, const int32_t* kt_max, void* stream);

/* The same operation with an exact safe softmax for ANY logit range at a fixed cost (~1.55 x ld_attn_fwd_bf16's launch at the DiT
 * shape): for checkpoints whose q.k * softmax_scale row maxima leave the window of the default launch's max-free fast pass (beyond ~76
 * in natural-log units, i.e. denominators beyond 2^110; profiles/r06_attn_logit_sweep.txt), where the default launch stays correct
 * but re-runs every affected 256-row block (up to 2.6 x).  Unmasked problems of >= 6 key tiles: two passes over the keys (row maxima
 * from QK^T only, then the pipelined loop with -max as the score accumulators' initial value; ld_attn_q64_exact.hip); every other
 * shape: the plain online-softmax kernel ld_attn_fwd_bf16 itself uses there.  No data-dependent branch; run to run identical. */
int ld_attn_fwd_bf16_exact(const void* Q, const void* K, const void* Vt, void* O,
                           int64_t B, int64_t H, int64_t Nq, int64_t Nk, int64_t Npad,
                           int64_t o_batch_stride, int64_t o_row_stride, float softmax_scale,
                           const int32_t* fid_q, const int32_t* fid_k,
                           const int32_t* kt_min, const int32_t* kt_max, void* stream);

/* Forgets the stream -> counter-set assignments of the current device and zeroes its sets (asynchronously on `stream`).
 * Only for a process that keeps creating streams, or after a device error: the caller guarantees that no ld_attn_fwd_bf16
 * launch of this device is enqueued or running on any other stream. */
int ld_reset(void* stream);

/* Test hook: fills counter set `set` (all 64 when set < 0) of the current device with `value`, as a launch that died
 * mid-flight could leave it.  Later launches must be unaffected (tests/test_gpu_attn.py). */
int ld_attn_queue_poke(int32_t set, uint32_t value, void* stream);

/* Name of the kernel the calling thread's last ld_attn_fwd_bf16 launched ("" before the first call): the launcher picks
 * by shape and tuning environment, and measurement code must label what actually ran. */
const char* ld_attn_last_kernel(void);

/* How many 256-row query blocks of the calling thread's most recent ld_attn_fwd_bf16 launch had a row whose softmax denominator
 * left the window of the max-free fast pass (2^-80 .. 2^110) and were recomputed by the running-max pass: a 4-byte value written to
 * `out` (DEVICE memory) by a copy enqueued on `stream`, which must be the stream of that launch and must not have seen another
 * attention launch since.  0 for launches by kernels without such a window (short or masked problems, the exact form); -1 when
 * the launch has a window but kept no count (the static dispatch used under graph capture or beyond 64 streams per device).
 * A caller that runs the same layers again and again (a sampler loop) reads it to switch a layer whose checkpoint leaves the
 * window to ld_attn_fwd_bf16_exact (landiff_amd/dit.py: attn_exact="auto"; profiles/r06_attn_logit_sweep.txt for the prices). */
int ld_attn_last_fallbacks(int32_t* out, void* stream);

/* ---- tokenizer encoder side (ld_tokenize.hip) ---- */

/* VideoVQ.norm_features + the "b t c h w -> b (t h w) c" rearrange in front of TiTokEncoder.patch_embed
 * (landiff/tokenizer/models/video_titok_vq.py:226-231, landiff/tokenizer/modules/blocks.py:598-600):
 * features [T][C][P] (fp32 if in_f32 else bf16, P = h*w) -> out [T*P][C] bf16 = bf16((x - mean[c]) / (std[c] + 1e-8)). */
int ld_feature_norm_cl(const void* features, int32_t in_f32, const float* mean, const float* stdv, void* out,
                       int64_t T, int64_t C, int64_t P, void* stream);

/* VideoVQ.denorm_features (video_titok_vq.py:228-233; active only when the config names a mean_std_path) on the
 * channels-last decoder output, then the cast back to the module dtype (condition.py:103-104):
 * x bf16 [rows][C] -> out bf16 = bf16(float(x) * (std[c] + 1e-8) + mean[c]).  out may alias x. */
int ld_feature_denorm(const void* x, const float* mean, const float* stdv, void* out, int64_t rows, int64_t C, void* stream);

/* Nearest code of vector-quantize-pytorch's EuclideanCodebook.forward in eval (argmax of -cdist, fp32), called by
 * VideoVQ.encode_to_index (video_titok_vq.py:196-200): x bf16 [rows][ldx] (first dim columns), codebook fp32 [V][dim]
 * -> idx int64 [rows] = the first code minimising |x|^2 + |e|^2 - 2 x.e. */
int ld_vq_nearest(const void* x, int64_t ldx, const float* codebook, int64_t* idx, int64_t rows, int64_t V,
                  int64_t dim, void* stream);

/* ---- HBM-bound small-batch kernels (ld_llm.hip) ---- */

/* y[b][n] = epi(sum_k in_act(x[b][k]) * W[n][k]), 1 <= B <= 4, weights streamed once (one wave per row).
 * bf16 weights: result rounds to bf16 (Linear output), then act, then `* bf16(W2 x)` (gated MLP,
 * LlamaMLP2: landiff/llm/modules/transformer_blocks.py:67-88), then `resid + y`.  w_f32: fp32 weights,
 * x and out fp32 (GPT head, landiff/llm/models/transformer.py:115-118).  in_act is applied to x (bf16-rounded),
 * e.g. the SiLU in front of adaLN_modulation (landiff/diffusion/dit_video_concat.py:499-503).  norm_w (fp32 [K], optional)
 * fuses the block's RMSNorm in front of the projection: x -> bf16(x * rsqrt(mean(x^2) + norm_eps) * norm_w). */
int ld_gemv(const void* x, int64_t ldx, int32_t x_f32, const void* W, const void* W2, int32_t w_f32,
            const void* bias, const void* resid, int64_t ldr, void* out, int64_t ldo, int32_t out_f32,
            int64_t B, int64_t N, int64_t K, int32_t in_act, int32_t act, const float* norm_w, float norm_eps,
            void* stream);

/* RMSNorm (transformer_blocks.py:22-40): bf16 rows [rows][D], fp32 weight, fp32 math, bf16 out. */
int ld_rmsnorm_bf16(const void* x, const float* w, void* out, int64_t rows, int64_t D, float eps, void* stream);

/* Final LayerNorm of GPT.sample (transformer.py:112-114): bf16 rows (stride ldx) -> fp32, fp32 affine. */
int ld_layernorm_bf16_to_f32(const void* x, int64_t ldx, const float* w, const float* b, float* out,
                             int64_t rows, int64_t D, float eps, void* stream);

/* apply_rope (landiff/modules/pos_emb.py:16-46) on q,k of qkv [B][m][3][H][128] at positions *pos + j, and KV append
 * into k_cache/v_cache [B][Lmax][H][128] (replaces the torch.cat growth of transformer_blocks.py:153-164). */
int ld_llm_rope_append(const void* qkv, const float* cos_t, const float* sin_t, const int32_t* pos,
                       void* q_out, void* k_cache, void* v_cache, int64_t B, int64_t m, int64_t H,
                       int64_t Lmax, void* stream);

/* Attention of query j (position *pos + j) over keys [0, *pos + j] (transformer_blocks.py:166-186):
 * bf16 scores, bf16(score / sqrt(128)), fp32 softmax -> bf16 p, bf16 output [B][m][H][128].
 * m == 1 with nsplit > 1 uses the key-split path (workspace: B*H*(nsplit*130 + 1) 4-byte words, caller-owned; the last B*H
 * words are arrival counters that must be ZERO before the first call -- the launch leaves them zero: the last split of a
 * (batch row, head) to arrive merges the partial results, no second launch; p kept fp32); there
 * qkv_fused ([B][3][H][128], q may be NULL) makes the kernel do apply_rope and the KV append of the new token itself.
 * nsplit is the MAXIMUM number of key splits (it sizes the workspace and must keep Lmax / nsplit <= 256): a step at context length
 * L = *pos + 1 uses 1 / 2 / 4 / nsplit of them by a fixed rule of L alone (short contexts are not worth a merge; one split writes
 * the output directly), evaluated on the device, so every way of issuing the step gives the same bits. */
int ld_llm_kv_attn(const void* q, const void* k_cache, const void* v_cache, const int32_t* pos, void* out,
                   int64_t B, int64_t m, int64_t H, int64_t Lmax, float* workspace, int64_t nsplit,
                   const void* qkv_fused, const float* cos_t, const float* sin_t, void* stream);

/* One decode step of GPT.sample (landiff/llm/models/transformer.py:91-119 with the cached blocks of
 * transformer_blocks.py:128-236) for B = 2 (the CFG pair): embedding of *token, n_layers x [RMSNorm+qkv GEMV, RoPE +
 * KV append + key-split attention at position *pos, wo GEMV + residual, RMSNorm + gated-GELU MLP GEMVs + residual],
 * final LayerNorm (fp32) and the fp32 head -> logits [B][vocab].  Exactly the launches a caller would issue through
 * ld_llm_embed / ld_gemv / ld_llm_kv_attn / ld_layernorm_bf16_to_f32, queued from native code in one call: the
 * ~150 launches of a step are then bound by the GPU (~1.2 ms) and not by the host language's per-call overhead.
 * emb_table == NULL: x already holds the embedding rows of the token (written by ld_llm_sample_advance).
 * All step state (*token, *pos) is read on the device; pos_value >= 0 tells the launches the value *pos holds (a decode loop
 * knows it: one more per step) so that the attention launches do not start with a dependent load of *pos, -1: read *pos (graph
 * capture); buffers are caller-owned: x/att [B][hidden], qkv [B][3*hidden],
 * gate [B][mlp] bf16, attn_ws B*heads*(nsplit*130 + 1) words (see ld_llm_kv_attn: the last B*heads zero), lnf_out [B][hidden] fp32,
 * logits [B][vocab] fp32. */
typedef struct ld_llm_layer {
  const void* wqkv; const void* wo; const void* w1; const void* w3; const void* w2;   /* bf16 [3h][h] [h][h] [mlp][h] [mlp][h] [h][mlp] */
  const float* n0; const float* n1;                                                    /* RMSNorm gains, fp32 [h] */
  void* k_cache; void* v_cache;                                                        /* bf16 [B][Lmax][heads][128] */
} ld_llm_layer;
int ld_llm_decode_forward(const ld_llm_layer* layers, int64_t n_layers, const float* emb_table, const int64_t* token,
                          const int32_t* pos, int32_t pos_value, void* x, void* qkv, void* att, void* gate, float* attn_ws,
                          const float* cos_t, const float* sin_t, const float* lnf_w, const float* lnf_b, float* lnf_out,
                          const float* head_w, float* logits, int64_t B, int64_t hidden, int64_t heads, int64_t mlp,
                          int64_t vocab, int64_t Lmax, int64_t nsplit, float rms_eps, float ln_eps, void* stream);

/* ---- Entry points of the VARIANTS build only (landiff_amd/variants/liblandiff_hip_variants.so, built by
 * `LD_BUILD_VARIANTS=1 landiff_amd/csrc/build.sh` with -DLD_VARIANTS): two other forms of the decode step that were built, are
 * bit-identical to ld_llm_decode_forward, measured slower and are kept under test as measured alternatives.  The shipped
 * library does not export them. ---- */
#ifdef LD_VARIANTS
/* The same step with all blocks in ONE persistent launch (ld_llm_fused.hip): identical bits, ~one launch instead of 144.
 * layers_dev: the ld_llm_layer table in DEVICE memory.  ctl: LD_LLM_FUSED_CTL_WORDS 32-bit words of device memory owned by
 * the caller, zeroed before the first step of a decode (word 0: steps done, word 1: error flag -- non-zero after a grid
 * barrier timed out; every later launch then returns without touching anything; words 32..: arrival counters).  The launch
 * needs the GPU's CUs to itself (its workgroups wait for each other): do not run it concurrently with other streams' kernels.
 * LD_ERR_UNSUPPORTED (nothing launched) unless B == 2, head_dim 128, hidden <= 2048, mlp <= 12288, nsplit >= 2 and at most
 * 256 keys per split -- use ld_llm_decode_forward then.  Replaces transformer_blocks.py:128-236 / transformer.py:91-119. */
#define LD_LLM_FUSED_CTL_WORDS 512
int ld_llm_decode_blocks_fused(const ld_llm_layer* layers_dev, int64_t n_layers, const int32_t* pos, void* x, void* qkv, void* att,
                               void* gate, float* attn_ws, const float* cos_t, const float* sin_t, int64_t B, int64_t hidden,
                               int64_t heads, int64_t mlp, int64_t Lmax, int64_t nsplit, float rms_eps, uint32_t* ctl, void* stream);
/* embedding (optional) -> ld_llm_decode_blocks_fused -> final LayerNorm -> fp32 head: the drop-in form of ld_llm_decode_forward */
int ld_llm_decode_forward_fused(const ld_llm_layer* layers_dev, int64_t n_layers, const float* emb_table, const int64_t* token,
                                const int32_t* pos, void* x, void* qkv, void* att, void* gate, float* attn_ws,
                                const float* cos_t, const float* sin_t, const float* lnf_w, const float* lnf_b, float* lnf_out,
                                const float* head_w, float* logits, int64_t B, int64_t hidden, int64_t heads, int64_t mlp,
                                int64_t vocab, int64_t Lmax, int64_t nsplit, float rms_eps, float ln_eps, uint32_t* ctl,
                                void* stream);

/* The blocks of the same step as DEPENDENT LAUNCHES ON TWO STREAMS: operation k (5 per block: qkv, attention, wo, w1.w3, w2) goes
 * to stream k & 1 with no stream dependency on operation k - 1; its workgroups request their first weight rows (the attention: its
 * K / V rows), then wait until every workgroup of operation k - 1 has arrived on that operation's counters, and arrive on their own
 * when their outputs are written (sc1) and drained -- a launch is dispatched and streaming weights while its predecessor runs.
 * Identical bits.  layers: HOST table (as ld_llm_decode_forward).  pos_value: the position *pos holds (the host knows it: one
 * more per step) -- the launches do not read *pos, so they do not depend on the previous step's sampling launch.  ctl:
 * LD_LLM_CHAIN_CTL_WORDS words of device memory, zeroed before the first step of a decode (word 0: error flag, set when a wait
 * timed out after ~1 s -- later waits then fall through); epoch: steps done since the zeroing.  Caller's duties: stream1 must
 * be ordered after whatever wrote the KV cache / x before the first step (e.g. the prefill), and whatever reads x after the call
 * must be ordered after BOTH streams.  Every launch fits twice on the chip (<= 128 registers, <= 512 workgroups; the attention
 * half the chip), so the waiting launch can never keep its predecessor from becoming resident -- provided nothing else occupies
 * the GPU for long.  LD_ERR_UNSUPPORTED outside B == 2, head_dim 128, hidden <= 4096, mlp <= 12288, <= 256 keys per split,
 * 5 * n_layers <= LD_LLM_CHAIN_MAX_OPS. */
#define LD_LLM_CHAIN_MAX_OPS 256
#define LD_LLM_CHAIN_CTL_WORDS (64 + LD_LLM_CHAIN_MAX_OPS * 8 * 16)
int ld_llm_decode_blocks_chained(const ld_llm_layer* layers, int64_t n_layers, int32_t pos_value, void* x, void* qkv, void* att,
                                 void* gate, float* attn_ws, const float* cos_t, const float* sin_t, int64_t B, int64_t hidden,
                                 int64_t heads, int64_t mlp, int64_t Lmax, int64_t nsplit, float rms_eps, uint32_t* ctl,
                                 uint32_t epoch, void* stream0, void* stream1);
#endif /* LD_VARIANTS */

/* nn.Embedding lookup of *token (fp32 table [V][D]) -> bf16 features [B][D] (landiff/llm/modules/tokenizer.py:10-55). */
int ld_llm_embed(const float* table, const int64_t* token, void* out, int64_t B, int64_t D, void* stream);

/* lm_model.py:417-454: logits [2][V] = (cond, uncond) -> CFG `u + s*(c-u)` (guided), / temperature, optional
 * restriction to allowed[*pos + 1][1 .. 1+allowed[..][0]], softmax -> probs [V] (V <= 4096).  cfg_logits (optional) gets
 * the CFG logits.  At unrestricted positions: top_k > 0 keeps logits >= the k-th largest (lm_model.py:441-443), and
 * top_p >= 0 is top_p_probability (landiff/utils.py:345-359: descending sort, sequential cumsum, drop sorted position
 * j >= 1 when cumsum[j-1] >= top_p, renormalise).  top_k <= 0 / top_p < 0 switch the filters off (the CLI default). */
int ld_llm_logits_to_probs(const float* logits, float* probs, float* cfg_logits, int64_t V, int32_t guided,
                           float scale, float temperature, const int32_t* pos, const int32_t* allowed,
                           int64_t allowed_stride, int32_t top_k, float top_p, void* stream);

/* ld_llm_logits_to_probs, the draw, ld_llm_decode_advance and ld_llm_embed of the next token in one launch (the tail of a
 * decode step: lm_model.py:417-508).  `noise` [V]: Exp(1) draws from the caller's generator -- torch.multinomial(p, 1, gen)
 * IS argmax(p / empty_like(p).exponential_(1, gen)) (lowest index on ties), so the token equals the reference's draw from
 * the same generator state.  probs / cfg_logits / sampled (the raw draw, before the forced-token override) are optional
 * outputs; x [B][D] bf16 receives the embedding rows of *token (ld_llm_decode_forward then takes emb_table = NULL). */
int ld_llm_sample_advance(const float* logits, float* probs, float* cfg_logits, int64_t V, int32_t guided, float scale,
                          float temperature, int32_t* pos, const int32_t* allowed, int64_t allowed_stride,
                          int32_t top_k, float top_p, const float* noise, const int32_t* forced, int64_t* token,
                          int64_t* out_tokens, int32_t* out_count, int64_t* sampled, const float* emb_table, void* x,
                          int64_t B, int64_t D, void* stream);

/* After torch.multinomial: forced-token override (forced[*pos + 1] >= 0), record sampled visual tokens, ++*pos
 * (the elif chain of lm_model.py:455-508). */
int ld_llm_decode_advance(const int64_t* sampled, const int32_t* forced, int32_t* pos, int64_t* token,
                          int64_t* out_tokens, int32_t* out_count, void* stream);

/* ---- normalisation kernels (ld_norm.hip) ---- */

/* LayerNorm over D (fp32 statistics) with optional AdaLN modulate `x*(1+scale)+shift` (bf16 ops) whose vectors are
 * picked per row: batch = row / rows_per_batch, region = text if (row % rows_per_batch) < text_len else image;
 * vector = mod + batch*mod_bstride + {shift,scale}_{img,txt}.  (dit_video_concat.py:388,577-586,601-611;
 * also nn.LayerNorm in TiTok blocks.py:292-304 with fp32 in / bf16 out.) */
int ld_layernorm(const void* x, int64_t ldx, int32_t x_f32, const void* w, const void* b, void* out, int64_t ldo,
                 int32_t out_f32, int64_t rows, int64_t D, float eps, const void* mod, int64_t mod_bstride,
                 int64_t shift_img, int64_t scale_img, int64_t shift_txt, int64_t scale_txt,
                 int64_t rows_per_batch, int64_t text_len, void* stream);

/* qkv [B*N][3*H*64] (thirds q|k|v, sat SelfAttention layout) -> Q,K [B][H][Npad][64] (zero padded) and
 * V^T [B][H][64][Npad].  mode 0: LayerNorm(64, eps) on q,k heads (dit_video_concat.py:649-653);
 * mode 1: interleaved-pair RoPE with cos/sin [N][32] (blocks.py:172-180, pos_emb.py:16-46). */
int ld_qkv_split(const void* qkv, void* Q, void* K, void* Vt, int64_t B, int64_t N, int64_t H, int64_t Npad,
                 int32_t mode, const void* q_w, const void* q_b, const void* k_w, const void* k_b, float eps,
                 const float* cos_t, const float* sin_t, void* stream);

/* GroupNorm statistics of channels-last x [F][P][C] (torch.nn.GroupNorm inside Normalize, vq_gan_blocks.py:35-38, and
 * SpatialNorm3D, cp_enc_dec.py:546-569): stats[f][g] = (sum, sum of squares) in double.  No floating-point atomics:
 * partials is a caller-owned workspace of F * ld_groupnorm_stats_blocks(P) * G * 2 doubles, reduced in a fixed order, so
 * the result is bit-identical from run to run.  stats need not be zeroed. */
int64_t ld_groupnorm_stats_blocks(int64_t P);
int ld_groupnorm_stats(const void* x, double* stats, double* partials, int64_t F, int64_t P, int64_t C, int64_t G, void* stream);

/* The same stats[g] (F = 1) from the partial sums ld_conv_cl_bf16_gn left for an output of P rows x C channels: a fold of the
 * 64-row units into at most 256 double pairs per group (partials: caller-owned workspace of 256 * G * 2 doubles), then the same
 * fixed-order reduce.  The sums are those of the same bf16 values, added in another order (fp32 within a 64 x 4 patch, double
 * above): statistics agree with ld_groupnorm_stats to ~1e-7 relative, not bit for bit.  C / 4 a power of two <= 256, G <= 64,
 * whole quads per group. */
int ld_groupnorm_stats_from_conv(const float* gn_partials, double* stats, double* partials, int64_t P, int64_t C, int64_t G,
                                 void* stream);

/* y = swish?( GN(x) [* zy + zb] ) written into the interior of a zero-bordered channels-last buffer
 * [F][T+tpad][H+2*hpad][W+2*wpad][C]; zy/zb [Tz][Hz][Wz][C] are conv_y(zq)/conv_b(zq) at latent resolution, gathered
 * with SpatialNorm3D's nearest rule incl. the first-frame split for odd T (cp_enc_dec.py:546-569); GroupNorm(32,
 * eps 1e-6) + swish of vq_gan_blocks.py:29-38,90-148 when zy is NULL. */
int ld_groupnorm_apply(const void* x, void* out_padded, const double* stats, const void* gamma, const void* beta,
                       const void* zy, const void* zb, int64_t F, int64_t T, int64_t H, int64_t W, int64_t C,
                       int64_t G, int64_t Tz, int64_t Hz, int64_t Wz, int64_t tpad, int64_t hpad, int64_t wpad,
                       int32_t swish, float eps, void* stream);

/* ---- layout / elementwise kernels (ld_elem.hip) ---- */

/* x [B][T][C][H][W] f32 (-> bf16, + sem [T][C][H][W] bf16 for the control net, dit_video_concat.py:991)
 * -> patch rows [B*T*(H/p)*(W/p)][C*p*p] bf16 for the patch-embed GEMM (:47-62). */
int ld_patchify(const float* x, const void* sem, void* out, int64_t B, int64_t T, int64_t C, int64_t H, int64_t W,
                int64_t p, void* stream);

/* unpatchify (:392-410) + Denoiser.forward `net*c_out + x*c_skip` (denoiser.py:25-41) + CFG `u + s*(c-u)`
 * (guiders.py:75-79): lin [2][T*hp*wp][C*p*p] bf16 (uncond, cond), x/out [T][C][H][W] f32. */
int ld_unpatchify_cfg(const void* lin, const float* x, float* out, int64_t T, int64_t C, int64_t H, int64_t W,
                      int64_t p, float c_out, float c_skip, float scale, void* stream);

/* out = a*x + b*y + c*z (y, z optional), products rounded separately, summed left to right (sampling.py:613-644,771-781). */
int ld_axpbypcz(float* out, const float* x, float a, const float* y, float b, const float* z, float c, int64_t n, void* stream);

/* timestep_embedding (sgm/modules/diffusionmodules/util.py:207-233): t [B] f32 -> [B][dim] bf16 (cos || sin). */
int ld_timestep_embedding(const float* t, void* out, int64_t B, int64_t dim, float max_period, void* stream);

/* Place a channels-last tensor in [F][Ti][Hi][Wi][Cin] into the interior of a zero-bordered buffer
 * [F][To+tpad][Ho+2*hpad][Wo+2*wpad][Cout].  mode 0 copy (+channel zero pad); mode 1 nearest x2 (time too when
 * time_up, first frame single for odd Ti: Upsample3D, cp_enc_dec.py:605-627); mode 2 PixelShuffle(2). */
int ld_place_cl(const void* in, void* out_padded, int64_t F, int64_t Ti, int64_t Hi, int64_t Wi, int64_t Cin,
                int64_t Cout, int32_t mode, int32_t time_up, int64_t tpad, int64_t hpad, int64_t wpad, void* stream);

/* dif_infer.py:37-49 + landiff/utils.py:327-331: x [P][ldx] bf16 (3 channels) -> uint8 [P][3] (truncation) and,
 * optionally, the fp32 video [3][P] in [0,1]. */
int ld_to_uint8(const void* x, int64_t ldx, uint8_t* out, float* video, int64_t P, void* stream);

/* f32 latent ([T][C][H][W] if src_tchw else [C][T][H][W]) -> bf16 channels-last [T][H][W][Cpad], value bf16(bf16(x)*mul)
 * (dif_infer.py:251 `1/scale_factor * latent` on the bf16 samples). */
int ld_latent_to_cl(const float* x, void* out, int64_t T, int64_t C, int64_t H, int64_t W, int64_t Cpad, float mul,
                    int32_t src_tchw, void* stream);

/* ---- T5 text encoders (HF transformers T5EncoderModel called from
 * landiff/llm/modules/text_encoder.py:36-42,82-112 and landiff/diffusion/sgm/modules/encoders/modules.py:249-292) ---- */

/* T5LayerNorm (transformers modeling_t5.py): fp32 variance, bf16(x*rsqrt) then bf16(weight * that); bf16 weight. */
int ld_t5_rmsnorm(const void* x, const void* w, void* out, int64_t rows, int64_t D, float eps, void* stream);

/* T5 self-attention, head_dim 64, N <= 512: q/k/v/out [N][ld] bf16 (head h at columns 64h..64h+63), no score scaling,
 * additive relative-position bias bias_table[bucket[j - i + N - 1]][h] (bf16 [num_buckets][H]), softmax in fp32. */
int ld_t5_attn(const void* q, const void* k, const void* v, void* out, int64_t ld, const void* bias_table,
               const int32_t* bucket, int64_t N, int64_t H, void* stream);

#ifdef __cplusplus
}
#endif
#endif

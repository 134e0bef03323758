"""Tokenize (the encoder half of VideoVQ): Theia feature maps -> TiTok encoder -> nearest VQ code.

Mirrors VideoVQ.encode_to_index(features=...) (landiff/tokenizer/models/video_titok_vq.py:172-202): norm_features (:226-231)
-> TiTokEncoder.forward (landiff/tokenizer/modules/blocks.py:570-656; patch_size 1, 3D RoPE, inside_latent_tokens) with
VideoEncoderMask (landiff/tokenizer/modules/flex_attention_mask.py:36-190) -> VectorQuantize.forward in eval
(vector-quantize-pytorch 1.19.2: project_in, first code at minimum Euclidean distance).  The Theia extractor itself
(theia_extractor.py: a Hugging Face `trust_remote_code` model) is not part of this image and stays outside: the entry point
takes its feature maps, exactly as the reference's `features=` argument does.

HBM layout: one [visual (T*h*w) | latent (I + (T-1)*P)] x width matrix with an fp32 residual stream, as in the decoder.

Attention mask.  The encoder mask is not one threshold order over the keys (a visual query of frame 1 sees V0,V1; the
first I latent sees V0,I0), so it does not fit the kernel's `fid_k[kv] <= fid_q[q]` form directly -- but it does per
query GROUP, and the two groups are contiguous row ranges:
  * visual queries (rows [0, n_visual)): key label = frame for visual keys, +inf for latent keys; query label = frame;
  * latent queries (rows [n_visual, N)): keys in the chain  V0 < I_0 < ... < I_last < V1 < P_1,0 < ... < V2 < P_2,0 ...;
    a latent query's label is its own key label (it sees everything up to itself, and its frame's visual tokens).
So every layer issues two launches of the same masked attention kernel on disjoint query row ranges (pointer offsets
only), each with its own label arrays and per-tile min/max tables for tile skipping.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .config import TokenizerConfig
from .detokenizer import _dev, rope3d_tables

BF = torch.bfloat16
INF = np.iinfo(np.int32).max


def encoder_attention_labels(cfg: TokenizerConfig):
    """(fid_q_visual, fid_k_visual, fid_q_latent, fid_k_latent), int32 [seq_len] each (entries of the other group unused):
    allowed(q, kv) <=> fid_k[kv] <= fid_q[q] within each query group reproduces VideoEncoderMask."""
    T, tpf, nI, nP = cfg.temporal, cfg.tokens_per_frame, cfg.iframe_tokens, cfg.pframe_tokens
    nv, N = T * tpf, cfg.seq_len
    frame = np.repeat(np.arange(T), tpf)
    qv = np.zeros(N, np.int64); qv[:nv] = frame
    kv = np.full(N, INF, np.int64); kv[:nv] = frame
    base = lambda f: 0 if f == 0 else nI + 1 + (f - 1) * (nP + 1)          # label of the visual tokens of frame f
    kl = np.zeros(N, np.int64)
    kl[:nv] = np.repeat(np.array([base(f) for f in range(T)]), tpf)
    kl[nv:nv + nI] = 1 + np.arange(nI)
    for f in range(1, T):
        s = nv + nI + (f - 1) * nP
        kl[s:s + nP] = base(f) + 1 + np.arange(nP)
    ql = np.zeros(N, np.int64); ql[nv:] = kl[nv:]
    return qv.astype(np.int32), kv.astype(np.int32), ql.astype(np.int32), kl.astype(np.int32)


class TokenizerEncoder:
    def __init__(self, enc_sd: dict, tc: TokenizerConfig, device):
        self.tc, self.dev = tc, device
        g = lambda k: _dev(enc_sd[k], device)
        f32 = lambda k: _dev(enc_sd[k], device, torch.float32)
        w = tc.width
        # norm_features is the identity unless the config names a mean_std_path (video_titok_vq.py:221-226; the shipped
        # tokenizer_cfg.py only sets mean_std_dim, so the checkpoint's buffers are NOT applied): zeros / ones make the
        # same kernel an exact pass-through ((x - 0) / (1 + 1e-8f) == x in fp32) that still does the channels-last transpose
        if tc.norm_features:
            self.mean, self.std = f32("mean"), f32("std")
        else:
            self.mean = torch.zeros(tc.out_channels, device=device, dtype=torch.float32)
            self.std = torch.ones(tc.out_channels, device=device, dtype=torch.float32)
        self.patch_w, self.patch_b = g("encoder.patch_embed.weight").reshape(w, -1).contiguous(), g("encoder.patch_embed.bias")
        self.latent = torch.cat([g("encoder.IFrame_latent_tokens"),
                                 g("encoder.PFrame_latent_tokens").repeat(tc.temporal - 1, 1)], 0).contiguous()
        self.ln_pre = (g("encoder.ln_pre.weight"), g("encoder.ln_pre.bias"))
        self.ln_post = (g("encoder.ln_post.weight"), g("encoder.ln_post.bias"))
        self.proj_out = (g("encoder.proj_out.weight"), g("encoder.proj_out.bias"))
        self.blocks = []
        for i in range(tc.layers):
            p = f"encoder.transformer.{i}."
            self.blocks.append(dict(
                ln1=(g(p + "ln_1.weight"), g(p + "ln_1.bias")), ln2=(g(p + "ln_2.weight"), g(p + "ln_2.bias")),
                wqkv=torch.cat([g(p + "attn.wq.weight"), g(p + "attn.wk.weight"), g(p + "attn.wv.weight")], 0).contiguous(),
                wo=g(p + "attn.wo.weight"),
                fc=(g(p + "mlp.c_fc.weight"), g(p + "mlp.c_fc.bias")), proj=(g(p + "mlp.c_proj.weight"), g(p + "mlp.c_proj.bias"))))
        # VQ: project_in (token_size -> 16, output padded to 64 columns for the MFMA GEMM) and the fp32 codebook
        self.pin_w = torch.zeros(64, tc.token_size, device=device, dtype=BF)
        self.pin_w[: tc.codebook_dim] = g("quantizer.project_in.weight")
        self.pin_b = torch.zeros(64, device=device, dtype=BF)
        self.pin_b[: tc.codebook_dim] = g("quantizer.project_in.bias")
        self.codebook = f32("quantizer._codebook.embed")[0].contiguous()             # [V, 16]
        # static tables: RoPE and the two label sets (+ per-64-key-tile min/max) of the mask
        N, nv = tc.seq_len, tc.n_visual
        nlat = N - nv
        self.N = N
        self.Npad = (nv + (nlat + 127) // 128 * 128 + 127) // 128 * 128               # latent q-blocks start at row n_visual
        qv, kv, ql, kl = encoder_attention_labels(tc)
        def padded(a, fill):
            out = np.full(self.Npad, fill, np.int32); out[:N] = a
            return out
        self.labels = []
        for fq, fk in ((qv, kv), (ql, kl)):
            fkp = padded(fk, INF)
            kt = fkp.reshape(-1, 64)
            self.labels.append(tuple(torch.from_numpy(x).to(device) for x in
                                     (padded(fq, 0), fkp, kt.min(1).copy(), kt.max(1).copy())))
        cos, sin = rope3d_tables(tc)
        self.cos, self.sin = cos.to(device), sin.to(device)

    @torch.no_grad()
    def encode(self, features: torch.Tensor) -> torch.Tensor:
        """features [T, C, h, w] (fp32 or bf16, on the device) -> latent tokens [L, token_size] bf16."""
        tc, dev = self.tc, self.dev
        w, N, H, nv = tc.width, self.N, tc.heads, tc.n_visual
        T, C, gh, gw = features.shape
        assert (T, C, gh, gw) == (tc.temporal, tc.out_channels, tc.grid_h, tc.grid_w), features.shape
        xin = torch.empty(nv, C, device=dev, dtype=BF)
        ops.feature_norm_cl(features.contiguous(), self.mean, self.std, xin, T, C, gh * gw)       # norm_features (identity by default) -> bf16, channels-last
        x0 = torch.empty(N, w, device=dev, dtype=BF)
        ops.gemm(xin, self.patch_w, out=x0[:nv], bias=self.patch_b)                               # 1x1 patch embedding
        x0[nv:] = self.latent
        x = torch.empty(N, w, device=dev, dtype=torch.float32)                                    # fp32 residual stream
        ops.layernorm(x0, *self.ln_pre, x, tc.ln_eps)
        ln = torch.empty(N, w, device=dev, dtype=BF)
        qkv = torch.empty(N, 3 * w, device=dev, dtype=BF)
        q = torch.zeros(1, H, self.Npad, 64, device=dev, dtype=BF)
        k = torch.zeros_like(q)
        vt = torch.zeros(1, H, 64, self.Npad, device=dev, dtype=BF)
        att = torch.empty(1, N, w, device=dev, dtype=BF)
        hid = torch.empty(N, 4 * w, device=dev, dtype=BF)
        scale = tc.head_dim ** -0.5
        for blk in self.blocks:
            ops.layernorm(x, *blk["ln1"], ln, tc.ln_eps)
            ops.gemm(ln, blk["wqkv"], out=qkv)
            ops.qkv_split(qkv, q, k, vt, 1, N, H, self.Npad, rope=(self.cos, self.sin))
            ops.attn_fwd(q, k, vt, att, nv, N, scale, *self.labels[0])                            # visual query rows
            ops.attn_fwd(q, k, vt, att, N - nv, N, scale, *self.labels[1], q_row0=nv)             # latent query rows
            ops.gemm(att.view(N, w), blk["wo"], out=x, resid=x, out_f32=True)
            ops.layernorm(x, *blk["ln2"], ln, tc.ln_eps)
            ops.gemm(ln, blk["fc"][0], out=hid, bias=blk["fc"][1], act="gelu_erf")
            ops.gemm(hid, blk["proj"][0], out=x, bias=blk["proj"][1], resid=x, out_f32=True)
        ops.layernorm(x[nv:], *self.ln_post, ln[nv:], tc.ln_eps)
        return ops.gemm(ln[nv:], self.proj_out[0], bias=self.proj_out[1])

    @torch.no_grad()
    def nearest_code(self, z: torch.Tensor) -> torch.Tensor:
        """z [L, token_size] bf16 -> indices int64 [L]: project_in (bf16 Linear) then the first code at minimum squared
        Euclidean distance, fp32."""
        tc = self.tc
        zc = ops.gemm(z, self.pin_w, bias=self.pin_b)                                             # [L, 64], cols >= 16 are 0
        idx = torch.empty(z.shape[0], device=self.dev, dtype=torch.int64)
        ops.vq_nearest(zc, self.codebook, idx, tc.codebook_dim)
        return idx

    @torch.no_grad()
    def encode_to_index(self, features: torch.Tensor) -> torch.Tensor:
        """Theia feature maps [T, C, h, w] -> semantic token ids int64 [L] (VideoVQ.encode_to_index, batch 1)."""
        return self.nearest_code(self.encode(features))

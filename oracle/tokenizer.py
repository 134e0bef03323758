"""Oracle: detokenize -- VQ lookup, TiTok decoder, conv upsampler, SemanticCond
(TEST INFRASTRUCTURE -- see oracle/__init__.py).

Follows landiff/tokenizer/models/video_titok_vq.py:82-106,250-277 (index_to_latent/index_to_feature),
landiff/tokenizer/modules/blocks.py:102-304 (MultiheadAttention, ResidualAttentionBlock), :659-976 (TiTokDecoder),
landiff/tokenizer/modules/flex_attention_mask.py:193-335 (VideoDecoderMask), landiff/modules/pos_emb.py:126-311
(Rope3DPosEmb, multiple=16), landiff/diffusion/semantic_models/condition.py:86-137 (SemanticCond),
landiff/diffusion/semantic_models/modules/vq_gan_blocks.py:35-66,90-148,480-604 (Decoder).
vector-quantize-pytorch's get_output_from_indices is restated (codebook gather + project_out) -- PARITY UNPINNED.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from .common import group_norm, layer_norm, linear, swish
from .llm import apply_rope


def frame_ids(cfg) -> np.ndarray:
    """Frame id of every position of [T x tokens_per_frame visual | I tokens | (T-1) x P tokens]
    (SURVEY.md Appendix B): allowed(q, kv) <=> fid[kv] <= fid[q]."""
    vis = np.repeat(np.arange(cfg.temporal), cfg.tokens_per_frame)
    lat = np.concatenate([np.zeros(cfg.iframe_tokens, np.int64), np.repeat(np.arange(1, cfg.temporal), cfg.pframe_tokens)])
    return np.concatenate([vis, lat]).astype(np.int32)


def decoder_mask_scalar(cfg, q_idx: int, kv_idx: int) -> bool:
    """Scalar restatement of VideoDecoderMask._mask_fn (flex_attention_mask.py:283-335)."""
    T, tpf, nI, nP = cfg.temporal, cfg.tokens_per_frame, cfg.iframe_tokens, cfg.pframe_tokens
    nv = T * tpf
    if q_idx < tpf or nv <= q_idx < nv + nI:
        return kv_idx < tpf or nv <= kv_idx < nv + nI
    if q_idx < nv:
        f = q_idx // tpf
        if kv_idx // tpf < T:
            return kv_idx // tpf <= f
        return nv <= kv_idx < nv + nI + f * nP
    if q_idx < cfg.seq_len:
        f = (q_idx - nI - nv) // nP + 1
        return kv_idx < (f + 1) * tpf or nv <= kv_idx < nv + nI + f * nP
    return False


def rope3d_table(cfg):
    """Rope3DPosEmb(multiple=16).get_freqs_cis_by_idx for the decoder sequence (blocks.py:862-904):
    visual token (t,h,w) -> [t x 8 freqs | h x 12 | w x 12]; latent token i -> position (i,i,i).
    Returns (cos, sin) fp32 [seq_len, head_dim/2]."""
    D = cfg.head_dim
    t_dim, hw_dim = D // 4, D // 8 * 3
    t_f = 1.0 / (cfg.rope_theta ** (torch.arange(0, t_dim, 2)[: t_dim // 2].float() / t_dim))
    hw_f = 1.0 / (cfg.rope_theta ** (torch.arange(0, hw_dim, 2)[: hw_dim // 2].float() / hw_dim))
    tt, hh, ww = torch.meshgrid(torch.arange(cfg.temporal), torch.arange(cfg.grid_h), torch.arange(cfg.grid_w), indexing="ij")
    vis = torch.stack([tt, hh, ww], -1).reshape(-1, 3)
    lat = torch.arange(cfg.num_latent_tokens)[:, None].expand(-1, 3)
    pos = torch.cat([vis, lat], 0).float()
    ang = torch.cat([torch.outer(pos[:, 0], t_f), torch.outer(pos[:, 1], hw_f), torch.outer(pos[:, 2], hw_f)], -1).float()
    cis = torch.polar(torch.ones_like(ang), ang)
    return cis.real.contiguous(), cis.imag.contiguous()


class DetokenizerOracle:
    def __init__(self, tok_state, ups_state, tok_cfg, ups_cfg, dtype=torch.bfloat16):
        self.t, self.u, self.tc, self.uc, self.dtype = tok_state, ups_state, tok_cfg, ups_cfg, dtype

    # ---- VQ + TiTok decoder ----
    def index_to_latent(self, tokens):
        """codes = codebook[idx]; project_out Linear(16 -> token_size) (bf16 under cuda autocast)."""
        t, dt = self.t, self.dtype
        codes = t["quantizer._codebook.embed"][0].to(dt)[tokens.reshape(-1)]
        return linear(codes, t["quantizer.project_out.weight"], t["quantizer.project_out.bias"], dt)  # [L, token_size]

    def titok_block(self, i, x, cos, sin, mask):
        t, c, dt = self.t, self.tc, self.dtype
        p = f"decoder.transformer.{i}."
        ln_out = torch.float32 if dt == torch.bfloat16 else dt   # autocast runs layer_norm in fp32
        h = layer_norm(x, t[p + "ln_1.weight"], t[p + "ln_1.bias"], c.ln_eps, ln_out)
        B, N, _ = h.shape
        q = linear(h, t[p + "attn.wq.weight"], None, dt).view(B, N, c.heads, c.head_dim)
        k = linear(h, t[p + "attn.wk.weight"], None, dt).view(B, N, c.heads, c.head_dim)
        v = linear(h, t[p + "attn.wv.weight"], None, dt).view(B, N, c.heads, c.head_dim)
        q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
        o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), attn_mask=mask)
        o = o.transpose(1, 2).reshape(B, N, c.width)
        x = x + linear(o, t[p + "attn.wo.weight"], None, dt)
        h = layer_norm(x, t[p + "ln_2.weight"], t[p + "ln_2.bias"], c.ln_eps, ln_out)
        h = F.gelu(linear(h, t[p + "mlp.c_fc.weight"], t[p + "mlp.c_fc.bias"], dt))
        return x + linear(h, t[p + "mlp.c_proj.weight"], t[p + "mlp.c_proj.bias"], dt)

    def index_to_feature(self, tokens):
        """VideoVQ.index_to_feature -> TiTokDecoder.forward (blocks.py:906-976): [1, T, C, H, W]."""
        t, c, dt = self.t, self.tc, self.dtype
        lat = self.index_to_latent(tokens)[None]                          # [1, L, token_size]
        x = linear(lat, t["decoder.decoder_embed.weight"], t["decoder.decoder_embed.bias"], dt)
        mask_tokens = t["decoder.mask_token"].to(x.dtype).expand(1, c.n_visual, -1)
        x = torch.cat([mask_tokens, x], 1)
        ln_out = torch.float32 if dt == torch.bfloat16 else dt
        x = layer_norm(x, t["decoder.ln_pre.weight"], t["decoder.ln_pre.bias"], c.ln_eps, ln_out)
        cos, sin = rope3d_table(c)
        fid = torch.from_numpy(frame_ids(c))
        mask = (fid[None, :] <= fid[:, None])[None, None]
        for i in range(c.layers):
            x = self.titok_block(i, x, cos[None], sin[None], mask)
        x = x[:, : c.n_visual]
        x = layer_norm(x, t["decoder.ln_post.weight"], t["decoder.ln_post.bias"], c.ln_eps, ln_out)
        x = linear(x.contiguous(), t["decoder.ffn.0.weight"], t["decoder.ffn.0.bias"], dt)
        x = linear(torch.tanh(x), t["decoder.ffn.2.weight"], t["decoder.ffn.2.bias"], dt)
        x = x.view(1, c.temporal, c.grid_h, c.grid_w, c.out_channels).permute(0, 1, 4, 2, 3)
        return x                                                          # denorm_features is the identity

    # ---- conv upsampler (vq_gan_blocks.Decoder) ----
    def _conv(self, x, name, pad=1):
        u, dt = self.u, self.dtype
        return F.conv2d(x.to(dt), u[name + ".weight"].to(dt), u[name + ".bias"].to(dt), padding=pad)

    def _res(self, x, p, cin, cout):
        u, c = self.u, self.uc
        h = swish(group_norm(x, c.gn_groups, u[p + "norm1.weight"], u[p + "norm1.bias"], c.gn_eps))
        h = self._conv(h, p + "conv1")
        h = swish(group_norm(h, c.gn_groups, u[p + "norm2.weight"], u[p + "norm2.bias"], c.gn_eps))
        h = self._conv(h, p + "conv2")
        if cin != cout:
            x = self._conv(x, p + "nin_shortcut", pad=0)
        return x + h

    def upsample(self, feats):
        """feats [N, z_channels, h, w] -> [N, out_ch, 2h, 2w] (vq_gan_blocks.py:573-604)."""
        from landiff_amd.weights import upsampler_levels
        u, c = self.u, self.uc
        p = "upsample_model."
        h = self._conv(feats, p + "conv_in")
        top = h.shape[1]
        h = self._res(h, p + "mid.block_1.", top, top)
        h = self._res(h, p + "mid.block_2.", top, top)
        for lvl, blocks, up in upsampler_levels(c):
            for j, (cin, cout) in enumerate(blocks):
                h = self._res(h, p + f"up.{lvl}.block.{j}.", cin, cout)
            if up:
                h = self._conv(F.pixel_shuffle(h, 2), p + f"up.{lvl}.upsample.conv")
        h = swish(group_norm(h, c.gn_groups, u[p + "norm_out.weight"], u[p + "norm_out.bias"], c.gn_eps))
        return self._conv(h, p + "conv_out")

    def semantic_cond(self, tokens):
        """SemanticCond.forward(indexs=tokens) (condition.py:112-137): -> [1, T, target_dim, 2h, 2w]."""
        f = self.index_to_feature(tokens).to(self.dtype)
        B, T = f.shape[:2]
        f = self.upsample(f.reshape(B * T, *f.shape[2:]))
        f = self._conv(f, "conv_out")
        return f.view(B, T, *f.shape[1:])

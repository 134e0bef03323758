"""Detokenize: semantic tokens -> TiTok decoder features -> conv upsampler -> latent-shaped control signal.

Mirrors SemanticCond.forward(indexs=...) (landiff/diffusion/semantic_models/condition.py:112-137):
VideoVQ.index_to_feature (landiff/tokenizer/models/video_titok_vq.py:82-106,250-277; VQ lookup restated from
vector-quantize-pytorch) -> TiTokDecoder (landiff/tokenizer/modules/blocks.py:659-976) with the
frame-block mask `fid[kv] <= fid[q]` (flex_attention_mask.py:193-335) and 3D RoPE
(landiff/modules/pos_emb.py:126-311) -> vq_gan_blocks.Decoder (vq_gan_blocks.py:480-604) -> zero-init Conv2d.

HBM layout: the decoder sequence is a [visual(T*h*w) | latent(1218)] x width matrix with an fp32 residual stream;
its first T*h*w rows ARE the channels-last [T][h][w][C] feature map the conv upsampler consumes, so no transpose
ever happens between the transformer and the convolutions.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .config import TokenizerConfig, UpsamplerConfig
from .weights import upsampler_levels

BF = torch.bfloat16


def decoder_frame_ids(cfg: TokenizerConfig) -> np.ndarray:
    vis = np.repeat(np.arange(cfg.temporal), cfg.tokens_per_frame)
    lat = np.concatenate([np.zeros(cfg.iframe_tokens, np.int64), np.repeat(np.arange(1, cfg.temporal), cfg.pframe_tokens)])
    return np.concatenate([vis, lat]).astype(np.int32)


def rope3d_tables(cfg: TokenizerConfig):
    """(cos, sin) fp32 [seq_len, head_dim/2]: visual (t,h,w) -> [t x D/8 | h x 3D/16 | w x 3D/16] frequencies,
    latent token i -> (i,i,i)  (Rope3DPosEmb multiple=16; blocks.py:862-904)."""
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


def _dev(t, device, dtype=BF):
    return t.detach().to(device=device, dtype=dtype).contiguous()


def _conv_w(w, device, cin_pad=None):
    """torch conv weight [Cout, Cin, (kT,) kH, kW] -> channels-last [Cout, kT, kH, kW, Cin(_pad)] bf16."""
    if w.dim() == 4:
        w = w.unsqueeze(2)
    w = w.permute(0, 2, 3, 4, 1)
    if cin_pad is not None and cin_pad > w.shape[-1]:
        w = torch.nn.functional.pad(w, (0, cin_pad - w.shape[-1]))
    return _dev(w, device)


class Detokenizer:
    def __init__(self, tok_sd: dict, ups_sd: dict, tc: TokenizerConfig, uc: UpsamplerConfig, device):
        self.tc, self.uc, self.dev = tc, uc, device
        g = lambda k: _dev(tok_sd[k], device)
        w = tc.width
        # VQ: codebook [V,16] and project_out (K padded 16 -> 64 for the MFMA GEMM)
        self.codebook = torch.zeros(tc.codebook_size, 64, device=device, dtype=BF)
        self.codebook[:, : tc.codebook_dim] = g("quantizer._codebook.embed")[0]
        self.proj_w = torch.zeros(tc.token_size, 64, device=device, dtype=BF)
        self.proj_w[:, : tc.codebook_dim] = g("quantizer.project_out.weight")
        self.proj_b = g("quantizer.project_out.bias")
        self.embed_w, self.embed_b = g("decoder.decoder_embed.weight"), g("decoder.decoder_embed.bias")
        self.mask_token = g("decoder.mask_token").view(1, w)
        self.ln_pre = (g("decoder.ln_pre.weight"), g("decoder.ln_pre.bias"))
        self.ln_post = (g("decoder.ln_post.weight"), g("decoder.ln_post.bias"))
        self.ffn0 = (g("decoder.ffn.0.weight"), g("decoder.ffn.0.bias"))
        self.ffn2 = (g("decoder.ffn.2.weight"), g("decoder.ffn.2.bias"))
        # denorm_features: identity unless the config names a mean_std_path (video_titok_vq.py:228-233)
        self.denorm = None
        if tc.norm_features:
            self.denorm = (_dev(tok_sd["mean"].reshape(-1), device, torch.float32), _dev(tok_sd["std"].reshape(-1), device, torch.float32))
        self.blocks = []
        for i in range(tc.layers):
            p = f"decoder.transformer.{i}."
            self.blocks.append(dict(
                ln1=(g(p + "ln_1.weight"), g(p + "ln_1.bias")), ln2=(g(p + "ln_2.weight"), g(p + "ln_2.bias")),
                wqkv=torch.cat([g(p + "attn.wq.weight"), g(p + "attn.wk.weight"), g(p + "attn.wv.weight")], 0).contiguous(),
                wo=g(p + "attn.wo.weight"),
                fc=(g(p + "mlp.c_fc.weight"), g(p + "mlp.c_fc.bias")), proj=(g(p + "mlp.c_proj.weight"), g(p + "mlp.c_proj.bias"))))
        # mask + RoPE tables (static)
        N = tc.seq_len
        self.N, self.Npad = N, (N + 127) // 128 * 128
        fid = decoder_frame_ids(tc)
        fq = np.zeros(self.Npad, np.int32); fq[:N] = fid
        fk = np.full(self.Npad, np.iinfo(np.int32).max, np.int32); fk[:N] = fid
        kt = fk.reshape(-1, 64)
        self.fid_q = torch.from_numpy(fq).to(device)
        self.fid_k = torch.from_numpy(fk).to(device)
        self.kt_min = torch.from_numpy(kt.min(1).copy()).to(device)
        self.kt_max = torch.from_numpy(kt.max(1).copy()).to(device)
        cos, sin = rope3d_tables(tc)
        self.cos, self.sin = cos.to(device), sin.to(device)
        # upsampler weights
        u = lambda k: _dev(ups_sd[k], device)
        self.u = {}
        for k, v in ups_sd.items():
            if k.endswith(".weight") and v.dim() == 4:
                if v.shape[-1] == 1:
                    self.u[k] = _dev(v.reshape(v.shape[0], v.shape[1]), device)        # 1x1 conv -> GEMM
                else:
                    self.u[k] = _conv_w(v, device)
            else:
                self.u[k] = u(k)

    # ---- TiTok decoder --------------------------------------------------------------------
    @torch.no_grad()
    def index_to_latent(self, tokens: torch.Tensor) -> torch.Tensor:
        """VideoVQ.index_to_latent (video_titok_vq.py:92-94): tokens int64 [L] on device -> project_out(codebook[tokens]),
        bf16 [L, token_size]."""
        codes = self.codebook[tokens.reshape(-1)]                                    # gather (index plumbing)
        return ops.gemm(codes, self.proj_w, bias=self.proj_b)

    @torch.no_grad()
    def index_to_feature(self, tokens: torch.Tensor) -> torch.Tensor:
        """tokens int64 [L] on device -> features [T, h, w, C] bf16 (channels-last view of [1,T,C,h,w])."""
        return self.latent_to_feature(self.index_to_latent(tokens))

    @torch.no_grad()
    def latent_to_feature(self, lat: torch.Tensor) -> torch.Tensor:
        """TiTokDecoder.forward (tokenizer/modules/blocks.py): latent tokens bf16 [L, token_size] on device -> features
        [T, h, w, C] bf16 (channels-last view of [1,T,C,h,w])."""
        tc, dev = self.tc, self.dev
        w, N, H = tc.width, self.N, tc.heads
        nv = tc.n_visual
        lat = lat.to(dev, BF).contiguous()
        x0 = torch.empty(N, w, device=dev, dtype=BF)
        x0[:nv] = self.mask_token
        ops.gemm(lat, self.embed_w, out=x0[nv:], bias=self.embed_b)
        x = torch.empty(N, w, device=dev, dtype=torch.float32)                        # fp32 residual stream
        ops.layernorm(x0, *self.ln_pre, x, tc.ln_eps)
        ln = torch.empty(N, w, device=dev, dtype=BF)
        qkv = torch.empty(N, 3 * w, device=dev, dtype=BF)
        q = torch.zeros(1, H, self.Npad, 64, device=dev, dtype=BF)
        k = torch.zeros_like(q)
        vt = torch.zeros(1, H, 64, self.Npad, device=dev, dtype=BF)
        att = torch.empty(1, N, w, device=dev, dtype=BF)
        hid = torch.empty(N, 4 * w, device=dev, dtype=BF)
        for blk in self.blocks:
            ops.layernorm(x, *blk["ln1"], ln, tc.ln_eps)
            ops.gemm(ln, blk["wqkv"], out=qkv)
            ops.qkv_split(qkv, q, k, vt, 1, N, H, self.Npad, rope=(self.cos, self.sin))
            ops.attn_fwd(q, k, vt, att, N, N, tc.head_dim ** -0.5, fid_q=self.fid_q, fid_k=self.fid_k,
                         kt_min=self.kt_min, kt_max=self.kt_max)
            ops.gemm(att.view(N, w), blk["wo"], out=x, resid=x, out_f32=True)
            ops.layernorm(x, *blk["ln2"], ln, tc.ln_eps)
            ops.gemm(ln, blk["fc"][0], out=hid, bias=blk["fc"][1], act="gelu_erf")
            ops.gemm(hid, blk["proj"][0], out=x, bias=blk["proj"][1], resid=x, out_f32=True)
        ops.layernorm(x[:nv], *self.ln_post, ln[:nv], tc.ln_eps)
        f1 = ops.gemm(ln[:nv], self.ffn0[0], bias=self.ffn0[1], act="tanh")
        feats = ops.gemm(f1, self.ffn2[0], bias=self.ffn2[1])
        if self.denorm is not None:
            ops.feature_denorm(feats, *self.denorm)
        return feats.view(tc.temporal, tc.grid_h, tc.grid_w, tc.out_channels)

    # ---- conv upsampler -------------------------------------------------------------------
    def _conv3x3(self, x_padded, name, F, H, W, **epi):
        return ops.conv_cl(x_padded, self.u[name + ".weight"], F, H, W, bias=self.u[name + ".bias"], **epi)

    def _norm_swish_pad(self, x, name, F, H, W, C):
        uc, dev = self.uc, self.dev
        stats = torch.empty(F, uc.gn_groups, 2, device=dev, dtype=torch.float64)
        ops.groupnorm_stats(x, stats, F, H * W, C, uc.gn_groups)
        out = torch.zeros(F, H + 2, W + 2, C, device=dev, dtype=BF)
        ops.groupnorm_apply(x, out, stats, self.u[name + ".weight"], self.u[name + ".bias"], F, 1, H, W, C, uc.gn_groups,
                            hpad=1, wpad=1, swish=True, eps=uc.gn_eps)
        return out

    def _res(self, x, p, cin, cout, F, H, W):
        h = self._conv3x3(self._norm_swish_pad(x, p + "norm1", F, H, W, cin), p + "conv1", F, H, W)
        hp = self._norm_swish_pad(h, p + "norm2", F, H, W, cout)
        if cin != cout:
            x = ops.gemm(x, self.u[p + "nin_shortcut.weight"], bias=self.u[p + "nin_shortcut.bias"])
        return self._conv3x3(hp, p + "conv2", F, H, W, resid=x)

    def _pad(self, x, F, H, W, C):
        out = torch.zeros(F, H + 2, W + 2, C, device=self.dev, dtype=BF)
        ops.place_cl(x, out, F, 1, H, W, C, C, mode=0)
        return out

    @torch.no_grad()
    def upsample(self, feats: torch.Tensor) -> torch.Tensor:
        """feats [F, h, w, z_channels] bf16 -> [F, 2h, 2w, out_ch] (channels-last)."""
        uc = self.uc
        F, H, W, C = feats.shape
        p = "upsample_model."
        h = self._conv3x3(self._pad(feats.reshape(-1, C), F, H, W, C), p + "conv_in", F, H, W)
        top = h.shape[1]
        h = self._res(h, p + "mid.block_1.", top, top, F, H, W)
        h = self._res(h, p + "mid.block_2.", top, top, F, H, W)
        ch = top
        for lvl, blocks, up in upsampler_levels(uc):
            for j, (cin, cout) in enumerate(blocks):
                h = self._res(h, p + f"up.{lvl}.block.{j}.", cin, cout, F, H, W)
                ch = cout
            if up:   # PixelShuffle(2) straight into the zero-bordered conv input
                ps = torch.zeros(F, 2 * H + 2, 2 * W + 2, ch // 4, device=self.dev, dtype=BF)
                ops.place_cl(h, ps, F, 1, H, W, ch, ch // 4, mode=2)
                H, W = 2 * H, 2 * W
                h = self._conv3x3(ps, p + f"up.{lvl}.upsample.conv", F, H, W)
        h = self._conv3x3(self._norm_swish_pad(h, p + "norm_out", F, H, W, ch), p + "conv_out", F, H, W)
        return h.view(F, H, W, uc.out_ch)

    @torch.no_grad()
    def semantic_condition_from_features(self, features: torch.Tensor) -> torch.Tensor:
        """SemanticCond.forward(semantic_feature_before_upsample=...) (condition.py:85-110,112-137): detokenizer features
        [1, T, C, h, w] (or [T, C, h, w]) given by the caller -> upsampler -> conv_out -> [T, target_dim, 2h, 2w] bf16."""
        f = features.reshape(-1, *features.shape[-3:]).to(self.dev, BF).permute(0, 2, 3, 1).contiguous()     # channels-last
        return self._condition_from_cl(f)

    @torch.no_grad()
    def semantic_condition(self, tokens: torch.Tensor) -> torch.Tensor:
        """-> [T, target_dim, 2h, 2w] bf16 (the tensor the control DiT adds to its input latent)."""
        return self._condition_from_cl(self.index_to_feature(tokens))

    def _condition_from_cl(self, feats_cl: torch.Tensor) -> torch.Tensor:
        uc = self.uc
        f = self.upsample(feats_cl)
        F, H, W, C = f.shape
        out = self._conv3x3(self._pad(f.reshape(-1, C), F, H, W, C), "conv_out", F, H, W)     # [F*H*W, target_dim]
        return out.view(F, H, W, uc.target_dim).permute(0, 3, 1, 2).contiguous()

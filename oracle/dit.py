"""Oracle: the control + main diffusion transformers (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Follows landiff/diffusion/dit_video_concat.py: ImagePatchEmbeddingMixin :25-68, get_3d_sincos_pos_embed :71-196,
Basic3DPositionEmbeddingMixin :200-246, modulate :388, unpatchify :392-410, FinalLayerMixin :413-460,
AdaLNMixin :490-664, DiffusionTransformer.forward :872-909, ControlDiffusionTransformer.forward :935-1027,
ControlDiffWarp :1164-1200, ControlOutAdaLNMixin :1203-1238, ControlAdaLNMixin :1241-1372.
The sat 0.4.12 internals (BaseTransformer/SelfAttention/MLP) are restated from SURVEY.md 8c -- PARITY UNPINNED.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from .common import layer_norm, linear, timestep_embedding


# ---- position table (numpy, as the reference) ----
def _sincos_1d(embed_dim, pos):
    omega = np.arange(embed_dim // 2, dtype=np.float64)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def _sincos_2d(embed_dim, grid):
    emb_h = _sincos_1d(embed_dim // 2, grid[0])
    emb_w = _sincos_1d(embed_dim // 2, grid[1])
    return np.concatenate([emb_h, emb_w], axis=1)


def sincos_pos_embed_3d(embed_dim, grid_h, grid_w, t_size, h_interp=1.0, w_interp=1.0, t_interp=1.0):
    """dit_video_concat.py:71-117 -> [T, H*W, D] (temporal quarter first, then spatial 3/4)."""
    d_sp, d_t = embed_dim // 4 * 3, embed_dim // 4
    gh = np.arange(grid_h, dtype=np.float32) / h_interp
    gw = np.arange(grid_w, dtype=np.float32) / w_interp
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape([2, 1, grid_h, grid_w])
    sp = _sincos_2d(d_sp, grid)
    gt = np.arange(t_size, dtype=np.float32) / t_interp
    tm = _sincos_1d(d_t, gt)
    tm = np.repeat(tm[:, None, :], grid_h * grid_w, axis=1)
    sp = np.repeat(sp[None], t_size, axis=0)
    return np.concatenate([tm, sp], axis=-1)


def pos_embedding_table(cfg):
    """Basic3DPositionEmbeddingMixin.reinit (:234-246): text rows zero."""
    pe = sincos_pos_embed_3d(cfg.hidden, cfg.grid_h, cfg.grid_w, cfg.latent_frames,
                             cfg.height_interpolation, cfg.width_interpolation, cfg.time_interpolation)
    pe = torch.from_numpy(pe).float().reshape(-1, cfg.hidden)
    out = torch.zeros(1, cfg.seq_len, cfg.hidden)
    out[0, cfg.text_len:] = pe
    return out


def modulate(x, shift, scale):
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


class DiTOracle:
    """One DiffusionTransformer (control=False) or ControlDiffusionTransformer (control=True)."""

    def __init__(self, state: dict, cfg, control: bool, dtype=torch.bfloat16):
        self.s, self.cfg, self.control, self.dtype = state, cfg, control, dtype
        self.L = cfg.layers_control if control else cfg.layers_main

    def _ln(self, x, name, eps):
        return layer_norm(x, self.s[name + ".weight"], self.s[name + ".bias"], eps)

    def embed(self, x, context):
        """word_embedding_forward (:47-62) + position_embedding_forward (:227-231)."""
        s, c, dt = self.s, self.cfg, self.dtype
        B, T = x.shape[:2]
        emb = F.conv2d(x.reshape(-1, *x.shape[2:]).to(dt), s["mixins.patch_embed.proj.weight"].to(dt),
                       s["mixins.patch_embed.proj.bias"].to(dt), stride=c.patch)
        emb = emb.view(B, T, *emb.shape[1:]).flatten(3).transpose(2, 3).reshape(B, -1, c.hidden)
        text = linear(context, s["mixins.patch_embed.text_proj.weight"], s["mixins.patch_embed.text_proj.bias"], dt)
        h = torch.cat([text, emb], 1)
        n = c.text_len + T * c.grid_h * c.grid_w
        return h + s["mixins.pos_embed.pos_embedding"][:, :n].to(dt)

    def attention(self, i, x):
        """sat SelfAttention (restated): qkv -> [q|k|v] thirds -> heads -> QK-LN (:649-653) -> SDPA -> dense."""
        s, c, dt = self.s, self.cfg, self.dtype
        p = f"transformer.layers.{i}.attention."
        B, N, _ = x.shape
        qkv = linear(x, s[p + "query_key_value.weight"], s[p + "query_key_value.bias"], dt)
        q, k, v = qkv.chunk(3, dim=-1)
        sh = lambda t: t.view(B, N, c.heads, c.head_dim).permute(0, 2, 1, 3)
        q, k, v = sh(q), sh(k), sh(v)
        q = self._ln(q, f"mixins.adaln_layer.query_layernorm_list.{i}", c.qk_ln_eps)
        k = self._ln(k, f"mixins.adaln_layer.key_layernorm_list.{i}", c.qk_ln_eps)
        o = F.scaled_dot_product_attention(q, k, v)
        o = o.permute(0, 2, 1, 3).reshape(B, N, c.hidden)
        return linear(o, s[p + "dense.weight"], s[p + "dense.bias"], dt)

    def mlp(self, i, x):
        s, dt = self.s, self.dtype
        p = f"transformer.layers.{i}.mlp."
        h = F.gelu(linear(x, s[p + "dense_h_to_4h.weight"], s[p + "dense_h_to_4h.bias"], dt), approximate="tanh")
        return linear(h, s[p + "dense_4h_to_h.weight"], s[p + "dense_4h_to_h.bias"], dt)

    def layer(self, i, h, emb, control_out=None):
        """AdaLNMixin.layer_forward (:540-629) (+ control add :1357-1370, + zero-linear :1234-1237)."""
        s, c, dt = self.s, self.cfg, self.dtype
        tl = c.text_len
        ada = linear(F.silu(emb), s[f"mixins.adaln_layer.adaLN_modulations.{i}.1.weight"],
                     s[f"mixins.adaln_layer.adaLN_modulations.{i}.1.bias"], dt)
        (sh_msa, sc_msa, g_msa, sh_mlp, sc_mlp, g_mlp,
         tsh_msa, tsc_msa, tg_msa, tsh_mlp, tsc_mlp, tg_mlp) = ada.chunk(12, dim=1)
        txt, img = h[:, :tl], h[:, tl:]
        ln = f"transformer.layers.{i}.input_layernorm"
        a_in = torch.cat([modulate(self._ln(txt, ln, c.block_ln_eps), tsh_msa, tsc_msa),
                          modulate(self._ln(img, ln, c.block_ln_eps), sh_msa, sc_msa)], 1)
        a = self.attention(i, a_in)
        img = img + g_msa.unsqueeze(1) * a[:, tl:]
        txt = txt + tg_msa.unsqueeze(1) * a[:, :tl]
        ln = f"transformer.layers.{i}.post_attention_layernorm"
        m_in = torch.cat([modulate(self._ln(txt, ln, c.block_ln_eps), tsh_mlp, tsc_mlp),
                          modulate(self._ln(img, ln, c.block_ln_eps), sh_mlp, sc_mlp)], 1)
        m = self.mlp(i, m_in)
        img = img + g_mlp.unsqueeze(1) * m[:, tl:]
        txt = txt + tg_mlp.unsqueeze(1) * m[:, :tl]
        h = torch.cat([txt, img], 1)
        if control_out is not None:
            h = h + control_out
        if self.control:
            h = linear(h, s[f"mixins.adaln_layer.zero_linears.{i}.weight"], None, dt)
        return h

    def time_emb(self, timesteps):
        s, c, dt = self.s, self.cfg, self.dtype
        t_emb = timestep_embedding(timesteps, c.hidden, dtype=dt)
        e = linear(t_emb, s["time_embed.0.weight"], s["time_embed.0.bias"], dt)
        return linear(F.silu(e), s["time_embed.2.weight"], s["time_embed.2.bias"], dt)

    def forward(self, x, timesteps, context, semantic_feature=None, control_outputs=None):
        """x [B,T,C,H,W]; returns per-layer hidden states (control) or the eps prediction [B,T,C,H,W]."""
        s, c, dt = self.s, self.cfg, self.dtype
        x = x.to(dt)
        if self.control:
            x = x + semantic_feature.to(dt)                     # :991
        emb = self.time_emb(timesteps)
        h = self.embed(x, context.to(dt))
        outs = []
        for i in range(self.L):
            co = control_outputs[i] if (control_outputs is not None and i < len(control_outputs)) else None
            h = self.layer(i, h, emb, co)
            outs.append(h)
        if self.control:
            return outs
        h = self._ln(h, "transformer.final_layernorm", c.block_ln_eps)
        xi = h[:, c.text_len:]                                   # FinalLayerMixin.final_forward :442-456
        mod = linear(F.silu(emb), s["mixins.final_layer.adaLN_modulation.1.weight"],
                     s["mixins.final_layer.adaLN_modulation.1.bias"], dt)
        shift, scale = mod.chunk(2, dim=1)
        xi = modulate(self._ln(xi, "mixins.final_layer.norm_final", c.final_ln_eps), shift, scale)
        xi = linear(xi, s["mixins.final_layer.linear.weight"], s["mixins.final_layer.linear.bias"], dt)
        B, T = x.shape[:2]
        p = c.patch
        xi = xi.view(B, T, c.grid_h, c.grid_w, c.out_channels, p, p)   # "b (t h w) (c p q) -> b t c (h p) (w q)"
        return xi.permute(0, 1, 4, 2, 5, 3, 6).reshape(B, T, c.out_channels, c.grid_h * p, c.grid_w * p)


class ControlDiTOracle:
    """ControlDiffWarp.forward (:1196-1200): control net, then main net with the control hidden states."""

    def __init__(self, main_state, control_state, cfg, dtype=torch.bfloat16):
        self.main = DiTOracle(main_state, cfg, False, dtype)
        self.ctrl = DiTOracle(control_state, cfg, True, dtype)

    def __call__(self, x, timesteps, context, semantic_feature):
        outs = self.ctrl.forward(x, timesteps, context, semantic_feature=semantic_feature)
        return self.main.forward(x, timesteps, context, control_outputs=outs)

"""Oracle: CogVideoX 3D-VAE decode with the reference's chunk schedule and conv caches
(TEST INFRASTRUCTURE -- see oracle/__init__.py).

Follows landiff/diffusion/dif_infer.py:245-271 (decode_latent: chunks 0:3, 3:5, ..., cache cleared on the last),
landiff/diffusion/vae_modules/cp_enc_dec.py:249-300,383-473 (causal conv + fake-CP cache),
:502-569 (SpatialNorm3D), :590-633 (Upsample3D), :683-782 (resblock), :912-1069 (decoder),
landiff/diffusion/dif_infer.py:37-49 (_post_process_cog_video), landiff/utils.py:327-331 (uint8 truncation).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .common import group_norm, swish


def nearest_zq(zq, f_shape):
    """SpatialNorm3D zq interpolation (cp_enc_dec.py:546-560): first-frame split when T is odd > 1."""
    T = f_shape[0]
    if T > 1 and T % 2 == 1:
        first = F.interpolate(zq[:, :, :1], size=(1, f_shape[1], f_shape[2]), mode="nearest")
        rest = F.interpolate(zq[:, :, 1:], size=(T - 1, f_shape[1], f_shape[2]), mode="nearest")
        return torch.cat([first, rest], dim=2)
    return F.interpolate(zq, size=tuple(f_shape), mode="nearest")


class VAEDecoderOracle:
    def __init__(self, state, cfg, dtype=torch.bfloat16):
        self.s, self.cfg, self.dtype = state, cfg, dtype
        self.cache = {}

    def causal_conv(self, x, name, clear_cache):
        """ContextParallelCausalConv3d.forward (:416-473), cp world size 1."""
        s, dt = self.s, self.dtype
        w = s[name + ".conv.weight"].to(dt)
        b = s[name + ".conv.bias"].to(dt)
        kt = w.shape[2]
        if kt > 1:
            pad = self.cache.get(name)
            if pad is None:
                x = torch.cat([x[:, :, :1]] * (kt - 1) + [x], dim=2)
            else:
                x = torch.cat([pad, x], dim=2)
            self.cache.pop(name, None)
            if not clear_cache:
                self.cache[name] = x[:, :, -kt + 1:].contiguous().clone()
        hp, wp = w.shape[3] // 2, w.shape[4] // 2
        x = F.pad(x, (wp, wp, hp, hp))
        return F.conv3d(x.to(dt), w, b)

    def spatial_norm(self, f, zq, name):
        s, c = self.s, self.cfg
        zq = nearest_zq(zq, f.shape[-3:])
        nf = group_norm(f, c.gn_groups, s[name + ".norm_layer.weight"], s[name + ".norm_layer.bias"], c.gn_eps)
        y = self.causal_conv(zq, name + ".conv_y", True)
        b = self.causal_conv(zq, name + ".conv_b", True)
        return nf * y + b

    def resblock(self, x, zq, p, cin, cout, clear):
        s, dt = self.s, self.dtype
        h = swish(self.spatial_norm(x, zq, p + "norm1"))
        h = self.causal_conv(h, p + "conv1", clear)
        h = swish(self.spatial_norm(h, zq, p + "norm2"))
        h = self.causal_conv(h, p + "conv2", clear)
        if cin != cout:
            x = F.conv3d(x.to(dt), s[p + "nin_shortcut.weight"].to(dt), s[p + "nin_shortcut.bias"].to(dt))
        return x + h

    def upsample(self, x, name, compress_time):
        """Upsample3D.forward (:605-633)."""
        s, dt = self.s, self.dtype
        if compress_time and x.shape[2] > 1:
            if x.shape[2] % 2 == 1:
                first = F.interpolate(x[:, :, 0], scale_factor=2.0, mode="nearest")
                rest = F.interpolate(x[:, :, 1:], scale_factor=2.0, mode="nearest")
                x = torch.cat([first[:, :, None], rest], dim=2)
            else:
                x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        else:
            t = x.shape[2]
            x2 = x.permute(0, 2, 1, 3, 4).reshape(-1, x.shape[1], *x.shape[3:])
            x2 = F.interpolate(x2, scale_factor=2.0, mode="nearest")
            x = x2.view(x.shape[0], t, *x2.shape[1:]).permute(0, 2, 1, 3, 4)
        t = x.shape[2]
        x2 = x.permute(0, 2, 1, 3, 4).reshape(-1, x.shape[1], *x.shape[3:])
        x2 = F.conv2d(x2.to(dt), s[name + ".conv.weight"].to(dt), s[name + ".conv.bias"].to(dt), padding=1)
        return x2.view(x.shape[0], t, *x2.shape[1:]).permute(0, 2, 1, 3, 4)

    def decode_chunk(self, z, clear_cache):
        """ContextParallelDecoder3D.forward (:1034-1069).  z [1, 16, t, h, w]."""
        from landiff_amd.weights import vae_levels
        c = self.cfg
        z = z.to(self.dtype)
        zq = z
        p = "decoder."
        h = self.causal_conv(z, p + "conv_in", clear_cache)
        top = h.shape[1]
        h = self.resblock(h, zq, p + "mid.block_1.", top, top, clear_cache)
        h = self.resblock(h, zq, p + "mid.block_2.", top, top, clear_cache)
        for lvl, blocks, up in vae_levels(c):
            for j, (cin, cout) in enumerate(blocks):
                h = self.resblock(h, zq, p + f"up.{lvl}.block.{j}.", cin, cout, clear_cache)
            if up:
                h = self.upsample(h, p + f"up.{lvl}.upsample", up == "space_time")
        h = swish(self.spatial_norm(h, zq, p + "norm_out"))
        return self.causal_conv(h, p + "conv_out", clear_cache)

    @torch.no_grad()
    def decode_latent(self, latent, stream_continue=False, stream_keep=False):
        """CogWrapper.decode_latent (dif_infer.py:245-271).  latent [1, C, T, h, w] -> [1, 3, 4T-3, 8h, 8w] fp32.
        Streaming: `stream_keep` keeps the conv caches after the last sub-chunk (clear_cache=False throughout,
        cp_enc_dec.py:436-466); `stream_continue` decodes T new latent frames in pairs against those caches -> 4T frames."""
        latent = 1.0 / self.cfg.scale_factor * latent
        T = latent.shape[2]
        if stream_continue:
            assert self.cache and T % 2 == 0
            spans = [(a, a + 2) for a in range(0, T, 2)]
        else:
            self.cache = {}
            spans = [((0, 3) if i == 0 else (i * 2 + 1, i * 2 + 3)) for i in range((T - 1) // 2)]
        recons = []
        for i, (a, b) in enumerate(spans):
            recons.append(self.decode_chunk(latent[:, :, a:b].contiguous(),
                                            clear_cache=(i == len(spans) - 1 and not stream_keep)))
        return torch.cat(recons, dim=2).to(torch.float32)


def post_process(video):
    """_post_process_cog_video (dif_infer.py:37-49): [-1,1] -> [0,1]."""
    return torch.clamp((video + 1.0) / 2.0, 0.0, 1.0)


def to_uint8_frames(video_cthw):
    """cthw_to_numpy_images (landiff/utils.py:327-331): *255, clip, uint8 TRUNCATION -> [T,H,W,C]."""
    img = video_cthw.permute(1, 2, 3, 0) * 255
    return img.clip(0, 255).to(torch.uint8)

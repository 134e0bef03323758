"""CogVideoX 3D-VAE decode on MI355X: channels-last activations, implicit-GEMM causal convs, HBM-resident halo caches.

Mirrors CogWrapper.decode_latent (landiff/diffusion/dif_infer.py:245-271: latent / 1.15258426, chunks of 3,2,2,...
latent frames, cache cleared on the last chunk) and ContextParallelDecoder3D.forward
(landiff/diffusion/vae_modules/cp_enc_dec.py:1034-1069) with ContextParallelCausalConv3d (:416-473),
SpatialNorm3D (:546-569), Upsample3D (:605-633), the resblock (:745-782), then _post_process_cog_video
(dif_infer.py:37-49) and the uint8 truncation of landiff/utils.py:327-331.

Differences in mechanism, not in results: the per-conv caches of the last two padded input frames stay in HBM
(the reference bounces them through host memory) -- in place: every causal conv owns a time-linear input buffer of two chunks'
length, a chunk's producer writes behind the previous chunk's last two frames, and the conv's window simply starts two frames
earlier (one two-frame copy when the buffer wraps, instead of a copy out and a copy in per conv and chunk); activations are [T][H][W][C] so every conv tap is a coalesced
128-byte row; conv_y/conv_b of SpatialNorm3D are evaluated at latent resolution and gathered with the nearest rule
(a 1x1x1 conv commutes with nearest upsampling); GroupNorm statistics stay per chunk, as in the reference.
"""
from __future__ import annotations

import os

import torch

from . import ops
from .config import VAEConfig
from .detokenizer import _conv_w, _dev
from .weights import vae_levels

BF = torch.bfloat16
ZQ_PAD = 64          # z/zq channels zero-padded 16 -> 64 for the MFMA K-tile


class VAEDecoder:
    def __init__(self, sd: dict, cfg: VAEConfig, device):
        self.cfg, self.dev = cfg, device
        self.w = {}
        for k, v in sd.items():
            if not k.startswith("decoder."):
                continue
            if k.endswith("weight") and v.dim() >= 4:
                if v.shape[-1] == 1 and v.shape[-2] == 1:            # 1x1x1 conv -> GEMM weight, K padded to 64
                    m = v.reshape(v.shape[0], v.shape[1])
                    if m.shape[1] < ZQ_PAD:
                        m = torch.nn.functional.pad(m, (0, ZQ_PAD - m.shape[1]))
                    self.w[k] = _dev(m, device)
                else:
                    self.w[k] = _conv_w(v, device, cin_pad=ZQ_PAD if v.shape[1] < ZQ_PAD else None)
            else:
                self.w[k] = _dev(v, device)
        # Causal-conv state (the reference's per-conv cache of the last two padded input frames): conv name -> position of those two
        # frames in the conv's own input buffer self._win[name] ([R][H+2][W+2][C], R = two chunks + 2 frames; 288 GB of HBM pay
        # for ~18 GB of such buffers at 480 x 720).  Empty = no state kept (fresh decode / cleared by the last chunk).
        self.cache = {}
        self._win = {}
        self._win_pos = {}          # where the window handed out last starts (between _conv_window and _causal_conv)
        # GroupNorm statistics from the producing convolution's epilogue (round 5) instead of a pass over the activation;
        # LD_VAE_GN_FUSE=0: the separate pass (A/B timing; same bf16 values summed in another order)
        self.fuse_gn_stats = os.environ.get("LD_VAE_GN_FUSE", "1") != "0"
        # Zero-bordered inputs of the upsamplers' 2D convs (no time halo, no state), one per shape, allocated and zero-filled
        # ONCE: the placement kernel rewrites the whole interior on every use, so the spatial borders stay zero; the decoder is
        # a chain -- a buffer's consumer (the conv) is queued before the next producer of that shape -- so one buffer per shape
        # is enough (stream order is the only hazard).  The causal convs' inputs live in self._win, zero-filled once the same way.
        self._padded = {}

    def _gn_ok(self, C: int) -> bool:
        """Can a convolution with C output channels leave GroupNorm partials (whole 4-channel quads per group, ld_conv_cl_bf16_gn)?"""
        q = C // 4
        return (self.fuse_gn_stats and C % 8 == 0 and 0 < q <= 256 and 256 % q == 0 and q % self.cfg.gn_groups == 0
                and self.cfg.gn_groups <= 64)

    def workspace_bytes(self) -> int:
        """HBM held by the conv input windows (self._win: 2T + 2 padded frames per causal conv) and the upsamplers' padded inputs
        (self._padded).  They are allocated and zero-filled on first use and KEPT -- across chunks (the halo lives in them),
        across decodes (no second zero-fill; a video's 6 chunks reuse them) and after a clear -- about 18 GB at 480 x 720, of the
        288 GB this path is laid out for.  A caller that needs the memory back between videos calls release()."""
        return sum(b.numel() * b.element_size() for b in list(self._win.values()) + list(self._padded.values()))

    def release(self) -> None:
        """Drop the window / padded-input buffers and any streaming state (the next decode allocates and zero-fills them again:
        ~18 GB of fills at 480 x 720).  Not to be called between decode(stream_keep=True) and its continuation."""
        self._win.clear(); self._win_pos.clear(); self._padded.clear(); self.cache = {}
        torch.cuda.empty_cache()

    def _padded_buf(self, *shape):
        buf = self._padded.get(shape)
        if buf is None:
            buf = self._padded[shape] = torch.zeros(*shape, device=self.dev, dtype=BF)
        return buf

    # ---- building blocks (x is a plain channels-last [T*H*W, C] tensor) ----------------------
    def _conv_window(self, name, T, H, W, C):
        """The zero-bordered input window [T+2][H+2][W+2][C] of causal conv `name` for this chunk -- the producer writes the T new
        frames at time offset 2.  A slice of the conv's own buffer that starts at the previous chunk's last two frames when
        there was one (self.cache[name]); when the window would run past the end, those two frames move to the front first."""
        shape = (H + 2, W + 2, C)
        buf = self._win.get(name)
        if buf is None or tuple(buf.shape[1:]) != shape or buf.shape[0] < T + 2:
            old, buf = buf, torch.zeros(2 * T + 2, *shape, device=self.dev, dtype=BF)
            if name in self.cache:                       # a longer chunk than any before, under a kept state: the halo moves along
                assert old is not None and tuple(old.shape[1:]) == shape, f"{name}: the frame shape changed under a live cache"
                buf[:2].copy_(old[self.cache[name]:self.cache[name] + 2])
                self.cache[name] = 0
            self._win[name] = buf
        pos = self.cache.get(name, 0)
        if pos + T + 2 > buf.shape[0]:
            src = buf[pos:pos + 2]
            buf[:2].copy_(src if pos >= 2 else src.clone())
            pos = self.cache[name] = 0
        self._win_pos[name] = pos
        return buf[pos:pos + T + 2]

    def _causal_conv(self, xp, name, T, H, W, clear, **epi):
        """xp: the window _conv_window(name, ...) with the current frames at time offset 2.  Its first two frames are the
        previous chunk's last two (already in place) or, on a fresh decode, replicas of the first frame; runs the conv and
        records where the next chunk's window starts (ContextParallelCausalConv3d.forward(x, clear_cache), cp_enc_dec.py:436-466).
        gn_partials=True: returns (out, GroupNorm partial sums of out) -- see _spatial_norm_swish."""
        if name not in self.cache:
            xp[0].copy_(xp[2]); xp[1].copy_(xp[2])
        if clear:
            self.cache.pop(name, None)
        else:
            self.cache[name] = self._win_pos[name] + T
        return ops.conv_cl(xp, self.w[name + ".conv.weight"], T, H, W, bias=self.w[name + ".conv.bias"], **epi)

    def _spatial_norm_swish(self, xg, name, T, H, W, C, zq, zshape, out):
        """swish(GN(x) * conv_y(zq) + conv_b(zq)) -> the interior of frames 2.. of `out` (zero-bordered [T+2][H+2][W+2][C], the
        consuming conv's window).  xg = (x, part): every GroupNorm of this
        decoder normalises a convolution's output, and the convolution's epilogue has already summed it (part: fp32 sums per
        64-row x 4-channel patch, ops.conv_cl(gn_partials=True)) -- the statistics cost a fold of those, not a read of x."""
        cfg, dev = self.cfg, self.dev
        x, part = xg
        zy = ops.gemm(zq, self.w[name + ".conv_y.conv.weight"], bias=self.w[name + ".conv_y.conv.bias"])
        zb = ops.gemm(zq, self.w[name + ".conv_b.conv.weight"], bias=self.w[name + ".conv_b.conv.bias"])
        stats = torch.empty(1, cfg.gn_groups, 2, device=dev, dtype=torch.float64)
        if part is None:
            ops.groupnorm_stats(x, stats, 1, T * H * W, C, cfg.gn_groups)
        else:
            ops.groupnorm_stats_from_conv(part, stats, T * H * W, C, cfg.gn_groups)
        ops.groupnorm_apply(x, out, stats, self.w[name + ".norm_layer.weight"], self.w[name + ".norm_layer.bias"],
                            1, T, H, W, C, cfg.gn_groups, zy=zy, zb=zb, zshape=zshape, tpad=2, hpad=1, wpad=1,
                            swish=True, eps=cfg.gn_eps)
        return out

    def _resblock(self, xg, p, cin, cout, T, H, W, zq, zshape, clear, norm_next=True):
        """xg = (x, GroupNorm partials of x or None) -> (out, partials of out when a norm consumes it next, else None)."""
        gn = self._gn_ok(cout)
        hp = self._spatial_norm_swish(xg, p + "norm1", T, H, W, cin, zq, zshape, self._conv_window(p + "conv1", T, H, W, cin))
        hg = self._causal_conv(hp, p + "conv1", T, H, W, clear, gn_partials=gn)
        del hp
        hp = self._spatial_norm_swish(hg if gn else (hg, None), p + "norm2", T, H, W, cout, zq, zshape,
                                      self._conv_window(p + "conv2", T, H, W, cout))
        del hg
        x = xg[0]
        if cin != cout:
            x = ops.gemm(x, self.w[p + "nin_shortcut.weight"], bias=self.w[p + "nin_shortcut.bias"])
        if gn and norm_next:
            return self._causal_conv(hp, p + "conv2", T, H, W, clear, resid=x, gn_partials=True)
        return self._causal_conv(hp, p + "conv2", T, H, W, clear, resid=x), None

    def _upsample(self, x, name, T, H, W, C, time_up):
        To = T
        if time_up and T > 1:
            To = 1 + 2 * (T - 1) if T % 2 == 1 else 2 * T
        xp = self._padded_buf(To, 2 * H + 2, 2 * W + 2, C)
        ops.place_cl(x, xp, 1, T, H, W, C, C, mode=1, time_up=time_up)
        out = ops.conv_cl(xp, self.w[name + ".conv.weight"], To, 2 * H, 2 * W, bias=self.w[name + ".conv.bias"],
                          gn_partials=self._gn_ok(C))
        return (out if self._gn_ok(C) else (out, None)), To, 2 * H, 2 * W

    # ---- one chunk ---------------------------------------------------------------------------
    def decode_chunk(self, z_cl: torch.Tensor, T: int, H: int, W: int, clear: bool) -> tuple:
        """z_cl [T*H*W, 64] bf16 channels-last latent chunk (already / scale_factor, zero-padded channels).
        Returns (rgb [T'*8H*8W, 8] bf16 with 3 valid channels, T', 8H, 8W)."""
        cfg = self.cfg
        zshape = (T, H, W)
        p = "decoder."
        xp = self._conv_window(p + "conv_in", T, H, W, ZQ_PAD)
        ops.place_cl(z_cl, xp, 1, T, H, W, ZQ_PAD, ZQ_PAD, mode=0, tpad=2)
        gn = self._gn_ok(self.w[p + "conv_in.conv.bias"].numel())
        h = self._causal_conv(xp, p + "conv_in", T, H, W, clear, gn_partials=gn)       # h: (activation, GroupNorm partials | None)
        if not gn:
            h = (h, None)
        top = h[0].shape[1]
        h = self._resblock(h, p + "mid.block_1.", top, top, T, H, W, z_cl, zshape, clear)
        h = self._resblock(h, p + "mid.block_2.", top, top, T, H, W, z_cl, zshape, clear)
        ch = top
        for lvl, blocks, up in vae_levels(cfg):
            for j, (cin, cout) in enumerate(blocks):
                # the last block of a level with an upsampler feeds the placement pass, not a norm
                h = self._resblock(h, p + f"up.{lvl}.block.{j}.", cin, cout, T, H, W, z_cl, zshape, clear,
                                   norm_next=not (up and j == len(blocks) - 1))
                ch = cout
            if up:
                h, T, H, W = self._upsample(h[0], p + f"up.{lvl}.upsample", T, H, W, ch, up == "space_time")
        hp = self._spatial_norm_swish(h, p + "norm_out", T, H, W, ch, z_cl, zshape, self._conv_window(p + "conv_out", T, H, W, ch))
        del h
        rgb = torch.empty(T * H * W, 8, device=self.dev, dtype=BF)
        self._causal_conv(hp, p + "conv_out", T, H, W, clear, out=rgb[:, : cfg.out_ch])
        return rgb, T, H, W

    # ---- whole latent ------------------------------------------------------------------------
    @torch.no_grad()
    def decode(self, latent: torch.Tensor, want_float: bool = False, stream_continue: bool = False,
               stream_keep: bool = False):
        """latent [1, T, C, h, w] fp32/bf16 (sampler output) -> uint8 frames [4T-3, 8h, 8w, 3] on the device
        (+ optionally the fp32 video [3, 4T-3, 8h, 8w] in [0,1]).

        Streaming: the causal-conv caches are the reference's cross-chunk state
        (ContextParallelCausalConv3d.forward(x, clear_cache), cp_enc_dec.py:436-466).  `stream_keep` leaves them in HBM after
        the last sub-chunk instead of clearing them; `stream_continue` decodes T *new* latent frames (T even) against the
        caches of the previous call in sub-chunks of two -> 4T frames."""
        cfg = self.cfg
        _, Tl, C, h, w = latent.shape
        lat32 = latent.float().contiguous()
        if stream_continue:
            assert self.cache, "stream_continue needs the caches of a previous decode(..., stream_keep=True)"
            assert Tl % 2 == 0, "a continued chunk decodes latent frames in pairs"
            spans = [(a, a + 2) for a in range(0, Tl, 2)]
            n_frames = 4 * Tl
        else:
            self.cache = {}
            spans = [((0, 3) if i == 0 else (i * 2 + 1, i * 2 + 3)) for i in range((Tl - 1) // 2)]
            # 9 frames from the first chunk, 8 from each later one; an even T leaves its last latent frame undecoded, as the
            # reference's loop does (dif_infer.py:253-259: "Must be 13, 11 or 9" in infer_cfgs/2b.yaml)
            n_frames = 1 + 8 * len(spans)
        P_total = n_frames * 8 * h * 8 * w
        frames = torch.empty(n_frames, 8 * h, 8 * w, 3, device=self.dev, dtype=torch.uint8)
        video = torch.empty(3, P_total, device=self.dev, dtype=torch.float32) if want_float else None
        f0 = 0
        for i, (a, b) in enumerate(spans):
            T = b - a
            z_cl = torch.empty(T * h * w, ZQ_PAD, device=self.dev, dtype=BF)
            ops.latent_to_cl(lat32[0, a:b], z_cl, T, C, h, w, ZQ_PAD, 1.0 / cfg.scale_factor, src_tchw=True)
            rgb, To, Ho, Wo = self.decode_chunk(z_cl, T, h, w, clear=(i == len(spans) - 1 and not stream_keep))
            P = To * Ho * Wo
            if want_float:   # chunk-local [3][P] then scattered into [3][frames] below
                tmp = torch.empty(3, P, device=self.dev, dtype=torch.float32)
                ops.to_uint8(rgb, frames[f0:f0 + To], tmp, P)
                video.view(3, n_frames, -1)[:, f0:f0 + To] = tmp.view(3, To, -1)
            else:
                ops.to_uint8(rgb, frames[f0:f0 + To], None, P)
            f0 += To
        assert f0 == n_frames
        if want_float:
            return frames, video.view(3, n_frames, 8 * h, 8 * w)
        return frames

"""Oracle: the whole per-prompt pipeline on CPU (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Follows landiff/infer_video.py:61-114 -> ArModelInferWrapper.forward (landiff/llm/llm_infer.py:74-105) ->
CogWrapper.forward (landiff/diffusion/dif_infer.py:152-243): tokens -> semantic condition -> sampler over
ControlDiffWarp -> chunked VAE decode -> post-process.
"""
from __future__ import annotations

import torch

from .dit import ControlDiTOracle
from .llm import LLMOracle
from .sampler import DiffusionSamplerOracle
from .tokenizer import DetokenizerOracle
from .vae import VAEDecoderOracle, post_process, to_uint8_frames


class PipelineOracle:
    def __init__(self, cfg, states, dtype=torch.bfloat16):
        self.cfg, self.dtype = cfg, dtype
        self.llm = LLMOracle(states["llm"], cfg.llm, dtype) if "llm" in states else None
        self.detok = DetokenizerOracle(states["tok"], states["ups"], cfg.tok, cfg.ups, dtype)
        self.dit = ControlDiTOracle(states["dit_main"], states["dit_control"], cfg.dit, dtype)
        self.sampler = DiffusionSamplerOracle(cfg.sampler)
        self.vae = VAEDecoderOracle(states["vae"], cfg.vae, dtype)

    @torch.no_grad()
    def tokens(self, text_emb, *, cfg_scale=7.5, motion_score=0.1, seed=42, multinomial_fn=None):
        g = torch.Generator().manual_seed(seed)
        return self.llm.sample(text_emb, motion_score=motion_score, num_frames=self.cfg.llm.segment_length,
                               guidance_scale=cfg_scale, generator=g, multinomial_fn=multinomial_fn).reshape(-1)

    @torch.no_grad()
    def latent(self, tokens, context, *, noise, randn_like=torch.randn_like, trace=None):
        """context [1, text_len, text_dim] (T5 states); uncond is the zero tensor (force_uc_zero_embeddings)."""
        sem = self.detok.semantic_cond(tokens.reshape(1, 1, -1))          # computed once per video
        net = lambda x, idx, ctx: self.dit(x, idx, ctx, sem)
        z = self.sampler.run(net, noise.clone(), context.float(), torch.zeros_like(context).float(), randn_like, trace)
        return z.to(self.dtype)                                           # diffusion_video.py:314

    @torch.no_grad()
    def frames(self, latent):
        rec = self.vae.decode_latent(latent.permute(0, 2, 1, 3, 4))       # b t c h w -> b c t h w
        video = post_process(rec)[0]                                      # [3, T, H, W] in [0,1]
        return video, to_uint8_frames(video)

    @torch.no_grad()
    def stream(self, tokens, context, *, n_chunks, prefix_frames, noises, randn_like=torch.randn_like):
        """The chunked long-video composition of the streaming primitives (see LanDiffPipeline.generate_stream): tokens
        [n_seg * num_latent_tokens] -> (latents per chunk, video [3, frames, H, W] in [0,1], uint8 frames)."""
        T = self.cfg.dit.latent_frames
        new = T - prefix_frames
        per_seg = self.cfg.tok.num_latent_tokens
        segs = tokens.reshape(-1, per_seg)
        sem_all = torch.cat([self.detok.semantic_cond(segs[s].reshape(1, 1, -1)) for s in range(segs.shape[0])], dim=1)
        ctx, uc = context.float(), torch.zeros_like(context).float()
        prev, lats, vids = None, [], []
        for c in range(n_chunks):
            sem = sem_all[:, c * new: c * new + T]
            net = lambda x, idx, cx: self.dit(x, idx, cx, sem)
            x0 = noises[c].clone()
            if c > 0:
                x0 = torch.cat([prev[:, T - prefix_frames:].float(), x0[:, prefix_frames:]], dim=1)   # diffusion_video.py:287-288
            z = self.sampler.run(net, x0, ctx, uc, randn_like, None, fixed_frames=0 if c == 0 else prefix_frames)
            prev = z.to(self.dtype)
            lats.append(prev)
            lat = prev if c == 0 else prev[:, prefix_frames:]
            rec = self.vae.decode_latent(lat.permute(0, 2, 1, 3, 4), stream_continue=c > 0, stream_keep=c < n_chunks - 1)
            vids.append(post_process(rec)[0])
        video = torch.cat(vids, dim=1)
        return lats, video, to_uint8_frames(video)

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

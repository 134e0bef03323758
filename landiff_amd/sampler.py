"""The diffusion sampling loop on the device (VPSDE DPM-Solver++(2M) SDE, or deterministic DDIM).

Mirrors SATVideoDiffusionEngine.sample (landiff/diffusion/diffusion_video.py:256-315) and
VPSDEDPMPP2MSampler / VideoDDIMSampler (landiff/diffusion/sgm/modules/diffusionmodules/sampling.py:538-837):
initial randn from the global device generator, two randn_like draws per step from step 2 on, last step returns
the denoised sample, result cast to bf16.  The state stays fp32 in HBM; the elementwise updates are HIP kernels,
torch supplies only the RNG (K10: kept for RNG-stream parity).
"""
from __future__ import annotations

import torch

from . import ops
from .config import SamplerConfig
from .schedule import build_plan


class DiffusionSampler:
    def __init__(self, cfg: SamplerConfig):
        self.cfg = cfg
        self.plan = build_plan(cfg)

    @torch.no_grad()
    def run(self, denoise_step, noise: torch.Tensor, randn_like=torch.randn_like, prefix=None, trace=None,
            fixed_frames: int = 0):
        """denoise_step(x, timestep, c_out, c_skip, cfg_scale, out) -> out (fp32, same shape as x).
        noise [1,T,C,H,W] fp32 on the device.  `prefix` latents overwrite the first frames (diffusion_video.py:287-288);
        `fixed_frames` > 0 pins the first frames to their initial value before every step and at the end
        (VPSDEDPMPP2MSampler.__call__, sampling.py:800-835, sdedit=False) -- the streaming primitive."""
        x = noise.clone()
        if prefix is not None:
            x[:, : prefix.shape[1]] = prefix
        pinned = x[:, :fixed_frames].clone() if fixed_frames > 0 else None
        den = torch.empty_like(x)
        den_d = torch.empty_like(x)
        old = torch.empty_like(x)
        have_old = False
        for sp in self.plan:
            if pinned is not None:
                x[:, :fixed_frames] = pinned                    # slab copy (plumbing)
            denoise_step(x, sp.timestep, sp.c_out, sp.c_skip, sp.cfg_scale, den)
            if trace is not None:
                trace.append((sp.index, sp.timestep, sp.cfg_scale))
            if self.cfg.sampler == "ddim":
                ops.axpbypcz(x, x, sp.a_t, den, sp.b_t)
                continue
            if sp.last:
                x.copy_(den)
                continue
            n1 = randn_like(x)
            if not have_old:
                ops.axpbypcz(x, x, sp.m1, den, -sp.m2, n1, sp.m_noise)
            else:
                n2 = randn_like(x)                              # x_standard's draw is discarded, as in the reference
                ops.axpbypcz(den_d, den, sp.m3, old, -sp.m4)
                ops.axpbypcz(x, x, sp.m1, den_d, -sp.m2, n2, sp.m_noise)
            old, den = den, old
            have_old = True
        if pinned is not None:
            x[:, :fixed_frames] = pinned
        return x

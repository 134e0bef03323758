"""Host-side noise schedule, denoiser preconditioning scalars, guidance scale and sampler multipliers.

Host logic only (fp32 torch CPU scalars evaluated in the reference's order, so the tables are bit-identical):
ZeroSNRDDPMDiscretization (landiff/diffusion/sgm/modules/diffusionmodules/discretizer.py:80-141),
DiscreteDenoiser sigma quantisation + VideoScaling (denoiser.py:43-77, denoiser_scaling.py:62-70),
DynamicCFG (guiders.py:60-79), VideoDDIMSampler / VPSDEDPMPP2MSampler coefficients (sampling.py:544-567,613-720).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch

from .config import SamplerConfig


def _alphas_cumprod(cfg: SamplerConfig) -> np.ndarray:
    betas = (torch.linspace(cfg.linear_start ** 0.5, cfg.linear_end ** 0.5, cfg.num_idx, dtype=torch.float64) ** 2).numpy()
    ac = np.cumprod(1.0 - betas, axis=0)
    return ac / (cfg.shift_scale + (1 - cfg.shift_scale) * ac)


def _zero_snr(ac: np.ndarray, n: int, num_idx: int):
    if n < num_idx:
        ts = np.linspace(num_idx - 1, 0, n, endpoint=False).astype(int)[::-1]
        sel = ac[ts]
    else:
        ts, sel = np.arange(num_idx), ac
    a = torch.tensor(sel, dtype=torch.float32).sqrt()
    a0, aT = a[0].clone(), a[-1].clone()
    a = (a - aT) * (a0 / (a0 - aT))
    return torch.flip(a, (0,)), ts


@dataclass
class StepPlan:
    """Everything one sampler step needs, as python floats (fp32-exact values)."""
    index: int
    timestep: int          # c_noise fed to the network
    c_skip: float
    c_out: float
    cfg_scale: float
    last: bool
    # DDIM
    a_t: float = 0.0
    b_t: float = 0.0
    # DPM++(2M) SDE
    m1: float = 0.0
    m2: float = 0.0
    m_noise: float = 0.0
    m3: float = 0.0        # only when a previous denoised exists
    m4: float = 0.0
    has_prev: bool = False


def build_plan(cfg: SamplerConfig) -> list[StepPlan]:
    ac = _alphas_cumprod(cfg)
    a, ts = _zero_snr(ac, cfg.num_steps, cfg.num_idx)
    a = torch.cat([a, a.new_ones([1])])
    timesteps = [-1] + [int(t) for t in ts]
    full, _ = _zero_snr(ac, cfg.num_idx, cfg.num_idx)
    den_sigmas = torch.flip(full, (0,))                     # DiscreteDenoiser.sigmas (flip=True)
    plan = []
    one = torch.ones(1)
    for i in range(cfg.num_steps):
        cur, nxt = one * a[i], one * a[i + 1]
        t = timesteps[-(i + 1)]
        q = den_sigmas[(cur - den_sigmas[:, None]).abs().argmin(dim=0)]       # sigma quantisation
        c_skip = float(q)
        c_out = float(-((1 - q ** 2) ** 0.5))
        scale = 1 + cfg.cfg_scale * (1 - math.cos(math.pi * ((cfg.num_steps - t) / cfg.num_steps) ** cfg.cfg_exp)) / 2
        sp = StepPlan(index=i, timestep=t, c_skip=c_skip, c_out=c_out, cfg_scale=scale,
                      last=(cfg.num_steps - i == 1))
        a_t = ((1 - nxt ** 2) / (1 - cur ** 2)) ** 0.5
        sp.a_t, sp.b_t = float(a_t), float(nxt - cur * a_t)
        if not sp.last:
            acur, anxt = cur ** 2, nxt ** 2
            lamb = ((acur / (1 - acur)) ** 0.5).log()
            lamb_next = ((anxt / (1 - anxt)) ** 0.5).log()
            h = lamb_next - lamb
            sp.m1 = float(((1 - nxt ** 2) / (1 - cur ** 2)) ** 0.5 * (-h).exp())
            sp.m2 = float((-2 * h).expm1() * nxt)
            sp.m_noise = float((1 - nxt ** 2) ** 0.5 * (1 - (-2 * h).exp()) ** 0.5)
            if i > 0:
                prev = one * a[i - 1]
                ap = prev ** 2
                lamb_prev = ((ap / (1 - ap)) ** 0.5).log()
                r = (lamb - lamb_prev) / h
                sp.m3, sp.m4, sp.has_prev = float(1 + 1 / (2 * r)), float(1 / (2 * r)), True
        plan.append(sp)
    return plan

"""Oracle: discretization, denoiser preconditioning, guidance and the diffusion samplers
(TEST INFRASTRUCTURE -- see oracle/__init__.py).

Follows landiff/diffusion/sgm/modules/diffusionmodules/discretizer.py:11-14,80-141,
util.py:20-33 (make_beta_schedule), denoiser.py:9-77, denoiser_scaling.py:62-70, guiders.py:22-79,
sampling.py:538-675 (VideoDDIMSampler), :678-837 (VPSDEDPMPP2MSampler), sampling_utils.py:8-13.
All scalars are fp32 torch tensors evaluated in the reference's order (bit-identical tables).
"""
from __future__ import annotations

import math

import numpy as np
import torch


def alphas_cumprod_table(linear_start=0.00085, linear_end=0.0120, num_timesteps=1000, shift_scale=3.0):
    betas = (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, num_timesteps, dtype=torch.float64) ** 2).numpy()
    ac = np.cumprod(1.0 - betas, axis=0)
    return ac / (shift_scale + (1 - shift_scale) * ac)      # SNR shift (discretizer.py:103-107)


def zero_snr_sigmas(n, ac_table, num_timesteps=1000):
    """ZeroSNRDDPMDiscretization.get_sigmas (discretizer.py:112-141): returns (flipped alpha_cumprod_sqrt fp32[n], timesteps)."""
    if n < num_timesteps:
        timesteps = np.linspace(num_timesteps - 1, 0, n, endpoint=False).astype(int)[::-1]
        ac = ac_table[timesteps]
    else:
        timesteps = np.arange(num_timesteps)
        ac = ac_table
    a = torch.tensor(ac, dtype=torch.float32).sqrt()
    a0, aT = a[0].clone(), a[-1].clone()
    a = a - aT
    a = a * (a0 / (a0 - aT))
    return torch.flip(a, (0,)), timesteps


def dynamic_cfg_scale(scale, exp, num_steps, step_index):
    """guiders.py:60-79 (python float64; step_index = num_steps - timestep, hugely negative)."""
    return 1 + scale * (1 - math.cos(math.pi * (step_index / num_steps) ** exp)) / 2


class DiffusionSamplerOracle:
    """VPSDEDPMPP2MSampler / VideoDDIMSampler around a DiscreteDenoiser with VideoScaling + DynamicCFG."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.ac = alphas_cumprod_table(cfg.linear_start, cfg.linear_end, cfg.num_idx, cfg.shift_scale)
        # DiscreteDenoiser.sigmas: num_idx points, flip=True of the already flipped table -> ascending index
        full, _ = zero_snr_sigmas(cfg.num_idx, self.ac, cfg.num_idx)
        self.denoiser_sigmas = torch.flip(full, (0,))

    def prepare(self):
        """VideoDDIMSampler.prepare_sampling_loop (sampling.py:544-567)."""
        a, timesteps = zero_snr_sigmas(self.cfg.num_steps, self.ac, self.cfg.num_idx)
        a = torch.cat([a, a.new_ones([1])])
        ts = torch.cat([torch.tensor(list(timesteps)).new_zeros([1]) - 1, torch.tensor(list(timesteps))])
        return a, ts

    def quantize_sigma(self, sigma):
        """DiscreteDenoiser.possibly_quantize_sigma (denoiser.py:63-71)."""
        dists = sigma - self.denoiser_sigmas[:, None]
        idx = dists.abs().argmin(dim=0).view(sigma.shape)
        return self.denoiser_sigmas[idx], idx

    def denoise(self, network, x, a_sqrt, timestep, cond, uc):
        """sampling.py:569-611 + Denoiser.forward (denoiser.py:25-41) + VideoScaling (:62-70)."""
        c = self.cfg
        xin = torch.cat([x] * 2)
        sig = torch.cat([a_sqrt] * 2)
        idx = torch.cat([x.new_ones([x.shape[0]]) * timestep] * 2)
        sig_q, _ = self.quantize_sigma(sig)
        s = sig_q.view(-1, *([1] * (x.ndim - 1)))
        c_skip, c_out = s, -((1 - s ** 2) ** 0.5)
        ctx = torch.cat([uc, cond], 0)                     # VanillaCFG.prepare_inputs: [uncond, cond]
        out = network(xin * torch.ones_like(s), idx, ctx)
        den = (out * c_out + xin * c_skip).to(torch.float32)
        x_u, x_c = den.chunk(2)
        scale = dynamic_cfg_scale(c.cfg_scale, c.cfg_exp, c.num_steps, (c.num_steps - timestep).item())
        return x_u + scale * (x_c - x_u), scale

    def run(self, network, x, cond, uc, randn_like=torch.randn_like, trace=None, fixed_frames=0):
        """network(x[2B,...] fp32, idx[2B], ctx[2B,...]) -> eps-like output (any float dtype)."""
        c = self.cfg
        a, ts = self.prepare()
        n = len(a)
        s_in = x.new_ones([x.shape[0]])
        old = None
        prefix_frames = x[:, :fixed_frames] if fixed_frames > 0 else None     # sampling.py:800-801
        for i in range(n - 1):
            if fixed_frames > 0:                                              # :803-817 (sdedit=False branch)
                x = torch.cat([prefix_frames, x[:, fixed_frames:]], dim=1)
            cur, nxt = s_in * a[i], s_in * a[i + 1]
            prev = None if i == 0 else s_in * a[i - 1]
            timestep = ts[-(i + 1)]
            den, scale = self.denoise(network, x, cur, timestep, cond, uc)
            if trace is not None:
                trace.append(dict(i=i, timestep=int(timestep), scale=scale))
            idx = c.num_steps - i
            if c.sampler == "ddim":                         # sampling.py:613-644
                a_t = ((1 - nxt ** 2) / (1 - cur ** 2)) ** 0.5
                b_t = nxt - cur * a_t
                x = _ap(a_t, x) * x + _ap(b_t, x) * den
                continue
            if idx == 1:                                    # last step returns denoised (:750-751)
                x = den
                old = den
                continue
            ac, acn = cur ** 2, nxt ** 2                    # get_variables :679-703
            lamb = ((ac / (1 - ac)) ** 0.5).log()
            lamb_next = ((acn / (1 - acn)) ** 0.5).log()
            h = lamb_next - lamb
            m1 = ((1 - nxt ** 2) / (1 - cur ** 2)) ** 0.5 * (-h).exp()
            m2 = (-2 * h).expm1() * nxt
            mn = (1 - nxt ** 2) ** 0.5 * (1 - (-2 * h).exp()) ** 0.5
            x_std = _ap(m1, x) * x - _ap(m2, x) * den + _ap(mn, x) * randn_like(x)
            if old is None or torch.sum(nxt) < 1e-14:
                x, old = x_std, den
                continue
            acp = prev ** 2
            lamb_prev = ((acp / (1 - acp)) ** 0.5).log()
            r = (lamb - lamb_prev) / h
            m3, m4 = 1 + 1 / (2 * r), 1 / (2 * r)
            den_d = _ap(m3, x) * den - _ap(m4, x) * old
            x = _ap(m1, x) * x - _ap(m2, x) * den_d + _ap(mn, x) * randn_like(x)   # second draw (:778)
            old = den
        if fixed_frames > 0:                                                  # :834-835
            x = torch.cat([prefix_frames, x[:, fixed_frames:]], dim=1)
        return x


def _ap(v, x):
    return v.view(-1, *([1] * (x.ndim - 1)))

"""Shared helpers of the oracle (dtype policy, small functional ops)."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def linear(x, w, b=None, dtype=torch.float32):
    """nn.Linear under the reference's autocast/bf16 module regime: operands cast to `dtype`."""
    return F.linear(x.to(dtype), w.to(dtype), None if b is None else b.to(dtype))


def layer_norm(x, w, b, eps, out_dtype=None):
    """nn.LayerNorm; statistics in fp32 (what both the CPU and the GPU kernels do internally)."""
    y = F.layer_norm(x.float(), (x.shape[-1],), None if w is None else w.float(),
                     None if b is None else b.float(), eps)
    return y.to(out_dtype if out_dtype is not None else x.dtype)


def timestep_embedding(t, dim, max_period=10000, dtype=torch.float32):
    """landiff/diffusion/sgm/modules/diffusionmodules/util.py:207-233 (cos || sin, fp32 then cast)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb.to(dtype)


def swish(x):
    return x * torch.sigmoid(x)


def group_norm(x, groups, w, b, eps):
    """torch.nn.GroupNorm on [N, C, ...]; fp32 statistics, output in x.dtype."""
    return F.group_norm(x.float(), groups, w.float(), b.float(), eps).to(x.dtype)

"""GPU parity of the weight-streaming GEMV (the LLM decode linears, transformer_blocks.py:22-40,128-236, and the
batch-2 conditioning linears of the DiT) against a plain torch fp32 reference that rounds to bf16 at the same points
as the reference's bf16 modules: normed x, Linear output, activation output, gate product, residual sum."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def _act(name, t):
    if name is None:
        return t
    if name == "gelu_tanh":
        return torch.nn.functional.gelu(t, approximate="tanh")
    if name == "silu":
        return torch.nn.functional.silu(t)
    raise AssertionError(name)


def _ref(x, w, *, w2=None, bias=None, resid=None, in_act=None, act=None, norm_w=None, eps=0.0, out_f32=False):
    rb = lambda t: t.to(BF).float()
    xf = x.float()
    if in_act:
        xf = rb(_act(in_act, xf))
    if norm_w is not None:
        xf = rb(xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps) * norm_w.float())
    y = xf.double() @ w.double().t()
    if bias is not None:
        y = y + bias.double()
    y = rb(y.float())
    if act:
        y = rb(_act(act, y))
    if w2 is not None:
        y = rb(y * rb((xf.double() @ w2.double().t()).float()))
    if resid is not None:
        y = resid.float() + y
        if not out_f32:
            y = rb(y)
    return y


def _close(out, ref):
    # one bf16 ulp of slack where the fp32 sums straddle a rounding boundary, plus the sum-order noise near zero
    err = (out.float() - ref).abs()
    tol = 2.0 ** -7 * ref.abs() + 2e-3 * ref.abs().max()
    assert bool((err <= tol).all()), f"max err {err.max().item()} at ref scale {ref.abs().max().item()}"


SHAPES = [  # (B, N, K, form)
    (2, 6144, 2048, "norm"), (2, 2048, 2048, "resid"), (2, 11008, 2048, "gated"), (2, 2048, 11008, "resid"),
    (1, 515, 2048, "norm"), (3, 1000, 1024, "bias_act"), (4, 129, 4096, "gated"), (2, 77, 8200, "resid"),
    (1, 64, 512, "in_act"), (4, 3072, 512, "in_act"), (2, 23040, 512, "in_act"), (3, 4099, 256, "in_act"), (2, 5000, 384, "bias_act"), (2, 30, 24576, "bias_act"), (3, 200, 11008, "resid"),
]


@pytest.mark.parametrize("B,N,K,form", SHAPES)
def test_gemv_forms(cuda, B, N, K, form):
    from landiff_amd import ops
    g = torch.Generator(device="cpu").manual_seed(B * 1000003 + N * 31 + K)
    x = torch.randn(B, K, generator=g).to(cuda, BF)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(cuda, BF)
    w[:, 0] += (torch.arange(N, device=cuda) % 7).to(BF) * 0.05          # row-dependent structure: catches row mix-ups
    kw, rkw = {}, {}
    if form in ("norm", "gated"):
        nw = (1.0 + 0.1 * torch.randn(K, generator=g)).to(cuda)
        kw.update(norm_w=nw, norm_eps=1e-5); rkw.update(norm_w=nw, eps=1e-5)
    if form == "gated":
        w2 = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(cuda, BF)
        kw.update(w2=w2, act="gelu_tanh"); rkw.update(w2=w2, act="gelu_tanh")
    if form == "resid":
        r = torch.randn(B, N, generator=g).to(cuda, BF)
        kw.update(resid=r); rkw.update(resid=r)
    if form == "bias_act":
        b = torch.randn(N, generator=g).to(cuda, BF)
        kw.update(bias=b, act="silu"); rkw.update(bias=b, act="silu")
    if form == "in_act":
        b = torch.randn(N, generator=g).to(cuda, BF)
        kw.update(bias=b, in_act="silu"); rkw.update(bias=b, in_act="silu")
    out = torch.full((B, N), float("nan"), device=cuda, dtype=BF)
    ops.gemv(x, w, out, **kw)
    _close(out, _ref(x, w, **rkw))


def test_gemv_inplace_residual_and_strided_rows(cuda):
    """The decode loop's wo / w2 calls: out aliases resid; x is a row-strided view."""
    from landiff_amd import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    xs = torch.randn(2, 4096, generator=g).to(cuda, BF)
    x = xs[:, :2048]
    w = (torch.randn(2048, 2048, generator=g) / 45.0).to(cuda, BF)
    h = torch.randn(2, 2048, generator=g).to(cuda, BF)
    ref = _ref(x, w, resid=h.clone())
    ops.gemv(x, w, h, resid=h)
    _close(h, ref)


def test_gemv_fp32_head(cuda):
    """fp32 weights / activations / logits (the LM head, lm_model.py:417-454) stay on the fp32 streaming kernel."""
    from landiff_amd import ops
    g = torch.Generator(device="cpu").manual_seed(6)
    x = torch.randn(2, 2048, generator=g).to(cuda)
    w = (torch.randn(2055, 2048, generator=g) / 45.0).to(cuda)
    out = torch.empty(2, 2055, device=cuda)
    ops.gemv(x, w, out)
    ref = (x.double() @ w.double().t()).float()
    assert (out - ref).abs().max().item() < 1e-4

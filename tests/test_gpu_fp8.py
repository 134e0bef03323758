"""GPU parity of the optional fp8 (OCP e4m3) linear path (BASELINE configs[4]): the quantiser against torch's
float8_e4m3fn conversion, the MFMA kernel against an fp64 product of the SAME quantised operands (only the fp32
accumulation order differs), and the quantise -> GEMM chain against the bf16 GEMM within the format's noise."""
import pytest
import torch

pytestmark = pytest.mark.gpu

BF = torch.bfloat16
F8 = torch.float8_e4m3fn


def _deq(q):
    return q.cpu().view(F8).float()


@pytest.mark.parametrize("M,K", [(5, 128), (300, 1920), (1000, 7680), (17, 8192)])
def test_quantize_fp8_rows(cuda, M, K):
    from landiff_amd import ops
    g = torch.Generator().manual_seed(M + K)
    x = (torch.randn(M, K, generator=g) * torch.logspace(-3, 2, M)[:, None]).to(BF)
    x[M // 2] = 0                                                    # all-zero row: scale 1, codes 0
    q, s = ops.quantize_fp8(x.to(cuda))
    amax = x.float().abs().amax(1)
    want_s = torch.where(amax > 0, amax * (1.0 / 448.0), torch.ones_like(amax))
    assert torch.allclose(s.cpu(), want_s, rtol=1e-6, atol=0)
    want_q = (x.float() * (1.0 / s.cpu())[:, None]).clamp(-448, 448).to(F8).view(torch.uint8)
    assert torch.equal(q.cpu(), want_q)
    assert not bool(torch.isnan(_deq(q)).any())


@pytest.mark.parametrize("M,N,K,form", [(256, 256, 128, "plain"), (300, 200, 256, "bias"), (1000, 1920, 1920, "gelu"),
                                         (2500, 5760, 1920, "bias"), (777, 1920, 7680, "resid")])
def test_gemm_fp8_exact_operands(cuda, M, N, K, form):
    from landiff_amd import ops
    g = torch.Generator().manual_seed(M * 3 + N)
    a8 = torch.randn(M, K, generator=g).clamp(-3, 3).to(F8)
    w8 = (torch.randn(N, K, generator=g) * 0.5).to(F8)
    a8.view(torch.uint8)[:, 0] = torch.tensor([0x38], dtype=torch.uint8)      # column of ones: row/col structure below
    sa = torch.rand(M, generator=g) + 0.5
    sw = torch.rand(N, generator=g) * 0.02 + 0.01
    sa[::7] *= 3.0
    ref = (a8.float().double() * sa.double()[:, None]) @ (w8.float().double() * sw.double()[:, None]).t()
    kw, post = {}, (lambda r: r)
    if form in ("bias", "gelu", "resid"):
        bias = torch.randn(N, generator=g).to(BF)
        kw["bias"] = bias.to(cuda)
        ref = ref + bias.double()
    if form == "gelu":
        kw["act"] = "gelu_tanh"
        post = lambda r: torch.nn.functional.gelu(r.float().to(BF).float(), approximate="tanh")
    if form == "resid":
        res = torch.randn(M, N, generator=g).to(BF)
        kw["resid"] = res.to(cuda)
        post = lambda r: res.float() + r.float().to(BF).float()
    out = ops.gemm_fp8(a8.view(torch.uint8).to(cuda), sa.to(cuda), w8.view(torch.uint8).to(cuda), sw.to(cuda), **kw)
    want = post(ref.float())
    err = (out.float().cpu() - want).abs().max().item() / (want.abs().max().item() + 1e-6)
    assert err < 1e-2, err                                           # bf16 output rounding only
    outf = ops.gemm_fp8(a8.view(torch.uint8).to(cuda), sa.to(cuda), w8.view(torch.uint8).to(cuda), sw.to(cuda),
                        out_f32=True) if form == "plain" else None
    if outf is not None:                                             # fp32 output still rounds the accumulator to bf16
        assert (outf.cpu() - ref.float()).abs().max().item() / ref.abs().max().item() < 1e-2


def test_fp8_chain_vs_bf16_gemm(cuda):
    """quantize_fp8 -> gemm_fp8 (per-row x per-channel scales) against ld_gemm_bf16 on the same bf16 operands."""
    from landiff_amd import ops
    g = torch.Generator().manual_seed(3)
    M, N, K = 4096, 1920, 1920
    a = torch.randn(M, K, generator=g).to(cuda, BF)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(cuda, BF)
    w8, sw = ops.quantize_fp8(w)                                     # weights: one scale per output channel
    a8, sa = ops.quantize_fp8(a)
    out8 = ops.gemm_fp8(a8, sa, w8, sw).float()
    out16 = ops.gemm(a, w).float()
    rel = ((out8 - out16).norm() / out16.norm()).item()
    assert rel < 0.05, rel                                           # e4m3: 3 mantissa bits on both operands
    assert rel > 1e-4                                                # ... and it really is the fp8 path

"""GPU parity of the MFMA GEMM / implicit-GEMM conv kernels against a plain torch fp32 reference."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-6)).item()


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 192), (35552 // 8, 1920, 1920), (77, 2055, 128), (1, 64, 64)])
def test_gemm_plain(cuda, M, N, K):
    from landiff_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N)
    a = torch.randn(M, K, generator=g).to(cuda, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(cuda, torch.bfloat16)
    # asymmetric check: distinct row/col structure catches transposed C writes
    a[:, 0] += torch.arange(M, device=cuda).to(torch.bfloat16) * 0.01
    out = ops.gemm(a, w)
    ref = a.float() @ w.float().t()
    assert _rel(out, ref) < 1e-2
    outf = ops.gemm(a, w, out_f32=True)
    # fp32 output still rounds the accumulator to bf16 (bf16 Linear output), so same tolerance
    assert _rel(outf, ref) < 1e-2


def test_gemm_epilogue_gated_residual(cuda):
    from landiff_amd import ops
    B, rows, text, N, K = 2, 200, 26, 256, 128
    M = B * rows
    g = torch.Generator(device="cpu").manual_seed(1)
    a = torch.randn(M, K, generator=g).to(cuda, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.1).to(cuda, torch.bfloat16)
    bias = torch.randn(N, generator=g).to(cuda, torch.bfloat16)
    h = torch.randn(M, N, generator=g).to(cuda, torch.bfloat16)
    ada = torch.randn(B, 12 * N, generator=g).to(cuda, torch.bfloat16)
    add2 = torch.randn(M, N, generator=g).to(cuda, torch.bfloat16)
    out = ops.gemm(a, w, bias=bias, resid=h, gate=ada, gate_bstride=12 * N, gate_off_img=2 * N,
                   gate_off_txt=8 * N, rows_per_batch=rows, text_len=text, add2=add2)
    y = (a.float() @ w.float().t() + bias.float()).to(torch.bfloat16)
    gate = torch.empty(M, N, device=cuda, dtype=torch.bfloat16)
    for b in range(B):
        gate[b * rows: b * rows + text] = ada[b, 8 * N: 9 * N]
        gate[b * rows + text: (b + 1) * rows] = ada[b, 2 * N: 3 * N]
    ref = (h + gate * y) + add2
    assert _rel(out, ref) < 2e-2
    # activation + mul path
    mul = torch.randn(M, N, generator=g).to(cuda, torch.bfloat16)
    out2 = ops.gemm(a, w, act="gelu_tanh", mul=mul)
    ref2 = torch.nn.functional.gelu((a.float() @ w.float().t()).to(torch.bfloat16).float(), approximate="tanh").to(torch.bfloat16) * mul
    assert _rel(out2, ref2) < 2e-2
    # fp32 residual stream
    hf = h.float()
    out3 = ops.gemm(a, w, bias=bias, resid=hf, out_f32=True)
    ref3 = hf + y.float()
    assert _rel(out3, ref3) < 1e-2


@pytest.mark.parametrize("kT,kH,kW,T,H,W,Cin,Cout", [(3, 3, 3, 3, 12, 20, 64, 128), (1, 3, 3, 2, 9, 7, 128, 64), (3, 3, 3, 2, 6, 5, 64, 3)])
def test_conv_cl(cuda, kT, kH, kW, T, H, W, Cin, Cout):
    from landiff_amd import ops
    g = torch.Generator(device="cpu").manual_seed(3)
    x = torch.randn(1, Cin, T + kT - 1, H, W, generator=g).to(cuda, torch.bfloat16)  # time halo included
    w = (torch.randn(Cout, Cin, kT, kH, kW, generator=g) * 0.05).to(cuda, torch.bfloat16)
    bias = torch.randn(Cout, generator=g).to(cuda, torch.bfloat16)
    ref = torch.nn.functional.conv3d(x.float(), w.float(), bias.float(), padding=(0, kH // 2, kW // 2))  # [1,Cout,T,H,W]
    xp = torch.zeros(T + kT - 1, H + kH - 1, W + kW - 1, Cin, device=cuda, dtype=torch.bfloat16)
    xp[:, kH // 2: kH // 2 + H, kW // 2: kW // 2 + W] = x[0].permute(1, 2, 3, 0)
    wcl = w.permute(0, 2, 3, 4, 1).contiguous()
    out = ops.conv_cl(xp, wcl, T, H, W, bias=bias)
    ref_cl = ref[0].permute(1, 2, 3, 0).reshape(T * H * W, Cout)
    assert _rel(out, ref_cl) < 1e-2

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


@pytest.mark.parametrize("T,H,W,Cout", [(1, 4, 16, 3), (3, 8, 48, 3), (9, 12, 32, 4), (2, 4, 16, 1), (8, 64, 96, 3)])
def test_conv_narrow_output_route(cuda, monkeypatch, T, H, W, Cout):
    """3x3x3 convolutions with <= 4 output channels over 128 input channels (the VAE's conv_out) take ld_conv_narrow.hip: a rolling
    three-frame halo window in LDS, K split over the four waves.  Against torch fp32 and against the implicit-GEMM route
    (LD_CONV_NARROW=0, re-read per call under LD_TUNING=1): a different summation order, so the two routes agree to the bf16
    rounding of the output, not bit for bit; the kernel itself repeats exactly; columns past Cout of the output rows stay untouched."""
    from landiff_amd import _lib, ops
    Cin = 128
    assert _lib.load().ld_conv_route(T, H, W, Cin, Cout, 3, 3, 3) == 5      # ROUTE_NARROW
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(1, Cin, T + 2, H, W, generator=g).to(cuda, torch.bfloat16)
    w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) * 0.05).to(cuda, torch.bfloat16)
    bias = torch.randn(Cout, generator=g).to(cuda, torch.bfloat16)
    ref = torch.nn.functional.conv3d(x.float(), w.float(), bias.float(), padding=(0, 1, 1))[0].permute(1, 2, 3, 0).reshape(T * H * W, Cout)
    xp = torch.randn(T + 2, H + 2, W + 2, Cin, generator=g).to(cuda, torch.bfloat16) * 0        # zero border
    xp[:, 1:1 + H, 1:1 + W] = x[0].permute(1, 2, 3, 0)
    wcl = w.permute(0, 2, 3, 4, 1).contiguous()
    buf = torch.full((T * H * W, 8), 7.0, device=cuda, dtype=torch.bfloat16)                       # the VAE's rgb buffer: 8 columns per row
    out = ops.conv_cl(xp, wcl, T, H, W, bias=bias, out=buf[:, :Cout])
    assert (buf[:, Cout:] == 7.0).all()
    assert _rel(out, ref) < 1e-2
    again = ops.conv_cl(xp, wcl, T, H, W, bias=bias).clone()
    assert torch.equal(again, out.contiguous())
    monkeypatch.setenv("LD_CONV_NARROW", "0")
    gemm_route = ops.conv_cl(xp, wcl, T, H, W, bias=bias)
    monkeypatch.delenv("LD_CONV_NARROW")
    d = (out.float() - gemm_route.float()).abs()
    assert d.max().item() <= 2.0 ** -7 * ref.abs().max().item() and (d > 0).float().mean().item() < 0.2      # at most one bf16 step, on few elements
    # shapes outside the narrow kernel keep the GEMM route
    assert _lib.load().ld_conv_route(T, H + 1, W, Cin, Cout, 3, 3, 3) != 5 and _lib.load().ld_conv_route(T, H, W, 256, Cout, 3, 3, 3) != 5


@pytest.mark.parametrize("T,H,W,Cin,Cout,route,resid", [(2, 12, 20, 64, 128, 0, True), (1, 8, 8, 64, 512, 0, False),
                                                         (3, 30, 44, 128, 256, 0, True), (2, 256, 264, 256, 256, 2, True),
                                                         (2, 256, 264, 256, 256, 2, False)])
def test_conv_groupnorm_partials(cuda, T, H, W, Cin, Cout, route, resid):
    """ld_conv_cl_bf16_gn: the convolution's epilogue also sums the bf16 values it stores (per 64-row x 4-channel patch), and
    ld_groupnorm_stats_from_conv folds those into the GroupNorm statistics -- the VAE's norms no longer read their input twice
    (cp_enc_dec.py:546-569 inside :745-782).  Output bit-identical to the plain launch; statistics equal to float64 sums over the
    output (and to ld_groupnorm_stats) to fp32-summation accuracy; repeatable bit for bit.  Shapes: both kernels (128 x 128
    two-stage, 256 x 256 8-phase), both epilogues (bias / bias + residual), row counts that are not multiples of 64 or of the
    tile, 4 / 8 / 16 channels per group."""
    from landiff_amd import _lib, ops
    assert _lib.load().ld_conv_route(T, H, W, Cin, Cout, 3, 3, 3) == route
    G, M = 32, T * H * W
    g = torch.Generator(device="cpu").manual_seed(11)
    xp = torch.zeros(T + 2, H + 2, W + 2, Cin, device=cuda, dtype=torch.bfloat16)
    xp[:, 1:1 + H, 1:1 + W] = torch.randn(T + 2, H, W, Cin, generator=g).to(cuda, torch.bfloat16)
    wcl = (torch.randn(Cout, 3, 3, 3, Cin, generator=g) * 0.03).to(cuda, torch.bfloat16)
    bias = torch.randn(Cout, generator=g).to(cuda, torch.bfloat16)
    epi = dict(bias=bias)
    if resid:
        epi["resid"] = (torch.randn(M, Cout, generator=g) + 0.5).to(cuda, torch.bfloat16)
    plain = ops.conv_cl(xp, wcl, T, H, W, **epi)
    out, part = ops.conv_cl(xp, wcl, T, H, W, gn_partials=True, **epi)
    assert torch.equal(out, plain)
    assert part.numel() == (M + 63) // 64 * (Cout // 4) * 2 and torch.isfinite(part).all()
    stats = torch.full((1, G, 2), float("nan"), device=cuda, dtype=torch.float64)
    ops.groupnorm_stats_from_conv(part, stats, M, Cout, G)
    o64 = out.double().view(M, G, Cout // G)
    want = torch.stack([o64.sum(dim=(0, 2)), (o64 * o64).sum(dim=(0, 2))], dim=-1)
    scale = torch.stack([o64.abs().sum(dim=(0, 2)), (o64 * o64).sum(dim=(0, 2))], dim=-1)      # the sums' own magnitude (sum of |x|: the mean may cancel)
    assert ((stats[0] - want).abs() / scale).max().item() < 2e-6
    old = torch.empty(1, G, 2, device=cuda, dtype=torch.float64)
    ops.groupnorm_stats(out, old, 1, M, Cout, G)
    assert ((stats[0] - old[0]).abs() / scale).max().item() < 2e-6
    # the patch sums themselves, against float64 sums of the same patches
    U = (M + 63) // 64
    padded = torch.zeros(U * 64, Cout, device=cuda, dtype=torch.float64)
    padded[:M] = out.double()
    p64 = padded.view(U, 64, Cout // 4, 4)
    want_part = torch.stack([p64.sum(dim=(1, 3)), (p64 * p64).sum(dim=(1, 3))], dim=-1)
    assert (part.view(U, Cout // 4, 2).double() - want_part).abs().max().item() < 1e-4 * max(1.0, want_part.abs().max().item())
    out2, part2 = ops.conv_cl(xp, wcl, T, H, W, gn_partials=True, **epi)
    stats2 = torch.empty_like(stats)
    ops.groupnorm_stats_from_conv(part2, stats2, M, Cout, G)
    assert torch.equal(part2, part) and torch.equal(stats2, stats)
    # refused where the epilogue cannot sum whole 8-channel rows
    with pytest.raises(Exception):
        ops.conv_cl(xp, wcl[:12].contiguous(), T, H, W, gn_partials=True, bias=bias[:12].contiguous())


@pytest.mark.parametrize("B,N,H,K", [(2, 456, 3, 128), (1, 1000, 2, 192), (2, 4440, 5, 320)])
def test_gemm_qkv_heads_fused_split(cuda, B, N, H, K):
    """ld_gemm_qkv_heads (qkv Linear with QK-LayerNorm / head split / V transpose in its epilogue) against the two-launch
    path it replaces (ld_gemm_bf16 + ld_qkv_split) and against a torch fp32 restatement of dit_video_concat.py:636-653.
    Shapes: token counts that are not multiples of the 128-row wave tile (batch boundary inside a tile), a column count
    that leaves dead waves in the last tile column, and one large enough for the 256x256 kernel + 128x128 tail launch."""
    from landiff_amd import ops
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + N)
    Npad = (N + 127) // 128 * 128
    a = torch.randn(B * N, K, generator=g).to(cuda, torch.bfloat16)
    w = (torch.randn(3 * H * 64, K, generator=g) * 0.08).to(cuda, torch.bfloat16)
    bias = torch.randn(3 * H * 64, generator=g).to(cuda, torch.bfloat16)
    ln = tuple((torch.randn(64, generator=g) * s + o).to(cuda, torch.bfloat16) for s, o in ((0.2, 1.0), (0.2, 0.0), (0.2, 1.0), (0.2, 0.0)))
    eps = 1e-6
    mk = lambda: (torch.zeros(B, H, Npad, 64, device=cuda, dtype=torch.bfloat16), torch.zeros(B, H, Npad, 64, device=cuda, dtype=torch.bfloat16),
                  torch.zeros(B, H, 64, Npad, device=cuda, dtype=torch.bfloat16))
    q1, k1, v1 = mk()
    ops.gemm_qkv_heads(a, w, bias, q1, k1, v1, B, N, H, Npad, ln, eps=eps)
    q2, k2, v2 = mk()
    qkv = ops.gemm(a, w, bias=bias)
    ops.qkv_split(qkv, q2, k2, v2, B, N, H, Npad, ln=ln, eps=eps)
    # V is a pure data movement of the same bf16 values: exact.  q / k: the same fp32 LayerNorm on the same bf16 inputs
    # (a fused multiply-add may contract differently in the two kernels: one bf16 ulp)
    assert torch.equal(v1, v2)
    for x1, x2 in ((q1, q2), (k1, k2)):
        d = (x1.float() - x2.float()).abs()
        assert (d <= 2.0 ** -7 * x2.float().abs() + 1e-6).all(), d.max().item()
        assert (x1 != x2).float().mean().item() < 1e-3
    assert float(q1[:, :, N:].abs().max()) == 0.0 and float(v1[:, :, :, N:].abs().max()) == 0.0     # padding rows untouched
    # independent reference
    y = (a.float() @ w.float().t() + bias.float()).to(torch.bfloat16).float().view(B, N, 3, H, 64)
    lnf = lambda t, wv, bv: torch.nn.functional.layer_norm(t, (64,), wv.float(), bv.float(), eps)
    qr = lnf(y[:, :, 0], ln[0], ln[1]).permute(0, 2, 1, 3)
    kr = lnf(y[:, :, 1], ln[2], ln[3]).permute(0, 2, 1, 3)
    vr = y[:, :, 2].permute(0, 2, 3, 1)
    assert _rel(q1[:, :, :N], qr) < 1.5e-2 and _rel(k1[:, :, :N], kr) < 1.5e-2 and _rel(v1[:, :, :, :N], vr) < 1e-2

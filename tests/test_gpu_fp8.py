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


# ---------------------------------------------------------------------------------------------- MXFP8
def _mx_quant_ref(x):
    """MX container reference: per 32 elements one E8M0 scale = the smallest power of two >= amax / 448 (fp32 arithmetic as
    in the kernel), elements = e4m3 cast of x / scale."""
    M, K = x.shape
    xb = x.float().view(M, K // 32, 32)
    amax = xb.abs().amax(-1)
    tb = (amax * torch.tensor(1.0 / 448.0, dtype=torch.float32)).view(torch.int32)
    sb = ((tb >> 23) & 0xff) + ((tb & 0x7fffff) != 0).to(torch.int32)
    sb = torch.where(amax > 0, sb.clamp(1, 254), torch.zeros_like(sb))
    inv = torch.pow(2.0, (127 - sb).float())
    q = (xb * inv[..., None]).clamp(-448, 448).to(F8).view(torch.uint8).view(M, K)
    return q, sb.to(torch.uint8)


def _tile_major(sb):
    """[M, K/32] scale bytes -> the kernels' K-tile-major layout [K/128, M, 4]."""
    M, nb = sb.shape
    return sb.view(M, nb // 4, 4).permute(1, 0, 2).contiguous()


def _row_major(st):
    return st.permute(1, 0, 2).reshape(st.shape[1], -1)


def _mx_deq(q, sb):
    M, K = q.shape
    return (q.view(F8).float().view(M, K // 32, 32) * torch.pow(2.0, sb.float() - 127)[..., None]).view(M, K)


@pytest.mark.parametrize("M,K", [(7, 128), (300, 1920), (129, 7680)])
def test_quantize_mxfp8(cuda, M, K):
    from landiff_amd import ops
    g = torch.Generator().manual_seed(M + K)
    x = (torch.randn(M, K, generator=g) * torch.logspace(-4, 3, K)[None, :]).to(BF)      # wildly different block magnitudes
    x[M // 2, 32:96] = 0
    q, st = ops.quantize_mxfp8(x.to(cuda))
    sb = _row_major(st.cpu())
    wq, ws = _mx_quant_ref(x)
    assert torch.equal(sb, ws)
    assert torch.equal(q.cpu(), wq)
    rel = ((_mx_deq(q.cpu(), sb) - x.float()).norm() / x.float().norm()).item()
    assert rel < 0.05, rel


@pytest.mark.parametrize("loop", ["8p", "2stage"])
@pytest.mark.parametrize("M,N,K,form", [(256, 256, 128, "plain"), (300, 200, 256, "bias"), (1000, 1920, 1920, "gelu"), (777, 1920, 7680, "resid"),
                                        (70003, 1000, 384, "bias")])
def test_gemm_mxfp8_exact_operands(cuda, monkeypatch, M, N, K, form, loop):
    """Both MXFP8 main loops -- the persistent two-phase loop on v_mfma_scale_f32_16x16x128_f8f6f4 (round 6, default) and the
    round-1 two-stage loop on 32x32x64 (LD_GEMM_MX8P=0, re-read per call under LD_TUNING=1) -- against the dequantised operands
    multiplied in fp64: wildly different block scales along K, ragged M / N, one / three / odd numbers of K-tiles, more tiles than CUs."""
    from landiff_amd import ops
    monkeypatch.setenv("LD_GEMM_MX8P", "1" if loop == "8p" else "0")
    g = torch.Generator().manual_seed(M * 5 + N)
    a = (torch.randn(M, K, generator=g) * torch.logspace(-2, 1, K // 32).repeat_interleave(32)[None, :]).to(BF)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(BF)
    a8, sa = _mx_quant_ref(a)
    w8, sw = _mx_quant_ref(w)
    ref = _mx_deq(a8, sa).double() @ _mx_deq(w8, sw).double().t()
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
    out = ops.gemm_mxfp8(a8.to(cuda), _tile_major(sa).to(cuda), w8.to(cuda), _tile_major(sw).to(cuda), **kw)
    want = post(ref.float())
    err = (out.float().cpu() - want).abs().max().item() / (want.abs().max().item() + 1e-6)
    assert err < 1e-2, err


@pytest.mark.parametrize("B,N,H,K", [(2, 304, 3, 256), (1, 1000, 30, 1920), (2, 17776, 30, 1920)])
def test_gemm_qkv_heads_mxfp8_fused_split(cuda, B, N, H, K):
    """ld_gemm_qkv_heads_mxfp8 against the two launches it replaces on the same MXFP8 operands (ld_gemm_mxfp8 + ld_qkv_split): V is the
    same bf16 values moved (exact), q / k the same fp32 LayerNorm on the same bf16 inputs (one ulp where a multiply-add contracts
    differently), padding rows untouched; and against a torch restatement on the dequantised operands."""
    from landiff_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + N)
    Npad = (N + 127) // 128 * 128
    a = torch.randn(B * N, K, generator=g).to(cuda, BF)
    w = (torch.randn(3 * H * 64, K, generator=g) * 0.08).to(cuda, BF)
    bias = torch.randn(3 * H * 64, generator=g).to(cuda, BF)
    ln = tuple((torch.randn(64, generator=g) * s_ + o).to(cuda, BF) for s_, o in ((0.2, 1.0), (0.2, 0.0), (0.2, 1.0), (0.2, 0.0)))
    a8, sa = ops.quantize_mxfp8(a); w8, sw = ops.quantize_mxfp8(w)
    mk = lambda: (torch.zeros(B, H, Npad, 64, device=cuda, dtype=BF), torch.zeros(B, H, Npad, 64, device=cuda, dtype=BF),
                  torch.zeros(B, H, 64, Npad, device=cuda, dtype=BF))
    q1, k1, v1 = mk()
    ops.gemm_qkv_heads_mxfp8(a8, sa, w8, sw, bias, q1, k1, v1, B, N, H, Npad, ln, eps=1e-6)
    q2, k2, v2 = mk()
    qkv = ops.gemm_mxfp8(a8, sa, w8, sw, bias=bias)
    ops.qkv_split(qkv, q2, k2, v2, B, N, H, Npad, ln=ln, eps=1e-6)
    assert torch.equal(v1, v2)
    for x1, x2 in ((q1, q2), (k1, k2)):
        d = (x1.float() - x2.float()).abs()
        assert (d <= 2.0 ** -7 * x2.float().abs() + 1e-6).all(), d.max().item()
        assert (x1 != x2).float().mean().item() < 1e-3
    assert float(q1[:, :, N:].abs().max()) == 0.0 and float(v1[:, :, :, N:].abs().max()) == 0.0
    if N <= 1000:
        y = (_mx_deq(a8.cpu(), _row_major(sa.cpu())) @ _mx_deq(w8.cpu(), _row_major(sw.cpu())).t() + bias.float().cpu()).to(BF).float().view(B, N, 3, H, 64)
        lnf = lambda t, wv, bv: torch.nn.functional.layer_norm(t, (64,), wv.float().cpu(), bv.float().cpu(), 1e-6)
        rel = lambda x, r: ((x.float().cpu() - r).abs().max() / r.abs().max()).item()
        assert rel(q1[:, :, :N], lnf(y[:, :, 0], ln[0], ln[1]).permute(0, 2, 1, 3)) < 1.5e-2
        assert rel(k1[:, :, :N], lnf(y[:, :, 1], ln[2], ln[3]).permute(0, 2, 1, 3)) < 1.5e-2
        assert rel(v1[:, :, :, :N], y[:, :, 2].permute(0, 2, 3, 1)) < 1e-2


def test_mxfp8_chain_vs_bf16_gemm(cuda):
    from landiff_amd import ops
    g = torch.Generator().manual_seed(4)
    M, N, K = 4096, 1920, 1920
    a = torch.randn(M, K, generator=g).to(cuda, BF)
    a[:, ::97] *= 30.0                                               # outlier columns: the case block scaling is for
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(cuda, BF)
    out16 = ops.gemm(a, w).float()
    a8, sa = ops.quantize_mxfp8(a); w8, sw = ops.quantize_mxfp8(w)
    rel_mx = ((ops.gemm_mxfp8(a8, sa, w8, sw).float() - out16).norm() / out16.norm()).item()
    q8, s8 = ops.quantize_fp8(a); v8, t8 = ops.quantize_fp8(w)
    rel_row = ((ops.gemm_fp8(q8, s8, v8, t8).float() - out16).norm() / out16.norm()).item()
    assert rel_mx < 0.05, rel_mx
    assert rel_mx < 1.25 * rel_row, (rel_mx, rel_row)                # e4m3 is floating point: both sit at its ~2^-4 relative step


def test_fused_mxfp8_producers(cuda):
    """The two fused producers equal the two-pass form bit for bit: LayerNorm+modulate -> MXFP8 == ld_layernorm then
    ld_quantize_mxfp8, and the GELU epilogue with MXFP8 output == bf16 epilogue then ld_quantize_mxfp8."""
    from landiff_amd import ops
    g = torch.Generator().manual_seed(9)
    B, N, D, T = 2, 300, 1920, 26
    x = torch.randn(B * N, D, generator=g).to(cuda, BF)
    w = (1 + 0.1 * torch.randn(D, generator=g)).to(cuda, BF)
    b = (0.1 * torch.randn(D, generator=g)).to(cuda, BF)
    mod = (0.3 * torch.randn(B, 12 * D, generator=g)).to(cuda, BF)
    kw = dict(mod=mod, mod_bstride=12 * D, shift_img=0, scale_img=D, shift_txt=6 * D, scale_txt=7 * D, rows_per_batch=N, text_len=T)
    ln = torch.empty_like(x)
    ops.layernorm(x, w, b, ln, 1e-5, **kw)
    q2, s2 = ops.quantize_mxfp8(ln)
    q1 = torch.empty(B * N, D, device=cuda, dtype=torch.uint8)
    s1 = torch.empty(D // 128, B * N, 4, device=cuda, dtype=torch.uint8)
    ops.layernorm_mxfp8(x, w, b, q1, s1, 1e-5, **kw)
    assert torch.equal(q1, q2) and torch.equal(s1, s2)
    # an odd row count and a text / image boundary inside a wave's row pair (two rows per wave in the fused kernel), D = 1024
    Bo, No, Do, To = 1, 77, 1024, 13
    xo = torch.randn(Bo * No, Do, generator=g).to(cuda, BF)
    wo = (1 + 0.1 * torch.randn(Do, generator=g)).to(cuda, BF); bo = (0.1 * torch.randn(Do, generator=g)).to(cuda, BF)
    modo = (0.3 * torch.randn(Bo, 12 * Do, generator=g)).to(cuda, BF)
    kwo = dict(mod=modo, mod_bstride=12 * Do, shift_img=0, scale_img=Do, shift_txt=6 * Do, scale_txt=7 * Do, rows_per_batch=No, text_len=To)
    lno = torch.empty_like(xo)
    ops.layernorm(xo, wo, bo, lno, 1e-5, **kwo)
    qo2, so2 = ops.quantize_mxfp8(lno)
    qo1 = torch.zeros(Bo * No, Do, device=cuda, dtype=torch.uint8)
    so1 = torch.zeros(Do // 128, Bo * No, 4, device=cuda, dtype=torch.uint8)
    ops.layernorm_mxfp8(xo, wo, bo, qo1, so1, 1e-5, **kwo)
    assert torch.equal(qo1, qo2) and torch.equal(so1, so2)
    # GELU epilogue -> MXFP8
    M, Nn, K = 700, 7680, 1920
    a = torch.randn(M, K, generator=g).to(cuda, BF)
    wt = (torch.randn(Nn, K, generator=g) / K ** 0.5).to(cuda, BF)
    bias = torch.randn(Nn, generator=g).to(cuda, BF)
    a8, sa = ops.quantize_mxfp8(a); w8, sw = ops.quantize_mxfp8(wt)
    h = ops.gemm_mxfp8(a8, sa, w8, sw, bias=bias, act="gelu_tanh")
    hq2, hs2 = ops.quantize_mxfp8(h)
    hq1 = torch.empty(M, Nn, device=cuda, dtype=torch.uint8)
    hs1 = torch.empty(Nn // 128, M, 4, device=cuda, dtype=torch.uint8)
    ops.gemm_mxfp8(a8, sa, w8, sw, out=hq1, out_scales=hs1, bias=bias, act="gelu_tanh")
    assert torch.equal(hs1, hs2) and torch.equal(hq1, hq2)


def test_streaming_with_mxfp8_linears_tiny(cuda):
    """BASELINE configs[4] in miniature: the chunked long-video driver (one multi-segment AR decode overlapped with the chunk
    loop, latent prefix pinned per later chunk, VAE conv caches resident in HBM) with the DiT's four large linears on MXFP8
    operands -- fp8 GEMMs, streaming and the cache carry-over running TOGETHER, against the same run in bf16 (same tokens, same
    injected noise).  Per chunk the fp8 video stays within 10 % (Frobenius, centred) of the bf16 one; errors do not grow along
    the chain of pinned prefixes."""
    from landiff_amd.config import PipelineConfig
    from landiff_amd.pipeline import LanDiffPipeline, synthetic_inputs
    from landiff_amd.weights import init_pipeline_state
    cfg = PipelineConfig.tiny(num_steps=3).check()
    st = init_pipeline_state(cfg, seed=1234)
    n_chunks, P = 3, 1
    T = cfg.dit.latent_frames
    new = T - P
    outs = {}
    for mode in (None, "mx"):
        pipe = LanDiffPipeline(cfg, st, cuda, max_llm_frames=3 * cfg.llm.segment_length, fp8_gemm=mode)
        assert pipe.dit.fp8 == mode
        assert pipe.stream_plan(n_chunks, P) == (T, new, 3)
        inp = synthetic_inputs(cfg, cuda, n_text=6, seed=42)
        d = cfg.dit
        g = torch.Generator().manual_seed(11)
        noises = [torch.randn(1, T, d.in_channels, d.latent_h, d.latent_w, generator=g) for _ in range(n_chunks)]
        nz = [torch.randn(noises[0].shape, generator=g) for _ in range(32)]
        it = iter(nz)
        frames, video = pipe.generate_stream(inp, n_chunks, prefix_frames=P, want_float=True, noises=noises,
                                             randn_like=lambda t: next(it).to(cuda))          # overlapped AR decode (default)
        assert "llm_overlapped" in pipe.timings and not pipe.vae.cache                       # caches cleared on the last chunk only
        outs[mode] = (frames, video.cpu(), pipe.llm.out_tokens[: 3 * cfg.tok.num_latent_tokens].clone().cpu())
        del pipe
    (f16, v16, t16), (f8, v8, t8) = outs[None], outs["mx"]
    n_frames = 4 * T - 3 + (n_chunks - 1) * 4 * new
    assert f8.shape == f16.shape == (n_frames, 8 * cfg.dit.latent_h, 8 * cfg.dit.latent_w, 3) and f8.dtype == torch.uint8
    assert torch.equal(t16, t8)                                   # the AR decode is untouched by the DiT's precision
    assert not torch.equal(f16, f8)                               # ... and the fp8 path really ran
    bounds = [0, 4 * T - 3] + [4 * T - 3 + (c + 1) * 4 * new for c in range(n_chunks - 1)]
    errs = []
    for c in range(n_chunks):
        a, b = v16[:, bounds[c]:bounds[c + 1]] - 0.5, v8[:, bounds[c]:bounds[c + 1]] - 0.5
        errs.append(((a - b).norm() / a.norm()).item())
    assert max(errs) < 0.10, errs

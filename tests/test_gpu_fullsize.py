"""Full-size (BASELINE.json configs[1] shapes) checks of the HIP path: spot checks against torch fp32 on sampled rows /
patches, and size-independent properties (softmax normalisation, linearity, determinism, uint8 = trunc(float))."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_dit_attention_full_shape_spot_and_properties(cuda):
    from landiff_amd import ops
    B, H, N = 2, 30, 17776
    Npad = (N + 127) // 128 * 128
    g = torch.Generator(device=cuda).manual_seed(0)
    q = torch.zeros(B, H, Npad, 64, device=cuda, dtype=torch.bfloat16)
    k = torch.zeros_like(q)
    vt = torch.zeros(B, H, 64, Npad, device=cuda, dtype=torch.bfloat16)
    q[:, :, :N] = torch.randn(B, H, N, 64, device=cuda, generator=g).to(torch.bfloat16)
    k[:, :, :N] = torch.randn(B, H, N, 64, device=cuda, generator=g).to(torch.bfloat16)
    vt[:, :, :, :N] = torch.randn(B, H, 64, N, device=cuda, generator=g).to(torch.bfloat16)
    out = torch.empty(B, N, H * 64, device=cuda, dtype=torch.bfloat16)
    ops.attn_fwd(q, k, vt, out, N, N, 0.125)
    # spot check: 48 query rows of 3 (batch, head) pairs against fp32 torch
    rows = torch.tensor([0, 1, 127, 128, 225, 226, 4000, 8191, 12345, 17000, 17774, 17775], device=cuda)
    for (b, h) in ((0, 0), (1, 17), (1, 29)):
        s = (q[b, h, rows].float() @ k[b, h, :N].float().t()) * 0.125
        ref = torch.softmax(s, -1) @ vt[b, h, :, :N].float().t()
        got = out[b, rows, h * 64:(h + 1) * 64].float()
        assert (got - ref).abs().max().item() < 2e-2
    # property: with V == 1 every output is exactly 1 (softmax weights sum to one; padding keys carry no weight)
    vt.zero_(); vt[:, :, :, :N] = 1.0
    ops.attn_fwd(q, k, vt, out, N, N, 0.125)
    assert (out.float() - 1.0).abs().max().item() < 8e-3
    # determinism
    out2 = torch.empty_like(out)
    ops.attn_fwd(q, k, vt, out2, N, N, 0.125)
    assert torch.equal(out, out2)


def test_titok_masked_attention_full_shape(cuda):
    from landiff_amd import ops
    from landiff_amd.config import TokenizerConfig
    from landiff_amd.detokenizer import decoder_frame_ids
    cfg = TokenizerConfig()
    fid = decoder_frame_ids(cfg)
    N, H = cfg.seq_len, cfg.heads
    Npad = (N + 127) // 128 * 128
    fq = np.zeros(Npad, np.int32); fq[:N] = fid
    fk = np.full(Npad, np.iinfo(np.int32).max, np.int32); fk[:N] = fid
    kt = fk.reshape(-1, 64)
    g = torch.Generator(device=cuda).manual_seed(1)
    q = torch.zeros(1, H, Npad, 64, device=cuda, dtype=torch.bfloat16)
    k = torch.zeros_like(q)
    vt = torch.zeros(1, H, 64, Npad, device=cuda, dtype=torch.bfloat16)
    q[:, :, :N] = torch.randn(1, H, N, 64, device=cuda, generator=g).to(torch.bfloat16)
    k[:, :, :N] = torch.randn(1, H, N, 64, device=cuda, generator=g).to(torch.bfloat16)
    vt[:, :, :, :N] = torch.randn(1, H, 64, N, device=cuda, generator=g).to(torch.bfloat16)
    out = torch.empty(1, N, H * 64, device=cuda, dtype=torch.bfloat16)
    ops.attn_fwd(q, k, vt, out, N, N, 0.125, fid_q=torch.from_numpy(fq).to(cuda), fid_k=torch.from_numpy(fk).to(cuda),
                 kt_min=torch.from_numpy(kt.min(1).copy()).to(cuda), kt_max=torch.from_numpy(kt.max(1).copy()).to(cuda))
    f = torch.from_numpy(fid.astype(np.int64)).to(cuda)
    # rows in frame 0, a middle frame (straddling tiles), the last frame, I tokens and P tokens
    rows = torch.tensor([0, 1349, 1350, 8100, 17549, 17550, 17879, 17880, 18767], device=cuda)
    for h in (0, 11):
        s = (q[0, h, rows].float() @ k[0, h, :N].float().t()) * 0.125
        s = s.masked_fill(~(f[None, :] <= f[rows][:, None]), float("-inf"))
        ref = torch.softmax(s, -1) @ vt[0, h, :, :N].float().t()
        got = out[0, rows, h * 64:(h + 1) * 64].float()
        assert (got - ref).abs().max().item() < 2e-2


def test_gemm_and_conv_full_shape_patches(cuda):
    from landiff_amd import ops
    g = torch.Generator(device=cuda).manual_seed(2)
    M, N, K = 2 * 17776, 5760, 1920
    a = torch.randn(M, K, device=cuda, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=cuda, generator=g) * 0.03).to(torch.bfloat16)
    bias = torch.randn(N, device=cuda, generator=g).to(torch.bfloat16)
    out = ops.gemm(a, w, bias=bias)
    for (r0, c0) in ((0, 0), (35552 - 64, 5760 - 64), (17776 - 10, 1900), (12345, 4000)):
        ref = a[r0:r0 + 64].float() @ w[c0:c0 + 64].float().t() + bias[c0:c0 + 64].float()
        assert (out[r0:r0 + 64, c0:c0 + 64].float() - ref).abs().max().item() < 3e-2 * ref.abs().max().item()
    # linearity (size independent): gemm(a, w1 + w2) ~ gemm(a, w1) + gemm(a, w2) with exactly representable weights
    w1 = torch.randint(-1, 2, (256, 64), device=cuda, generator=g).to(torch.bfloat16)
    w2 = torch.randint(-1, 2, (256, 64), device=cuda, generator=g).to(torch.bfloat16)
    ai = torch.randint(-1, 2, (35552, 64), device=cuda, generator=g).to(torch.bfloat16)
    lhs = ops.gemm(ai, (w1 + w2), out_f32=True)
    rhs = ops.gemm(ai, w1, out_f32=True) + ops.gemm(ai, w2, out_f32=True)
    assert torch.equal(lhs, rhs)        # small integers (|sum| <= 128): exact in fp32 and representable in bf16
    # VAE level-0 causal conv (128 -> 128, 8 frames at 480x720): patch check against torch conv3d
    T, H, W, C = 8, 480, 720, 128
    xp = torch.zeros(T + 2, H + 2, W + 2, C, device=cuda, dtype=torch.bfloat16)
    xp[:, 1:-1, 1:-1] = torch.randn(T + 2, H, W, C, device=cuda, generator=g).to(torch.bfloat16)
    wt = (torch.randn(128, 3, 3, 3, C, device=cuda, generator=g) * 0.02).to(torch.bfloat16)
    y = ops.conv_cl(xp, wt, T, H, W).view(T, H, W, 128)
    t0, h0, w0 = 5, 470, 700
    patch = xp[t0:t0 + 3, h0:h0 + 12, w0:w0 + 22].permute(3, 0, 1, 2)[None].float()      # [1,C,3,12,22]
    ref = torch.nn.functional.conv3d(patch, wt.permute(0, 4, 1, 2, 3).float())[0].permute(1, 2, 3, 0)   # [1,10,20,128]
    got = y[t0, h0:h0 + 10, w0:w0 + 20].float()
    assert (got - ref[0]).abs().max().item() < 3e-2 * ref.abs().max().item()


def test_llm_full_size_decode_properties(cuda):
    from landiff_amd.config import LLMConfig
    from landiff_amd.llm import LLMRunner
    from landiff_amd.weights import init_state, llm_spec
    cfg = LLMConfig()
    run = LLMRunner(init_state(llm_spec(cfg), 3, dtype=torch.bfloat16, device=cuda), cfg, cuda)
    g = torch.Generator(device=cuda).manual_seed(4)
    text = torch.randn(48, cfg.text_dim, device=cuda, generator=g)
    t1 = run.sample(text, guidance_scale=7.5, seed=42).clone()
    t2 = run.sample(text, guidance_scale=7.5, seed=42).clone()
    assert t1.shape == (1218,) and t1.dtype == torch.int64
    assert int(t1.min()) >= 0 and int(t1.max()) <= 2047
    assert torch.equal(t1, t2)                                   # same seed -> same ids (graph replay is deterministic)
    t3 = run.sample(text, guidance_scale=7.5, seed=42, use_graph=False)
    assert torch.equal(t1, t3)                                   # eager and HIP-graph paths agree bit for bit
    t4 = run.sample(text, guidance_scale=7.5, seed=43)
    assert not torch.equal(t1, t4)


def test_dit_layer_full_shape_vs_oracle(cuda):
    """One AdaLN layer of the main DiT at the BASELINE shape (B=2 CFG pair, 17 776 tokens, hidden 1920, 30 heads) through
    the HIP path -- pipelined attention kernel, specialised GEMM epilogues, M-split launches -- against the fp32 oracle on
    the host cores (the same call bench.py's cpu_baseline times, ~10 s)."""
    import dataclasses
    from landiff_amd.config import PipelineConfig
    from landiff_amd.dit import ControlDiTRunner
    from landiff_amd.weights import dit_spec, init_state
    from oracle.dit import DiTOracle
    d1 = dataclasses.replace(PipelineConfig.full().dit, layers_main=1, layers_control=1)
    sd_main = init_state(dit_spec(d1, False), 1)
    sd_ctrl = init_state(dit_spec(d1, True), 2)
    g = torch.Generator().manual_seed(5)
    h = torch.randn(2, d1.seq_len, d1.hidden, generator=g).to(torch.bfloat16)
    emb = torch.randn(2, d1.time_embed_dim, generator=g).to(torch.bfloat16)
    torch.set_num_threads(min(64, torch.get_num_threads() * 8))
    with torch.no_grad():
        ref = DiTOracle(sd_main, d1, False, torch.float32).layer(0, h.float(), emb.float())
    run = ControlDiTRunner(sd_main, sd_ctrl, d1, cuda)
    run.emb.copy_(emb.to(cuda))
    run._modulations(run.main)
    h_dev = h.to(cuda).reshape(-1, d1.hidden).contiguous()
    out = torch.empty_like(h_dev)
    run._layer(run.main, 0, h_dev, out)
    got = out.view(2, d1.seq_len, d1.hidden).float().cpu()
    assert torch.isfinite(got).all()
    err = (got - ref).abs()
    scale = ref.abs().max().item()
    # bf16 activations through LN -> QKV -> attention -> gated residual -> MLP: a few bf16 ulps of the output range
    assert err.max().item() / scale < 3e-2, (err.max().item(), scale)
    assert err.mean().item() / ref.abs().mean().item() < 1e-2, (err.mean().item(), ref.abs().mean().item())


def test_tokenizer_encoder_full_size_causality(cuda):
    """Full-size encoder (13 x 30 x 45 visual + 1218 latent tokens, 12 layers): the mask's frame causality as a property.
    Changing the features of frames >= f must leave the I tokens (f >= 1) and the P tokens of frames < f bit-identical --
    every masked key contributes exactly zero and skipped tiles are skipped in both runs -- and must change the rest."""
    import time
    from landiff_amd.config import TokenizerConfig
    from landiff_amd.tokenizer_encoder import TokenizerEncoder
    from landiff_amd.weights import init_state, tokenizer_encoder_spec
    cfg = TokenizerConfig()
    enc = TokenizerEncoder(init_state(tokenizer_encoder_spec(cfg), 11, dtype=torch.bfloat16, device=cuda), cfg, cuda)
    g = torch.Generator(device=cuda).manual_seed(3)
    x = torch.randn(cfg.temporal, cfg.out_channels, cfg.grid_h, cfg.grid_w, device=cuda, generator=g)
    base = enc.encode(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ids = enc.encode_to_index(x)
    torch.cuda.synchronize()
    print(f"full-size encode_to_index: {(time.perf_counter() - t0) * 1e3:.1f} ms")
    assert ids.shape == (cfg.num_latent_tokens,) and int(ids.max()) < cfg.codebook_size
    nI, nP = cfg.iframe_tokens, cfg.pframe_tokens
    for f in (1, 7, 12):
        y = x.clone()
        y[f:] = torch.randn(cfg.temporal - f, cfg.out_channels, cfg.grid_h, cfg.grid_w, device=cuda, generator=g)
        out = enc.encode(y)
        keep = nI + (f - 1) * nP                    # I tokens + P tokens of frames 1..f-1
        assert torch.equal(out[:keep], base[:keep]), f
        assert not torch.equal(out[keep:keep + nP], base[keep:keep + nP]), f
    assert torch.isfinite(base.float()).all()

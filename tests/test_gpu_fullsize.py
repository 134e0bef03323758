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


def test_dit_layer_full_shape_vs_oracle(cuda, oracle_bg):
    """One AdaLN layer of the main DiT at the BASELINE shape (B=2 CFG pair, 17 776 tokens, hidden 1920, 30 heads) through
    the HIP path -- pipelined attention kernel, specialised GEMM epilogues, M-split launches -- against the fp32 oracle on
    the host cores (the same call bench.py's cpu_baseline times, ~10 s; a background job of the session: tests/oracle_jobs.py)."""
    from landiff_amd.dit import ControlDiTRunner
    from oracle_jobs import dit_layer_inputs
    d1, sd_main, sd_ctrl, h, emb = dit_layer_inputs()
    ref, _ = oracle_bg.result("dit_layer")
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


def test_dit_layer_exact_attention_flag_at_full_shape(cuda, monkeypatch):
    """ControlDiTRunner(attn_exact=True) / LD_DIT_ATTN_EXACT=1 routes the layer's attention through ld_attn_fwd_bf16_exact (two-pass
    safe softmax): on data inside the fast pass's window the layer output agrees with the default runner's to bf16 rounding."""
    from landiff_amd import _lib
    from landiff_amd.dit import ControlDiTRunner
    from oracle_jobs import dit_layer_inputs
    d1, sd_main, sd_ctrl, h, emb = dit_layer_inputs()
    outs = []
    for flag in (False, True):
        run = ControlDiTRunner(sd_main, sd_ctrl, d1, cuda, attn_exact=flag)
        run.emb.copy_(emb.to(cuda))
        run._modulations(run.main)
        h_dev = h.to(cuda).reshape(-1, d1.hidden).contiguous()
        out = torch.empty_like(h_dev)
        run._layer(run.main, 0, h_dev, out)
        torch.cuda.synchronize()
        assert (_lib.load().ld_attn_last_kernel() or b"").decode() == ("ld_attn_q64_exact_kernel" if flag else "ld_attn_q64_dyn_kernel")
        outs.append(out.float().cpu())
        del run
    assert torch.isfinite(outs[1]).all()
    d = (outs[0] - outs[1]).abs().max().item() / outs[0].abs().max().item()
    assert d < 2e-2, d
    monkeypatch.setenv("LD_DIT_ATTN_EXACT", "1")
    assert ControlDiTRunner(sd_main, sd_ctrl, d1, cuda).attn_exact is True
    monkeypatch.setenv("LD_DIT_ATTN_EXACT", "0")
    r0 = ControlDiTRunner(sd_main, sd_ctrl, d1, cuda)
    assert r0.attn_exact is False and r0.attn_auto is False
    monkeypatch.delenv("LD_DIT_ATTN_EXACT")
    ra = ControlDiTRunner(sd_main, sd_ctrl, d1, cuda)              # the default: per-layer choice from the launches' own counts
    assert ra.attn_exact is False and ra.attn_auto is True


def test_dit_attn_exact_auto_switches_only_the_layer_that_leaves_the_window(cuda):
    """ControlDiTRunner(attn_exact="auto") at the BASELINE shape, 1 control + 1 main layer: the main layer's QK-LayerNorm gains are
    scaled by 5 (row maxima of q.k/8 ~ 100: every block of its attention launch falls back), the control layer keeps unit gains.
    The per-layer counts of step s are acted on while step s + 1 is enqueued: after three steps the main layer (slot 1) runs
    ld_attn_fwd_bf16_exact, the control layer (slot 0) the default launch; the step output stays within bf16 rounding of the default
    runner's (whose launch falls back inside the kernel) at every step; steps 3 and 4 of the auto runner are run-to-run identical."""
    from landiff_amd.dit import ControlDiTRunner
    from oracle_jobs import dit_layer_inputs
    d1, sd_main, sd_ctrl, _, _ = dit_layer_inputs()
    sd_main = dict(sd_main)
    for nm in ("query", "key"):
        kname = f"mixins.adaln_layer.{nm}_layernorm_list.0.weight"
        sd_main[kname] = sd_main[kname] * 5.0
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, d1.latent_frames, d1.in_channels, d1.latent_h, d1.latent_w, generator=g).to(cuda)
    ctx = torch.randn(1, d1.text_len, d1.text_dim, generator=g).to(torch.bfloat16).float()
    sem = (0.5 * torch.randn(d1.latent_frames, d1.in_channels, d1.latent_h, d1.latent_w, generator=g)).to(torch.bfloat16)
    outs = {}
    for mode in (False, "auto"):
        run = ControlDiTRunner(sd_main, sd_ctrl, d1, cuda, attn_exact=mode)
        run.set_condition(ctx, sem)
        res = []
        for s in range(4):
            out = torch.empty_like(x)
            run.step(x, 500 - 10 * min(s, 2), -0.8, 0.6, 4.0, out)       # steps 2 and 3: the same call
            res.append(out.cpu())
            if mode == "auto":
                # step 0's counts are read at the end of step 1: from step 2 on the main layer takes the exact form
                assert run.exact_layers == (set() if s == 0 else {1}), (s, run.exact_layers)
        outs[mode] = res
        if mode == "auto":
            assert int(run._fb_host[0][1]) == run._attn_blocks and int(run._fb_host[0][0]) == 0     # step 0: all blocks / none
        del run
    for s, (a, b) in enumerate(zip(outs[False], outs["auto"])):
        assert torch.isfinite(b).all()
        if s < 2:
            assert torch.equal(a, b)                 # the same launches
        else:                                        # near one-hot softmax rows: the two safe forms round a few of them apart
            err = (a - b).abs()
            print(f"step {s}: auto vs default runner: mean |d| {err.mean().item():.3e} of mean |x| {a.abs().mean().item():.3e}, max |d| {err.max().item():.3e}")
            assert err.mean().item() <= 1e-2 * a.abs().mean().item()
    assert torch.equal(outs["auto"][2], outs["auto"][3])


def test_dit_multi_layer_step_full_shape_vs_oracle(cuda, oracle_bg):
    """A whole denoiser evaluation at the BASELINE shape with the depth cut to 3 control + 3 main layers (B=2 CFG pair, 13 x 30 x 45
    image tokens + 226 text tokens, hidden 1920): patch / text embedding with the semantic condition added on the control side,
    the control chain with its zero-linears, three resident control states added into the main chain, sat's final_layernorm,
    the final adaLN layer, unpatchify, denoiser scaling and the CFG combine -- ControlDiTRunner.step against the fp32 oracle
    on the host cores (~1 min, in a child process since the start of the session: tests/oracle_jobs.py: job_dit_3p3_eps).  What only appears at this size: the fused qkv launch, persistent 8-phase GEMMs with M-split
    tails, the 64-row attention tile, workspaces shared across layers while several control states are live."""
    from landiff_amd.dit import ControlDiTRunner
    from oracle_jobs import dit_3p3_inputs
    d3, sd_main, sd_ctrl, x, ctx, sem, timestep = dit_3p3_inputs()
    c_out, c_skip, scale = -0.8, 0.6, 4.0
    run = ControlDiTRunner(sd_main, sd_ctrl, d3, cuda)
    assert run.fuse_qkv
    run.set_condition(ctx, sem)
    outs = {}
    for s_cfg in (0.0, 1.0, scale):
        out = torch.empty(1, *x.shape[1:], device=cuda)
        run.step(x.to(cuda), timestep, c_out, c_skip, s_cfg, out)
        outs[s_cfg] = out.cpu()
    out2 = torch.empty(1, *x.shape[1:], device=cuda)
    run.step(x.to(cuda), timestep, c_out, c_skip, scale, out2)
    assert torch.equal(out2.cpu(), outs[scale])                       # run-to-run deterministic
    eps, _ = oracle_bg.result("dit_3p3_eps")
    den = eps * c_out + torch.cat([x, x]) * c_skip                   # Denoiser.forward: [uncond, cond]
    ref = {0.0: den[:1], 1.0: den[1:], scale: den[:1] + scale * (den[1:] - den[:1])}
    for s_cfg, amp in ((0.0, 1.0), (1.0, 1.0), (scale, 2 * scale - 1)):
        got, r = outs[s_cfg], ref[s_cfg]
        assert torch.isfinite(got).all()
        err = (got - r).abs()
        rng, mean = r.abs().max().item(), r.abs().mean().item()
        print(f"3+3-layer step, CFG scale {s_cfg}: max err {err.max().item() / rng:.4f} of the range, mean err {err.mean().item() / mean:.4f} of the mean")
        # seven bf16 layer-calls deep (measured on MI355X: 0.008 / 0.007 at scale 0 and 1, 0.022 / 0.020 at scale 4: the CFG combine
        # amplifies rounding noise by up to 2 s - 1); a stride / aliasing fault shows up as O(1)
        assert err.max().item() / rng < (2e-2 if amp == 1.0 else 6e-2), (s_cfg, err.max().item(), rng)
        assert err.mean().item() / mean < (1.5e-2 if amp == 1.0 else 5e-2), (s_cfg, err.mean().item(), mean)
    # the two rows are different problems (zero text vs text) whose outputs differ by LESS than the bf16 noise with random weights
    # (226 of 17 776 keys): the device's cond - uncond difference must still point the way the oracle's does, i.e. the rows are
    # not swapped or shared (a swap gives the opposite sign, a shared row zero)
    dd, dr = (outs[1.0] - outs[0.0]).flatten().double(), (ref[1.0] - ref[0.0]).flatten().double()
    cos = float(dd @ dr / (dd.norm() * dr.norm() + 1e-30))
    print(f"cond - uncond: device vs oracle cosine {cos:.3f} (|device| {dd.abs().mean():.5f}, |oracle| {dr.abs().mean():.5f})")
    assert cos > 0.15, cos


def test_dit_step_overlapped_equals_serial_at_full_shape(cuda, monkeypatch):
    """The control / main overlap at the BASELINE shape (4 control + 6 main layers): with LD_DIT_OVERLAP=1 the 64-row attention
    launches of one chain share the chip -- SIMDs included -- with the other chain's LayerNorm + modulate, GEMM epilogue and gated-
    residual kernels.  Same launches in another interleaving: two consecutive denoiser evaluations must equal the serial step
    (LD_DIT_OVERLAP=0) bit for bit, twice.  Tiny-size twin: tests/test_gpu_stages.py.  (Round 5 found packed-fp32 code of the AR decode
    that did NOT reproduce beside this attention kernel -- DESIGN.md section 5; this is the same question asked of the DiT's own kernels.)"""
    import dataclasses
    from landiff_amd.config import PipelineConfig
    from landiff_amd.dit import ControlDiTRunner
    from landiff_amd.weights import dit_spec, init_state
    d = dataclasses.replace(PipelineConfig.full().dit, layers_main=6, layers_control=4)
    sd_main, sd_ctrl = init_state(dit_spec(d, False), 1), init_state(dit_spec(d, True), 2)
    g = torch.Generator().manual_seed(13)
    x = torch.randn(1, d.latent_frames, d.in_channels, d.latent_h, d.latent_w, generator=g).to(cuda)
    ctx = torch.randn(1, d.text_len, d.text_dim, generator=g)
    sem = (0.5 * torch.randn(d.latent_frames, d.in_channels, d.latent_h, d.latent_w, generator=g)).to(torch.bfloat16)
    outs = []
    for knob in ("0", "1", "1"):
        monkeypatch.setenv("LD_DIT_OVERLAP", knob)
        run = ControlDiTRunner(sd_main, sd_ctrl, d, cuda)
        assert run.overlap == (knob == "1") and run.fuse_qkv
        run.set_condition(ctx, sem)
        o1, o2 = torch.empty_like(x), torch.empty_like(x)
        run.step(x, 700, -0.6, 0.8, 3.0, o1)
        run.step(o1, 500, -0.8, 0.6, 5.0, o2)
        torch.cuda.synchronize()
        outs.append((o1.clone(), o2.clone()))
        del run
    for o in outs[1:]:
        assert torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1])


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


def _rel(a, b):
    return ((a.float().cpu() - b.float()).abs().max() / b.float().abs().max()).item()


def test_llm_full_size_prefill_and_decode_vs_oracle(cuda):
    """The AR decoder at its real size (24 blocks x 2048, MLP 11008, vocabulary 2055, CFG pair): prefill of 64 text tokens + the
    conditioning prefix, then 3 teacher-forced decode steps -- CFG logits of the HIP path (bf16 GEMM prefill, fused GEMV decode
    steps, split-K KV attention with in-kernel RoPE + append) against the fp32 oracle on the host cores, within 2x the bf16
    oracle's own distance from it."""
    from landiff_amd.config import LLMConfig
    from landiff_amd.llm import LLMRunner
    from landiff_amd.weights import init_state, llm_spec
    from oracle.llm import LLMOracle, rope_table
    cfg = LLMConfig()
    sd_dev = init_state(llm_spec(cfg), 3, dtype=torch.bfloat16, device=cuda)
    sd = {k: v.cpu() for k, v in sd_dev.items()}                      # the same bf16 weights for the oracle (4 GB)
    run = LLMRunner(sd_dev, cfg, cuda)
    del sd_dev
    g = torch.Generator().manual_seed(4)
    text = torch.randn(64, cfg.text_dim, generator=g).to(torch.bfloat16)
    n_steps = 3
    fed = torch.randint(0, cfg.visual_vocab, (2000,), generator=g)   # teacher-fed ids (only the first n_steps matter below)
    log = []
    run.sample(text.to(cuda), guidance_scale=7.5, motion_score=0.1, seed=42, logits_log=log, teacher_fed=fed.to(cuda))
    dev = torch.cat(log[: n_steps + 1], 0).cpu()                       # CFG logits of the prefill + 3 decode steps
    torch.set_num_threads(min(64, max(torch.get_num_threads(), (torch.get_num_threads() * 8))))

    def oracle_logits(dtype):
        orc = LLMOracle(sd, cfg, dtype)
        with torch.no_grad():
            feats = orc.prefix_features(text.float(), 13.0, 0.1, True)
            S = feats.shape[1] - 1
            cos, sin = rope_table(cfg.head_dim, S + 2 + n_steps, cfg.rope_theta)
            cache = [None] * cfg.num_layers
            emb = sd["visual_embedding_model.tok_emb_code.weight"]
            out = []
            lg = orc.gpt_step(feats, cache, cos[None, : S + 1], sin[None, : S + 1]).float()
            out.append(lg[1:] + 7.5 * (lg[:1] - lg[1:]))
            for it in range(n_steps):
                f = emb[fed[it]].float().reshape(1, 1, -1)
                pos = S + 1 + it
                lg = orc.gpt_step(torch.cat([f, f], 0), cache, cos[None, pos:pos + 1], sin[None, pos:pos + 1]).float()
                out.append(lg[1:] + 7.5 * (lg[:1] - lg[1:]))
        return torch.cat(out, 0)

    ref32 = oracle_logits(torch.float32)
    ref16 = oracle_logits(torch.bfloat16)
    assert dev.shape == ref32.shape == (n_steps + 1, cfg.vocab)
    floor, err = _rel(ref16, ref32), _rel(dev, ref32)
    print(f"full-size LLM CFG logits: err {err:.4f}, bf16-oracle floor {floor:.4f}, |logit|max {ref32.abs().max():.2f}")
    assert err < max(2 * floor, 2e-2), (err, floor)
    for i in range(n_steps + 1):                                       # every step on its own, prefill and decode alike
        assert _rel(dev[i], ref32[i]) < max(2 * _rel(ref16[i], ref32[i]), 3e-2), i


def test_titok_decoder_layer_full_size_vs_oracle(cuda):
    """One full-size TiTok decoder layer in context (18 768 tokens = 13 x 30 x 45 mask tokens + 1218 latent tokens, width 768,
    12 heads, 3D RoPE, frame-block mask with tile skipping): VQ lookup -> decoder_embed -> ln_pre -> block -> ln_post -> tanh
    FFN head, HIP path vs the fp32 oracle, within 2x the bf16 oracle's own distance."""
    import dataclasses
    from landiff_amd.config import TokenizerConfig, UpsamplerConfig
    from landiff_amd.detokenizer import Detokenizer
    from landiff_amd.weights import init_state, tokenizer_spec, upsampler_spec
    from oracle.tokenizer import DetokenizerOracle
    tc = dataclasses.replace(TokenizerConfig(), layers=1)
    uc = UpsamplerConfig.tiny()                                        # (not exercised: index_to_feature stops before the upsampler)
    tok_sd = init_state(tokenizer_spec(tc), 21)
    ups_sd = init_state(upsampler_spec(uc), 22)
    tokens = torch.randint(0, tc.codebook_size, (tc.num_latent_tokens,), generator=torch.Generator().manual_seed(6))
    det = Detokenizer(tok_sd, ups_sd, tc, uc, cuda)
    got = det.index_to_feature(tokens.to(cuda)).float().cpu()         # [T, h, w, C]
    torch.set_num_threads(min(64, max(torch.get_num_threads(), torch.get_num_threads() * 8)))
    with torch.no_grad():
        ref32 = DetokenizerOracle(tok_sd, ups_sd, tc, uc, torch.float32).index_to_feature(tokens.reshape(1, 1, -1))[0]
        ref16 = DetokenizerOracle(tok_sd, ups_sd, tc, uc, torch.bfloat16).index_to_feature(tokens.reshape(1, 1, -1))[0]
    ref32, ref16 = ref32.permute(0, 2, 3, 1).float(), ref16.permute(0, 2, 3, 1).float()   # [T, C, h, w] -> [T, h, w, C]
    assert got.shape == ref32.shape == (tc.temporal, tc.grid_h, tc.grid_w, tc.out_channels)
    floor, err = _rel(ref16, ref32), _rel(got, ref32)
    print(f"full-size TiTok layer: err {err:.4f}, bf16-oracle floor {floor:.4f}")
    assert err < max(2 * floor, 1e-2), (err, floor)
    # frame causality survives at full size: frame 0's features do not depend on the P tokens (masked keys contribute exactly 0)
    t2 = tokens.clone()
    t2[tc.iframe_tokens:] = (t2[tc.iframe_tokens:] + 1) % tc.codebook_size
    got2 = det.index_to_feature(t2.to(cuda)).float().cpu()
    assert torch.equal(got2[0], got[0]) and not torch.equal(got2[1], got[1])


def test_vae_level0_resblock_full_resolution_vs_oracle(cuda, oracle_bg):
    """One level-0 resblock of the 3D-VAE decoder at 480 x 720 (128 channels, 4 frames = half a chunk): SpatialNorm3D with the
    latent-resolution conv_y / conv_b gather + swish, causal 3x3x3 conv with the replicated-first-frame halo, twice, + residual
    -- HIP path vs the fp32 oracle (the unit BASELINE.md section 3 names for the CPU baseline), 2x-floor rule."""
    from landiff_amd.vae import VAEDecoder, ZQ_PAD
    from oracle_jobs import vae_level0_inputs
    cfg, sd, p, C, T, H, W, x, zq = vae_level0_inputs()
    Tz, hz, wz = 1, 60, 90
    dec = VAEDecoder(sd, cfg, cuda)
    x_cl = x[0].permute(1, 2, 3, 0).reshape(T * H * W, C).contiguous().to(cuda)
    z_cl = torch.zeros(Tz * hz * wz, ZQ_PAD, device=cuda, dtype=torch.bfloat16)
    z_cl[:, : cfg.z_channels] = zq[0].permute(1, 2, 3, 0).reshape(-1, cfg.z_channels).to(cuda)
    # (x, None): norm1 takes its statistics from a pass over x; norm2 from the partial sums conv1's epilogue leaves (round 5)
    out, part = dec._resblock((x_cl, None), p, C, C, T, H, W, z_cl, (Tz, hz, wz), True)
    assert dec.fuse_gn_stats and part is not None
    got = out.view(T, H, W, C).permute(3, 0, 1, 2).float().cpu()
    del dec, out, x_cl, part
    (ref32, ref16), _ = oracle_bg.result("vae_level0")
    assert got.shape == ref32.shape == (C, T, H, W)
    floor, err = _rel(ref16, ref32), _rel(got, ref32)
    mean_err = (got - ref32).abs().mean().item() / ref32.abs().mean().item()
    mean_floor = (ref16 - ref32).abs().mean().item() / ref32.abs().mean().item()
    print(f"level-0 VAE resblock 480x720: err {err:.4f} (mean {mean_err:.5f}), bf16-oracle floor {floor:.4f} (mean {mean_floor:.5f})")
    assert err < max(2 * floor, 2e-2), (err, floor)
    assert mean_err < max(2 * mean_floor, 5e-3), (mean_err, mean_floor)


def test_vae_full_resolution_two_chunks_vs_oracle(cuda, oracle_bg):
    """The whole 3D-VAE decoder at 480 x 720 on the reference's chunk schedule: 5 latent frames = chunk 0:3 (odd T: replicated
    first-frame halo, the first-frame rule of both time upsamples and of the SpatialNorm zq gather, 9 frames) + chunk 3:5
    continued from the kept causal-conv caches (8 frames) -- conv_in, mid blocks, all four levels, conv_out, post-process, uint8 --
    against the fp32 oracle (~110 TFLOP of fp32 conv3d, a few minutes of the host cores: it runs in a child process from the
    start of the session, tests/oracle_jobs.py: job_vae_two_chunks, and is joined here).
    No bf16-oracle floor at this size (a bf16 conv3d on the host is several times slower still); measured on MI355X: first chunk
    max 0.058 / mean 0.0035 of the [0, 1] video range, continued chunk 0.044 / 0.0035, uint8 frames 0.9 grey levels apart on
    average -- the level of the tiny-size test against the bf16 oracle (tests/test_gpu_stages.py::test_vae_decode: 6e-2 / 4e-3).
    The bounds below leave ~1.4x of that for a different summation order of the GroupNorm statistics.
    Reference: landiff/diffusion/vae_modules/cp_enc_dec.py:416-473,605-633,1034-1069, landiff/diffusion/dif_infer.py:245-271."""
    from landiff_amd.vae import VAEDecoder
    from oracle.vae import to_uint8_frames
    from oracle_jobs import vae_two_chunks_inputs
    cfg, sd, latent = vae_two_chunks_inputs()
    vae = VAEDecoder(sd, cfg, cuda)
    frames, video = vae.decode(latent.to(cuda), want_float=True)
    assert tuple(frames.shape) == (17, 480, 720, 3) and frames.dtype == torch.uint8
    frames, video = frames.cpu(), video.cpu()
    # streaming at full resolution: decode(0:3, keep) + decode(3:5, continue) are the one-call frames, bit for bit
    fa = vae.decode(latent[:, :3].to(cuda), stream_keep=True)
    fb = vae.decode(latent[:, 3:5].to(cuda), stream_continue=True)
    assert torch.equal(torch.cat([fa, fb], 0).cpu(), frames)
    del vae, fa, fb
    torch.cuda.empty_cache()
    assert torch.equal(frames, to_uint8_frames(video))                      # uint8 = trunc(255 x float video), exactly
    ref, dt = oracle_bg.result("vae_two_chunks")                            # [3, 17, 480, 720]
    assert ref.shape == video.shape == (3, 17, 480, 720)
    err = (video - ref).abs()
    per_chunk = [(err[:, :9].max().item(), err[:, :9].mean().item()), (err[:, 9:].max().item(), err[:, 9:].mean().item())]
    grey = (frames.float() - to_uint8_frames(ref).float()).abs()
    print(f"VAE 480x720, 5 latent frames -> 17 frames vs fp32 oracle ({dt:.0f} s of host CPU, in the background): first chunk max {per_chunk[0][0]:.4f} "
          f"mean {per_chunk[0][1]:.5f}, continued chunk max {per_chunk[1][0]:.4f} mean {per_chunk[1][1]:.5f}; uint8 frames differ by "
          f"{grey.mean().item():.3f} grey levels on average, {int(grey.max().item())} at most; oracle video std {ref.std().item():.3f}")
    for mx, mean in per_chunk:
        assert mx < 8e-2 and mean < 5e-3, per_chunk
    assert grey.mean().item() < 1.5

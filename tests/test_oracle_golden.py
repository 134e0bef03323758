"""The oracle (CPU restatement) against golden vectors produced by the reference itself
(oracle/gen_golden.py, run in the authoring container).  CPU only."""
import hashlib
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


def sd_from(npz, prefix="sd/"):
    return {k[len(prefix):]: torch.from_numpy(npz[k]) for k in npz.files if k.startswith(prefix)}


def T(a):
    return torch.from_numpy(np.asarray(a))


# Floating-point tables and trajectories are bit-exact against the fixtures on the host that generated them (this container:
# the CPU tier, where anyone can regenerate tests/golden/ byte for byte with oracle/gen_golden.py).  torch's vectorised CPU
# kernels (cumprod, sqrt, exp, sin/cos) round the last bit differently on other micro-architectures -- the MI355X box's
# EPYC 9575F returns 0.82490206 where the fixture holds 0.8249021 -- and so would the reference itself there.  The GPU-tier
# re-run of this file (tests/test_gpu_oracle_pin.py) therefore sets CROSS_HOST: float comparisons relax to rtol 2e-4 (measured: 1e-5 on one timestep-embedding entry),
# integer / boolean / token-id comparisons stay exact.
CROSS_HOST = False


def feq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if not CROSS_HOST:
        return np.array_equal(a, b)
    return a.shape == b.shape and bool(np.allclose(a, b, rtol=2e-4, atol=2e-5))      # (50 sampler steps of sin / cos / exp deep)


# ---------------------------------------------------------------- schedule / sampler
def test_schedule_tables_bit_exact():
    from landiff_amd.config import SamplerConfig
    from oracle.common import timestep_embedding
    from oracle.sampler import DiffusionSamplerOracle, dynamic_cfg_scale
    g = load("schedule")
    s = DiffusionSamplerOracle(SamplerConfig())
    a, ts = s.prepare()
    assert feq(a.numpy(), g["alpha_cumprod_sqrt"])
    assert np.array_equal(ts.numpy(), g["timesteps"])
    assert feq(s.denoiser_sigmas.numpy(), g["denoiser_sigmas"])
    q, idx = s.quantize_sigma(a[:-1])
    assert feq(q.numpy(), g["quantized"]) and np.array_equal(idx.numpy(), g["quantized_idx"])
    scales = [dynamic_cfg_scale(6, 5, 50, int(50 - t)) for t in ts.tolist()[::-1][:50]]
    assert feq(np.array(scales), g["cfg_scales"])
    assert abs(scales[0] - 2.2915) < 1e-3 and abs(scales[2] - 6.9744) < 1e-3   # SURVEY 3.3 quirk values
    te = timestep_embedding(torch.tensor([999.0, 19.0, 500.0]), 64)
    assert feq(te.numpy(), g["timestep_embedding"])


@pytest.mark.parametrize("name,n,kind", [("vpsde", 50, "vpsde_dpmpp2m"), ("vpsde7", 7, "vpsde_dpmpp2m"), ("ddim", 10, "ddim"),
                                         ("vpsde_fixed2", 6, "vpsde_dpmpp2m")])
def test_sampler_trajectory_bit_exact(name, n, kind):
    from landiff_amd.config import SamplerConfig
    from oracle.sampler import DiffusionSamplerOracle
    g = load("sampler")

    def network(x, t, ctx):
        c = ctx.mean(dim=(1, 2)).view(-1, 1, 1, 1, 1)
        return 0.6 * x * torch.cos(t * 0.01).view(-1, 1, 1, 1, 1) + 0.1 * c + 0.05 * torch.sin(x * 3.0)

    s = DiffusionSamplerOracle(SamplerConfig(num_steps=n, sampler=kind))
    torch.manual_seed(1234)
    x = torch.randn(1, 3, 4, 4, 6)
    cond = torch.randn(1, 5, 8)
    assert np.array_equal(x.numpy(), g[name + "_x0"]) and np.array_equal(cond.numpy(), g[name + "_cond"])
    out = s.run(network, x.clone(), cond, torch.zeros_like(cond), fixed_frames=2 if name == "vpsde_fixed2" else 0)
    assert feq(out.numpy(), g[name + "_out"])
    if name == "vpsde_fixed2":
        assert np.array_equal(out[:, :2].numpy(), g[name + "_x0"][:, :2])      # pinned frames come back untouched


# ---------------------------------------------------------------- RoPE / mask
def test_rope_tables_and_apply():
    from landiff_amd.config import TokenizerConfig
    from oracle.llm import apply_rope, rope_table
    from oracle.tokenizer import rope3d_table
    g = load("rope")
    cos, sin = rope_table(128, 40)
    assert feq(cos.numpy(), g["f1_real"]) and feq(sin.numpy(), g["f1_imag"])
    q, k = T(g["q"]), T(g["k"])
    np.testing.assert_allclose(apply_rope(q, cos[None], sin[None]).numpy(), g["q_out"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(apply_rope(k, cos[None], sin[None]).numpy(), g["k_out"], rtol=0, atol=1e-6)
    Tn, H, W, nI, nP = g["grid"].tolist()
    cfg = TokenizerConfig(width=128, heads=2, grid_h=H, grid_w=W, temporal=Tn, pframe_tokens=nP, num_latent_tokens=nI + (Tn - 1) * nP)
    c3, s3 = rope3d_table(cfg)
    assert feq(c3.numpy(), g["f3_real"]) and feq(s3.numpy(), g["f3_imag"])


def test_decoder_mask_closed_form():
    from landiff_amd.config import TokenizerConfig
    from oracle.tokenizer import decoder_mask_scalar, frame_ids
    g = load("decoder_mask")
    Tn, tpf, nI, nP = g["tiny_cfg"].tolist()
    cfg = TokenizerConfig(grid_h=1, grid_w=tpf, temporal=Tn, pframe_tokens=nP, num_latent_tokens=nI + (Tn - 1) * nP)
    fid = frame_ids(cfg)
    dense = fid[None, :] <= fid[:, None]
    assert np.array_equal(dense, g["tiny_dense"])
    n = cfg.seq_len
    scal = np.array([[decoder_mask_scalar(cfg, q, k) for k in range(n)] for q in range(n)])
    assert np.array_equal(scal, g["tiny_dense"])
    # full size: sha256 of the padded boolean mask and per-128-block occupancy
    Tn, tpf, nI, nP = g["full_cfg"].tolist()
    cfg = TokenizerConfig()
    assert (cfg.temporal, cfg.tokens_per_frame, cfg.iframe_tokens, cfg.pframe_tokens) == (Tn, tpf, nI, nP)
    fid = frame_ids(cfg)
    L = len(fid)
    Lp = (L + 127) // 128 * 128
    fq = np.full(Lp, -1, np.int64); fq[:L] = fid           # padded queries see nothing
    fk = np.full(Lp, 1 << 30, np.int64); fk[:L] = fid      # padded keys are never seen
    h = hashlib.sha256()
    occ = np.zeros((Lp // 128, Lp // 128), np.int32)
    for r0 in range(0, Lp, 128):
        blk = (fk[None, :] <= fq[r0:r0 + 128, None]).astype(np.uint8)
        h.update(np.packbits(blk, axis=None).tobytes())
        occ[r0 // 128] = blk.reshape(128, Lp // 128, 128).sum(axis=(0, 2))
    assert np.array_equal(np.frombuffer(h.digest(), dtype=np.uint8), g["full_sha256"])
    assert np.array_equal(occ, g["full_block_occupancy"])
    dens = occ.sum() / (L * L)
    assert abs(dens - 0.5385) < 2e-3


# ---------------------------------------------------------------- LLM decode
def _llm(tag, dtype):
    from landiff_amd.config import LLMConfig
    from landiff_amd.weights import init_state, llm_spec
    from oracle.llm import LLMOracle
    g = load(f"llm_{tag}")
    cfg = LLMConfig.tiny()
    return g, cfg, LLMOracle(init_state(llm_spec(cfg), int(g["seed"])), cfg, dtype)


def test_llm_fp32_logits_and_tokens():
    g, cfg, orc = _llm("fp32", torch.float32)
    torch.manual_seed(42)
    codes, logits = orc.sample(T(g["text"]), motion_score=0.1, num_frames=cfg.segment_length, guidance_scale=7.5,
                               return_logits=True)
    ref = T(g["logits"]).view(-1, 2, cfg.vocab)
    ref_cfg = ref[:, 1] + 7.5 * (ref[:, 0] - ref[:, 1])
    assert np.array_equal(codes.numpy(), g["codes"])            # bit-exact ids (same CPU RNG stream)
    np.testing.assert_allclose(logits.numpy(), ref_cfg.numpy(), rtol=0, atol=2e-4)
    torch.manual_seed(43)
    codes2 = orc.sample(T(g["text"]), motion_score=0.1, num_frames=2 * cfg.segment_length, guidance_scale=7.5)
    assert np.array_equal(codes2.numpy(), g["codes_2seg"])


def test_llm_gt_first_frame_fp32():
    """use_gt_first_frame (lm_model.py:332-352) of the reference's Semantic1DLM.sample: the I frame of a given token stream
    is prefilled, sampling starts at the first P token; ids bit-exact (same CPU RNG stream), CFG logits of every step."""
    g, cfg, orc = _llm("fp32", torch.float32)
    gt = load("llm_fp32_gt_first_frame")
    first = T(gt["gt"])[: cfg.iframe_len]
    torch.manual_seed(44)
    codes, logits = orc.sample(T(gt["text"]), motion_score=0.1, num_frames=cfg.segment_length, guidance_scale=7.5,
                               return_logits=True, first_frame_tokens=first)
    assert np.array_equal(codes.numpy(), gt["codes"])
    assert np.array_equal(codes.numpy()[0, : cfg.iframe_len], gt["gt"][: cfg.iframe_len])
    ref = T(gt["logits"]).view(-1, 2, cfg.vocab)
    ref_cfg = ref[:, 1] + 7.5 * (ref[:, 0] - ref[:, 1])
    np.testing.assert_allclose(logits.numpy(), ref_cfg.numpy(), rtol=0, atol=2e-4)


def test_llm_top_k_top_p_fp32():
    """top-k and top-p sampling (lm_model.py:441-447) against the reference's own runs: ids bit-exact on the CPU RNG stream."""
    g, cfg, orc = _llm("fp32", torch.float32)
    gk = load("llm_fp32_topk_topp")
    torch.manual_seed(45)
    ck = orc.sample(T(gk["text"]), motion_score=0.1, num_frames=cfg.segment_length, guidance_scale=7.5, top_k=5)
    assert np.array_equal(ck.numpy(), gk["codes_top_k5"])
    torch.manual_seed(46)
    cp = orc.sample(T(gk["text"]), motion_score=0.1, num_frames=cfg.segment_length, guidance_scale=7.5, temperature=0.7, top_p=0.8)
    assert np.array_equal(cp.numpy(), gk["codes_top_p08_t07"])


def test_llm_bf16_dtype_flow():
    """bf16 mode follows the reference's autocast flow (reference run under CPU autocast)."""
    g, cfg, orc = _llm("bf16", torch.bfloat16)
    torch.manual_seed(42)
    ref = T(g["logits"]).view(-1, 2, cfg.vocab)
    ref_cfg = ref[:, 1] + 7.5 * (ref[:, 0] - ref[:, 1])
    ref_codes = T(g["codes"])
    codes, logits = orc.sample(T(g["text"]), motion_score=0.1, num_frames=cfg.segment_length, guidance_scale=7.5,
                               return_logits=True, teacher_tokens=torch.cat([T(g["fed_tokens"]), torch.zeros(1, dtype=torch.long)]))
    # teacher-forced: every step sees the reference's history, so logits are comparable step by step
    err = (logits - ref_cfg).abs().max().item()
    assert err < 0.25, err
    assert (codes == ref_codes).float().mean().item() > 0.9


def test_llm_forced_schedule_full_size():
    from landiff_amd.config import LLMConfig
    from oracle.llm import forced_schedule
    cfg = LLMConfig()
    S = 70
    full_len, forced, restricted, n_visual = forced_schedule(cfg, S, 13)
    assert full_len == S + 1244 + 1 and n_visual == 1218 and full_len - (S + 1) == 1244
    assert forced[S + 331] == cfg.END_I and forced[S + 332] == cfg.START_P and forced[S + 332 + 75] == cfg.END_P
    assert forced[full_len - 1] == cfg.EOS and len(forced) == 26


# ---------------------------------------------------------------- detokenizer
def test_titok_decoder_fp32():
    from landiff_amd.config import TokenizerConfig, UpsamplerConfig
    from landiff_amd.weights import init_state, tokenizer_spec
    from oracle.tokenizer import DetokenizerOracle
    g = load("titok_fp32")
    cfg = TokenizerConfig.tiny()
    orc = DetokenizerOracle(init_state(tokenizer_spec(cfg), int(g["seed"])), {}, cfg, UpsamplerConfig.tiny(), torch.float32)
    z = T(g["z"])                                    # [1, C, 1, L] as the reference decoder takes it
    orc.index_to_latent = lambda tokens: z[0, :, 0].t()
    out = orc.index_to_feature(torch.zeros(1, dtype=torch.long))
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=0, atol=2e-5)


def test_titok_encoder_fp32():
    """TokenizerEncoderOracle.encode against TiTokEncoder.forward of the reference (tiny config, same seeded weights)."""
    from landiff_amd.config import TokenizerConfig
    from landiff_amd.weights import init_state, tokenizer_encoder_spec
    from oracle.tokenizer import TokenizerEncoderOracle
    g = load("titok_enc_fp32")
    cfg = TokenizerConfig.tiny()
    orc = TokenizerEncoderOracle(init_state(tokenizer_encoder_spec(cfg), int(g["seed"])), cfg, torch.float32)
    out = orc.encode(T(g["x"]))                                        # [1, L, token_size]
    ref = T(g["out"])[0, :, 0].t()                                     # reference returns [1, token_size, 1, L]
    np.testing.assert_allclose(out[0].numpy(), ref.numpy(), rtol=0, atol=2e-5)


def test_feature_norm_gate():
    """norm_features / denorm_features act only when the config names a mean_std_path (video_titok_vq.py:221-233): with the
    shipped config the checkpoint's mean/std buffers must NOT be applied.  Golden = the reference's own two methods run with
    mean_std_path None / set on non-trivial statistics."""
    import dataclasses
    from landiff_amd.config import TokenizerConfig
    from oracle.tokenizer import DetokenizerOracle, TokenizerEncoderOracle
    g = load("feature_norm")
    x, st = T(g["x"]), {"mean": T(g["mean"]), "std": T(g["std"])}
    off = dataclasses.replace(TokenizerConfig.tiny(), out_channels=x.shape[2])
    on = dataclasses.replace(off, norm_features=True)
    assert not TokenizerConfig().norm_features                         # the shipped configuration
    np.testing.assert_array_equal(TokenizerEncoderOracle(st, off, torch.float32).norm_features(x).numpy(), g["norm_off"])
    np.testing.assert_array_equal(g["norm_off"], g["x"])
    np.testing.assert_array_equal(TokenizerEncoderOracle(st, on, torch.float32).norm_features(x).numpy(), g["norm_on"])
    xcl = x.permute(0, 1, 3, 4, 2)
    cl = lambda a: np.transpose(a, (0, 1, 3, 4, 2))
    np.testing.assert_array_equal(DetokenizerOracle(st, None, off, None, torch.float32).denorm_features(xcl).numpy(), cl(g["denorm_off"]))
    np.testing.assert_array_equal(DetokenizerOracle(st, None, on, None, torch.float32).denorm_features(xcl).numpy(), cl(g["denorm_on"]))
    out = DetokenizerOracle(st, None, on, None, torch.bfloat16).denorm_features(xcl.to(torch.bfloat16))
    np.testing.assert_array_equal(out.float().numpy(), cl(g["denorm_on_bf16"]))


def test_encoder_mask_closed_form_and_launch_labels():
    """VideoEncoderMask: scalar and vectorised restatements equal the reference's mask (tiny dense; full size by sha256
    and row counts), and the two-launch label scheme of the product (landiff_amd/tokenizer_encoder.py) reproduces it."""
    from landiff_amd.config import TokenizerConfig
    from landiff_amd.tokenizer_encoder import encoder_attention_labels
    from oracle.tokenizer import encoder_mask_dense, encoder_mask_scalar
    g = load("encoder_mask")

    def from_labels(cfg, r0, r1):
        qv, kv, ql, kl = encoder_attention_labels(cfg)
        nv = cfg.n_visual
        rows = np.arange(r0, min(r1, cfg.seq_len))
        vis = kv[None, :].astype(np.int64) <= qv[rows][:, None]
        lat = kl[None, :].astype(np.int64) <= ql[rows][:, None]
        return np.where((rows < nv)[:, None], vis, lat)

    Tn, tpf, nI, nP = g["tiny_cfg"].tolist()
    cfg = TokenizerConfig(grid_h=1, grid_w=tpf, temporal=Tn, pframe_tokens=nP, num_latent_tokens=nI + (Tn - 1) * nP)
    n = cfg.seq_len
    scal = np.array([[encoder_mask_scalar(cfg, q, k) for k in range(n)] for q in range(n)])
    assert np.array_equal(scal, g["tiny_dense"])
    assert np.array_equal(encoder_mask_dense(cfg).numpy(), g["tiny_dense"])
    assert np.array_equal(from_labels(cfg, 0, n), g["tiny_dense"])
    Tn, tpf, nI, nP = g["full_cfg"].tolist()
    cfg = TokenizerConfig()
    assert (cfg.temporal, cfg.tokens_per_frame, cfg.iframe_tokens, cfg.pframe_tokens) == (Tn, tpf, nI, nP)
    L = cfg.seq_len
    h1, h2 = hashlib.sha256(), hashlib.sha256()
    rows = np.zeros(L, np.int64)
    for r0 in range(0, L, 128):
        blk = encoder_mask_dense(cfg, r0, r0 + 128).numpy()
        h1.update(np.packbits(blk.astype(np.uint8), axis=None).tobytes())
        h2.update(np.packbits(from_labels(cfg, r0, r0 + 128).astype(np.uint8), axis=None).tobytes())
        rows[r0:r0 + blk.shape[0]] = blk.sum(1)
    assert np.array_equal(np.frombuffer(h1.digest(), dtype=np.uint8), g["full_sha256"])
    assert np.array_equal(np.frombuffer(h2.digest(), dtype=np.uint8), g["full_sha256"])
    assert np.array_equal(rows, g["full_row_counts"])


def test_vq_nearest_code_round_trip():
    """Nearest-code restatement (vector-quantize-pytorch, parity unpinned): the code vectors themselves map to their own
    indices, and project_out(codebook[idx]) fed back through project_in's pseudo-inverse is not needed for that property."""
    from landiff_amd.config import TokenizerConfig
    from landiff_amd.weights import init_state, tokenizer_encoder_spec
    from oracle.tokenizer import TokenizerEncoderOracle
    cfg = TokenizerConfig.tiny()
    sd = init_state(tokenizer_encoder_spec(cfg), 3)
    # make project_in an exact left inverse on a 16-dim subspace: z = e @ pinv(W)^T  =>  W z + b = e when b = 0
    W = sd["quantizer.project_in.weight"].double()
    sd["quantizer.project_in.bias"].zero_()
    e = sd["quantizer._codebook.embed"][0].double()
    z = (e @ torch.linalg.pinv(W).t()).float()
    orc = TokenizerEncoderOracle(sd, cfg, torch.float32)
    assert torch.equal(orc.nearest_code(z), torch.arange(cfg.codebook_size))


def test_upsampler_and_semantic_conv_fp32():
    from landiff_amd.config import TokenizerConfig, UpsamplerConfig
    from landiff_amd.weights import init_state, upsampler_spec
    from oracle.tokenizer import DetokenizerOracle
    g = load("upsampler_fp32")
    cfg = UpsamplerConfig.tiny()
    orc = DetokenizerOracle({}, init_state(upsampler_spec(cfg), int(g["seed"])), TokenizerConfig.tiny(), cfg, torch.float32)
    up = orc.upsample(T(g["x"]))
    np.testing.assert_allclose(up.numpy(), g["up"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(orc._conv(up, "conv_out").numpy(), g["out"], rtol=0, atol=5e-5)


# ---------------------------------------------------------------- VAE
def test_vae_chunked_decode_fp32():
    from landiff_amd.config import VAEConfig
    from landiff_amd.weights import init_state, vae_spec
    from oracle.vae import VAEDecoderOracle
    g = load("vae_fp32")
    cfg = VAEConfig.tiny()
    orc = VAEDecoderOracle(init_state(vae_spec(cfg), int(g["seed"])), cfg, torch.float32)
    out = orc.decode_latent(T(g["latent"]))
    assert out.shape == (1, 3, 25, 32, 48)
    np.testing.assert_allclose(out.numpy(), g["chunked"], rtol=0, atol=2e-4)
    # the witness that the chunk schedule matters: a single full-length pass differs (per-chunk GroupNorm)
    assert np.abs(g["chunked"][:, :, :9] - g["full_first9"]).max() > 1e-2


# ---------------------------------------------------------------- DiT
def test_vae_stream_decode_equals_reference_schedule():
    """Streaming use of the conv caches: decoding latent frames [0:5] in one call (the reference's chunk schedule 0:3, 3:5 with
    clear_cache only on the last, dif_infer.py:245-271) must equal decode([0:3], keep caches) + decode([3:5], continue)."""
    from landiff_amd.config import VAEConfig
    from landiff_amd.weights import init_state, vae_spec
    from oracle.vae import VAEDecoderOracle
    cfg = VAEConfig.tiny()
    orc = VAEDecoderOracle(init_state(vae_spec(cfg), 5), cfg, torch.float32)
    g = torch.Generator().manual_seed(3)
    lat = torch.randn(1, cfg.z_channels, 7, 6, 8, generator=g)
    whole = orc.decode_latent(lat)
    a = orc.decode_latent(lat[:, :, :3], stream_keep=True)
    b = orc.decode_latent(lat[:, :, 3:5], stream_continue=True, stream_keep=True)
    c = orc.decode_latent(lat[:, :, 5:7], stream_continue=True)
    assert not orc.cache                                   # the last call cleared the caches
    assert torch.equal(torch.cat([a, b, c], dim=2), whole)


def test_dit_layers_fp32():
    from landiff_amd.config import DiTConfig
    from landiff_amd.weights import dit_spec, init_state
    from oracle.dit import DiTOracle, sincos_pos_embed_3d
    g = load("dit_fp32")
    cfg = DiTConfig.tiny()
    main = DiTOracle(init_state(dit_spec(cfg, False), 9), cfg, False, torch.float32)
    ctrl = DiTOracle(init_state(dit_spec(cfg, True), 10), cfg, True, torch.float32)
    h, emb = T(g["control_h"]), T(g["control_emb"])
    for i in range(cfg.layers_control):
        h = ctrl.layer(i, h, emb)
        np.testing.assert_allclose(h.numpy(), g["control_out"][i], rtol=0, atol=5e-5)
    h, emb = T(g["main_h"]), T(g["main_emb"])
    cl = T(g["main_ctrl"])
    for i in range(cfg.layers_main):
        h = main.layer(i, h, emb, cl[i] if i < cfg.layers_control else None)
        np.testing.assert_allclose(h.numpy(), g["main_out"][i], rtol=0, atol=5e-5)
    # patch embed + text proj (no pos table), final layer incl. unpatchify, sin-cos table
    s = main.s
    pos = s["mixins.pos_embed.pos_embedding"]
    s["mixins.pos_embed.pos_embedding"] = torch.zeros_like(pos)
    e = main.embed(T(g["embed_img"]), T(g["embed_ctx"]))
    np.testing.assert_allclose(e.numpy(), g["embed_out"], rtol=0, atol=5e-5)
    pe = sincos_pos_embed_3d(cfg.hidden, cfg.grid_h, cfg.grid_w, cfg.latent_frames, 1.875, 1.875)
    np.testing.assert_allclose(pe.astype(np.float32), g["pos_embed"], rtol=0, atol=1e-6)
    # final_forward only (the sat final_layernorm in front of it is the build's restatement)
    import torch.nn.functional as F
    from oracle.common import layer_norm, linear
    from oracle.dit import modulate
    x = T(g["final_in"])[:, cfg.text_len:]
    mod = linear(F.silu(emb), s["mixins.final_layer.adaLN_modulation.1.weight"], s["mixins.final_layer.adaLN_modulation.1.bias"])
    shift, scale = mod.chunk(2, dim=1)
    x = modulate(layer_norm(x, s["mixins.final_layer.norm_final.weight"], s["mixins.final_layer.norm_final.bias"], 1e-6), shift, scale)
    x = linear(x, s["mixins.final_layer.linear.weight"], s["mixins.final_layer.linear.bias"])
    B, Tn, p = x.shape[0], cfg.latent_frames, cfg.patch
    x = x.view(B, Tn, cfg.grid_h, cfg.grid_w, cfg.out_channels, p, p).permute(0, 1, 4, 2, 5, 3, 6).reshape(B, Tn, cfg.out_channels, cfg.latent_h, cfg.latent_w)
    np.testing.assert_allclose(x.numpy(), g["final_out"], rtol=0, atol=5e-5)

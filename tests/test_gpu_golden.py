"""The HIP path against the REFERENCE's own numbers, directly: tests/golden/*.npz hold inputs and outputs of the reference's
modules (oracle/gen_golden.py, run in the authoring container with the seeded synthetic state dicts of landiff_amd/weights.py
loaded strictly into them).  Here the same state dicts are rebuilt from the stored seeds, packed by the product runners
(LLMRunner / Detokenizer / VAEDecoder / ControlDiTRunner), fed the stored inputs on the GPU through the C ABI and compared with
the stored outputs -- no oracle between the device and the reference (the oracle only supplies the yardstick: the bf16
restatement's own distance from the same golden numbers, DESIGN.md section 5's 2x-floor rule).

The reference ran in fp32 (and the LLM once more under bf16 autocast); the device computes in bf16, as the reference does on
its production path.
"""
import dataclasses
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


def T(a):
    return torch.from_numpy(np.asarray(a))


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def check(name, got, ref, bf16_restatement, abs_floor):
    """2x-floor rule against the reference's numbers: the device may be at most twice as far from the golden output as the
    bf16 CPU restatement of the same computation is (both relative to the output range)."""
    err, floor = rel(got, ref), rel(bf16_restatement, ref)
    print(f"{name}: device vs reference {err:.5f}, bf16 restatement vs reference {floor:.5f}")
    assert torch.isfinite(got.float()).all()
    assert err < max(2 * floor, abs_floor), (name, err, floor)


# ------------------------------------------------------------------------------------------ LLM
def _llm_logits(cuda, tag):
    from landiff_amd.config import LLMConfig
    from landiff_amd.llm import LLMRunner
    from landiff_amd.weights import init_state, llm_spec
    from oracle.llm import LLMOracle
    g = load(f"llm_{tag}")
    cfg = LLMConfig.tiny()
    sd = init_state(llm_spec(cfg), int(g["seed"]))
    text, fed = T(g["text"]), T(g["fed_tokens"]).long()
    ref = T(g["logits"]).view(-1, 2, cfg.vocab)                      # reference rows per step: (cond, uncond)
    ref_cfg = ref[:, 1] + 7.5 * (ref[:, 0] - ref[:, 1])
    run = LLMRunner(sd, cfg, cuda, max_text=32, max_frames=cfg.segment_length)
    log = []
    # teacher-forced on the tokens the REFERENCE fed back (forced ones included), so every step sees the reference's history
    run.sample(text, motion_score=0.1, num_frames=cfg.segment_length, guidance_scale=7.5, seed=42, logits_log=log,
               teacher_fed=torch.cat([fed, torch.zeros(1, dtype=torch.long)]).to(cuda))
    dev = torch.cat(log, 0).cpu()
    assert dev.shape == ref_cfg.shape, (dev.shape, ref_cfg.shape)
    teacher = torch.cat([fed, torch.zeros(1, dtype=torch.long)])
    orc = lambda dt: LLMOracle(sd, cfg, dt).sample(text, motion_score=0.1, num_frames=cfg.segment_length, guidance_scale=7.5,
                                                    return_logits=True, teacher_tokens=teacher,
                                                    multinomial_fn=lambda p: torch.multinomial(p, 1))[1]
    return g, cfg, dev, ref_cfg, orc


def test_llm_logits_vs_reference_fp32_run(cuda):
    """Semantic1DLM.sample of the reference in fp32 (llm_fp32.npz: CFG logits of every step of a 2-frame decode, prefill
    included): LLMRunner on the same weights / text / fed-back tokens."""
    g, cfg, dev, ref_cfg, orc = _llm_logits(cuda, "fp32")
    o16 = orc(torch.bfloat16)
    check("LLM CFG logits (reference fp32 run)", dev, ref_cfg, o16, 2e-2)
    for s in range(dev.shape[0]):          # and every step on its own, so one bad position cannot hide behind the global maximum
        assert rel(dev[s], ref_cfg[s]) < max(2 * rel(o16[s], ref_cfg[s]), 3e-2), s


def test_llm_logits_vs_reference_bf16_autocast_run(cuda):
    """The reference's own bf16 flow (llm_bf16.npz: the same model under CPU autocast, its own sampled history).  Yardstick:
    the reference's bf16 logits' distance from fp32 on that history -- the device, teacher-forced on the same tokens, must be as
    close to fp32 as 2x that, and within the sum of both distances of the reference's bf16 numbers."""
    g, cfg, dev, ref_bf16, orc = _llm_logits(cuda, "bf16")
    ref32 = orc(torch.float32)                                        # fp32 on the bf16 run's history (the oracle is pinned to the fp32 golden)
    scale = ref32.abs().max().item()
    floor = (ref_bf16 - ref32).abs().max().item() / scale
    err = (dev - ref32).abs().max().item() / scale
    direct = (dev - ref_bf16).abs().max().item() / scale
    print(f"LLM vs reference bf16-autocast run: device-fp32 {err:.4f}, reference(bf16)-fp32 {floor:.4f}, device-reference(bf16) {direct:.4f}")
    assert err < max(2 * floor, 2e-2), (err, floor)
    assert direct < max(3 * floor, 3e-2), (direct, floor)


def test_llm_reference_ids_reproduced_from_reference_logits(cuda):
    """The reference's sampled ids from the device's sampling kernel: fed the reference's own per-step (cond, uncond) logits
    and the CPU generator's Exp(1) draws of torch.manual_seed(42), ld_llm_logits_to_probs + argmax(p / q) must return the
    reference's codes bit for bit (CFG combine, temperature, position restrictions, forced schedule, clamp)."""
    from landiff_amd import ops
    from landiff_amd.config import LLMConfig
    from landiff_amd.llm import forced_token_schedule
    g = load("llm_fp32")
    cfg = LLMConfig.tiny()
    ref = T(g["logits"]).view(-1, 2, cfg.vocab)
    S = g["text"].shape[0] + 3
    full_len, forced, restricted, n_vis = forced_token_schedule(cfg, S, cfg.segment_length)
    assert ref.shape[0] == full_len - (S + 1)
    al = torch.zeros(full_len + 2, 4, dtype=torch.int32)
    for p, ids in restricted.items():
        al[p, 0] = len(ids)
        al[p, 1:1 + len(ids)] = torch.tensor(ids, dtype=torch.int32)
    al = al.to(cuda)
    probs = torch.empty(1, cfg.vocab, device=cuda); cfgl = torch.empty(1, cfg.vocab, device=cuda)
    pos = torch.zeros(1, device=cuda, dtype=torch.int32)
    torch.manual_seed(42)                                            # the reference's torch.multinomial stream (CPU generator)
    out = []
    for s, i in enumerate(range(S + 1, full_len)):
        pos.fill_(i - 1)            # the kernel looks the restriction of the position being sampled up at *pos + 1
        ops.llm_logits_to_probs(ref[s].to(cuda).contiguous(), probs, cfgl, True, 7.5, 1.0, pos, al)
        p = probs.cpu()
        q = torch.empty_like(p).exponential_(1.0)                    # ATen multinomial: argmax(p / q)
        tok = int(torch.argmax(p / q, dim=-1))
        if i not in forced:
            out.append(tok)
    codes = torch.tensor(out).clamp(0, cfg.visual_vocab - 1)
    assert np.array_equal(codes.numpy(), g["codes"].reshape(-1)), (codes.numpy() != g["codes"].reshape(-1)).sum()


# ------------------------------------------------------------------------------------------ detokenizer
def test_titok_decoder_vs_reference(cuda):
    """TiTokDecoder.forward of the reference (titok_fp32.npz) on the latent tokens it was given: Detokenizer.latent_to_feature."""
    from landiff_amd.config import TokenizerConfig, UpsamplerConfig
    from landiff_amd.detokenizer import Detokenizer
    from landiff_amd.weights import init_state, tokenizer_spec, upsampler_spec
    from oracle.tokenizer import DetokenizerOracle
    g = load("titok_fp32")
    tc, uc = TokenizerConfig.tiny(), UpsamplerConfig.tiny()
    sd = init_state(tokenizer_spec(tc), int(g["seed"]))
    z = T(g["z"])[0, :, 0].t().contiguous()                           # [L, token_size]; the reference takes [1, C, 1, L]
    det = Detokenizer(sd, init_state(upsampler_spec(uc), 7), tc, uc, cuda)
    got = det.latent_to_feature(z.to(cuda)).permute(0, 3, 1, 2)[None]  # [T,h,w,C] -> [1,T,C,h,w]
    ref = T(g["out"])
    assert tuple(got.shape) == tuple(ref.shape)
    orc = DetokenizerOracle(sd, {}, tc, uc, torch.bfloat16)
    orc.index_to_latent = lambda tokens: z.to(torch.bfloat16)
    check("TiTok decoder", got, ref, orc.index_to_feature(torch.zeros(1, dtype=torch.long)), 1e-2)


def test_upsampler_vs_reference(cuda):
    """vq_gan_blocks.Decoder (PixelShuffle upsampler) + SemanticCond.conv_out of the reference (upsampler_fp32.npz)."""
    from landiff_amd.config import TokenizerConfig, UpsamplerConfig
    from landiff_amd.detokenizer import Detokenizer
    from landiff_amd.weights import init_state, tokenizer_spec, upsampler_spec
    from oracle.tokenizer import DetokenizerOracle
    g = load("upsampler_fp32")
    tc, uc = TokenizerConfig.tiny(), UpsamplerConfig.tiny()
    sd = init_state(upsampler_spec(uc), int(g["seed"]))
    det = Detokenizer(init_state(tokenizer_spec(tc), 6), sd, tc, uc, cuda)
    x = T(g["x"])                                                     # [F, z_channels, h, w]
    x_cl = x.permute(0, 2, 3, 1).contiguous().to(cuda, torch.bfloat16)
    up = det.upsample(x_cl)                                           # [F, 2^k h, 2^k w, out_ch]
    out = det._condition_from_cl(x_cl)                                # [F, target_dim, H, W]
    orc = DetokenizerOracle({}, sd, tc, uc, torch.bfloat16)
    up16 = orc.upsample(x.to(torch.bfloat16))
    check("upsampler", up.permute(0, 3, 1, 2), T(g["up"]), up16, 1e-2)
    check("upsampler + conv_out", out, T(g["out"]), orc._conv(up16, "conv_out"), 1e-2)


# ------------------------------------------------------------------------------------------ VAE
def test_vae_chunked_decode_vs_reference(cuda):
    """ContextParallelDecoder3D under CogWrapper.decode_latent's chunk schedule (vae_fp32.npz: 7 latent frames -> 25 frames,
    chunks 3 + 2 + 2, conv caches handed over, cleared on the last): VAEDecoder.decode, compared after the reference's
    post-processing ((x + 1) / 2 clamp) which the device applies in the same kernel that writes the uint8 frames."""
    from landiff_amd.config import VAEConfig
    from landiff_amd.vae import VAEDecoder
    from landiff_amd.weights import init_state, vae_spec
    from oracle.vae import VAEDecoderOracle, post_process, to_uint8_frames
    g = load("vae_fp32")
    cfg = VAEConfig.tiny()
    sd = init_state(vae_spec(cfg), int(g["seed"]))
    latent = T(g["latent"])                                           # [1, C, T, h, w]
    vae = VAEDecoder(sd, cfg, cuda)
    frames, video = vae.decode(latent.permute(0, 2, 1, 3, 4).contiguous().to(cuda), want_float=True)
    ref = post_process(T(g["chunked"]))[0]                            # [3, 25, 32, 48]
    assert tuple(video.shape) == tuple(ref.shape)
    orc16 = post_process(VAEDecoderOracle(sd, cfg, torch.bfloat16).decode_latent(latent.to(torch.bfloat16).float()))[0]
    check("VAE chunked decode", video, ref, orc16, 2e-2)
    mean_err = (video.cpu() - ref).abs().mean().item()
    assert mean_err < max(2 * (orc16.float() - ref).abs().mean().item(), 4e-3), mean_err
    assert torch.equal(frames.cpu(), to_uint8_frames(video.cpu()))    # integer output: exactly the truncation of the float video
    # the reference's witness that the chunk schedule matters holds for the device too: closer to chunked than to one full pass
    full9 = post_process(T(g["full_first9"]))[0]
    assert (video.cpu()[:, :9] - ref[:, :9]).abs().mean() < (video.cpu()[:, :9] - full9).abs().mean()


# ------------------------------------------------------------------------------------------ DiT
def _dit_runner(cuda, n_img_frames):
    from landiff_amd.config import DiTConfig
    from landiff_amd.dit import ControlDiTRunner
    from landiff_amd.weights import dit_spec, init_state
    cfg = DiTConfig.tiny()
    sd_main, sd_ctrl = init_state(dit_spec(cfg, False), 9), init_state(dit_spec(cfg, True), 10)
    # the golden layer chains run on text_len + ONE frame of image tokens; the position table still covers cfg.latent_frames
    rc = dataclasses.replace(cfg, latent_frames=n_img_frames, pos_frames=cfg.latent_frames)
    return cfg, rc, sd_main, sd_ctrl, ControlDiTRunner(sd_main, sd_ctrl, rc, cuda)


def test_dit_layer_chains_vs_reference_mixins(cuda):
    """ControlOutAdaLNMixin / ControlAdaLNMixin.layer_forward of the reference driven layer by layer (dit_fp32.npz: the control
    chain with its zero-linears, the main chain with the control adds): ControlDiTRunner._control_chain / _main_chain."""
    from oracle.dit import DiTOracle
    g = load("dit_fp32")
    cfg, rc, sd_main, sd_ctrl, run = _dit_runner(cuda, 1)
    d = cfg.hidden
    assert rc.seq_len == g["control_h"].shape[1]
    BF = torch.bfloat16
    # ---- control chain: every layer's zero-linear output
    h, emb = T(g["control_h"]), T(g["control_emb"])
    run.emb.copy_(emb.to(cuda, BF))
    run._modulations(run.ctrl)
    run.hc.copy_(h.reshape(-1, d).to(cuda, BF))
    run._control_chain(run.hc)
    o16 = DiTOracle(sd_ctrl, cfg, True, BF)
    hb = h.to(BF)
    for i in range(cfg.layers_control):
        hb = o16.layer(i, hb, emb.to(BF))
        check(f"DiT control layer {i} (+ zero linear)", run.ctrl_out[i].view(2, -1, d), T(g["control_out"][i]), hb, 2e-2)
    # ---- main chain with the golden control states added
    h, emb, cl = T(g["main_h"]), T(g["main_emb"]), T(g["main_ctrl"])
    run.emb.copy_(emb.to(cuda, BF))
    run._modulations(run.main)
    for i in range(cfg.layers_control):
        run.ctrl_out[i].copy_(cl[i].reshape(-1, d).to(cuda, BF))
    o16 = DiTOracle(sd_main, cfg, False, BF)
    hb = h.to(BF)
    run.h.copy_(h.reshape(-1, d).to(cuda, BF))
    for i in range(cfg.layers_main):
        run._layer(run.main, i, run.h, run.h, run.ctrl_out[i] if i < cfg.layers_control else None)
        hb = o16.layer(i, hb, emb.to(BF), cl[i].to(BF) if i < cfg.layers_control else None)
        check(f"DiT main layer {i}", run.h.view(2, -1, d), T(g["main_out"][i]), hb, 2e-2)


def test_dit_final_layer_and_patch_embed_vs_reference(cuda):
    """FinalLayerMixin.final_forward (+ unpatchify) and ImagePatchEmbeddingMixin.word_embedding_forward of the reference
    (dit_fp32.npz: final_in/final_out, embed_img/embed_ctx/embed_out) through ControlDiTRunner._final / _embed; the CFG
    combine of ld_unpatchify_cfg selects the uncond row at scale 0 and the cond row at scale 1."""
    from oracle.dit import DiTOracle
    g = load("dit_fp32")
    cfg, rc, sd_main, sd_ctrl, run = _dit_runner(cuda, 3)
    assert rc.seq_len == g["final_in"].shape[1]
    d, BF = cfg.hidden, torch.bfloat16
    emb = T(g["main_emb"])
    run.emb.copy_(emb.to(cuda, BF))
    hx = T(g["final_in"])
    ref = T(g["final_out"])                                            # [2, T, C, H, W]
    x0 = torch.zeros(1, *ref.shape[1:], device=cuda)
    out = torch.empty_like(x0)
    # bf16 restatement of final_forward on the same input
    import torch.nn.functional as F
    from oracle.common import layer_norm, linear
    from oracle.dit import modulate
    s = sd_main
    xb = hx[:, cfg.text_len:].to(BF)
    mod = linear(F.silu(emb.to(BF)), s["mixins.final_layer.adaLN_modulation.1.weight"], s["mixins.final_layer.adaLN_modulation.1.bias"], BF)
    shift, scale = mod.chunk(2, dim=1)
    xb = modulate(layer_norm(xb, s["mixins.final_layer.norm_final.weight"], s["mixins.final_layer.norm_final.bias"], cfg.final_ln_eps), shift, scale)
    xb = linear(xb, s["mixins.final_layer.linear.weight"], s["mixins.final_layer.linear.bias"], BF)
    p = cfg.patch
    xb = xb.view(2, cfg.latent_frames, cfg.grid_h, cfg.grid_w, cfg.out_channels, p, p).permute(0, 1, 4, 2, 5, 3, 6).reshape(ref.shape)
    for b, scale_cfg in ((0, 0.0), (1, 1.0)):                          # c_out 1, c_skip 0: out = eps_u + scale * (eps_c - eps_u)
        run._final(hx.reshape(-1, d).to(cuda, BF), x0, 1.0, 0.0, scale_cfg, out, sat_final_layernorm=False)
        check(f"DiT final layer (batch row {b})", out[0], ref[b], xb[b], 2e-2)
    # ---- patch embed + text projection (+ the sin-cos position rows the runner adds in the same epilogue)
    img, ctx, eref = T(g["embed_img"]), T(g["embed_ctx"]), T(g["embed_out"])
    pos = sd_main["mixins.pos_embed.pos_embedding"][0, : rc.seq_len]
    o16 = DiTOracle(sd_main, cfg, False, BF)
    for b in range(2):
        run.set_condition(ctx[b:b + 1], torch.zeros(cfg.latent_frames, cfg.in_channels, cfg.latent_h, cfg.latent_w))
        run._embed(run.main, img[b:b + 1].to(cuda), run.h, run.txt_main, None)
        got = run.h.view(2, rc.seq_len, d)[1]                          # row 1 = the conditional branch (row 0 carries zero text)
        check(f"DiT patch/text embed (sample {b})", got, eref[b] + pos, o16.embed(img[b:b + 1], ctx[b:b + 1])[0], 1e-2)

"""The slow fp32 / bf16 oracle legs of the full-size GPU tests, as functions of seeds only -- so that they can run in child
processes on the host cores WHILE the GPU tests run (tests/conftest.py starts the ones whose tests were selected right after
collection; the tests join them through the `oracle_bg` fixture).  Test infrastructure: imports oracle/, never the HIP library.

Each job has an `*_inputs()` function that rebuilds its weights and inputs from fixed seeds on the CPU -- the test calls the
same function for the device side, so both sides see the same tensors without passing anything between processes -- and a
`job_*()` function that returns what the test compares against.

    python tests/oracle_jobs.py <job> <out.pt> [<input.pt>]      # what conftest.py runs
"""
import dataclasses
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch


def _threads(n):
    """n threads unless the session handed this child its own cores (conftest.py: LD_ORACLE_JOB_THREADS = size of its cpuset)."""
    given = int(os.environ.get("LD_ORACLE_JOB_THREADS", "0"))
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(max(1, min(given or n, avail)))


# ---- test_gpu_fullsize.py::test_vae_full_resolution_two_chunks_vs_oracle ----
def vae_two_chunks_inputs():
    from landiff_amd.config import VAEConfig
    from landiff_amd.weights import init_state, vae_spec
    cfg = VAEConfig()
    sd = init_state(vae_spec(cfg), 41)
    g = torch.Generator().manual_seed(8)
    latent = torch.randn(1, 5, cfg.z_channels, 60, 90, generator=g).to(torch.bfloat16).float()
    return cfg, sd, latent


def job_vae_two_chunks():
    """fp32 oracle video [3, 17, 480, 720] of the 3 + 2 latent-frame chunk schedule (~110 TFLOP of fp32 conv3d)."""
    from oracle.vae import VAEDecoderOracle, post_process
    cfg, sd, latent = vae_two_chunks_inputs()
    _threads(64)
    orc = VAEDecoderOracle(sd, cfg, torch.float32)
    with torch.no_grad():
        return post_process(orc.decode_latent(latent.permute(0, 2, 1, 3, 4)))[0].contiguous()


# ---- test_gpu_llm_longctx.py::test_llm_two_blocks_full_width_decode_at_real_context_lengths_vs_oracle ----
def llm_two_blocks_inputs():
    from landiff_amd.config import LLMConfig
    from landiff_amd.weights import init_state, llm_spec
    from oracle.llm import forced_schedule
    cfg = dataclasses.replace(LLMConfig(), num_layers=2)
    sd = init_state(llm_spec(cfg), 5, dtype=torch.bfloat16)            # CPU stream: the child processes draw the same weights
    g = torch.Generator().manual_seed(6)
    text = torch.randn(64, cfg.text_dim, generator=g).to(torch.bfloat16)
    fed = torch.randint(0, cfg.visual_vocab, (2000,), generator=g)
    S = 64 + 4 - 1                                                     # prefix = [BOS][frames][motion][text x 64][START_I]
    full_len = forced_schedule(cfg, S, 13)[0]
    steps = full_len - (S + 1) - 1
    # decode step `it` appends position S + 1 + it: KV length S + 2 + it
    want_len = [128, 255, 256, 257, 700, 1024, 1300, S + 1 + steps]
    check_it = sorted({L - S - 2 for L in want_len})
    return cfg, sd, text, fed, S, full_len, steps, check_it


def _llm_two_blocks(dtype):
    from oracle.llm import LLMOracle, rope_table
    cfg, sd, text, fed, S, full_len, steps, check_it = llm_two_blocks_inputs()
    _threads(32)
    sdt = {k: (v.to(dtype) if v.dtype == torch.bfloat16 else v) for k, v in sd.items()}     # no per-call weight casts
    orc = LLMOracle(sdt, cfg, dtype)
    with torch.no_grad():
        feats = orc.prefix_features(text.float(), 13.0, 0.1, True)
        assert feats.shape[1] - 1 == S
        cos, sin = rope_table(cfg.head_dim, full_len + 1, cfg.rope_theta)
        cache = [None] * cfg.num_layers
        emb = sd["visual_embedding_model.tok_emb_code.weight"]
        out = {}
        orc.gpt_step(feats, cache, cos[None, : S + 1], sin[None, : S + 1])
        for it in range(steps):
            f = emb[fed[it]].float().reshape(1, 1, -1)
            pos = S + 1 + it
            lg = orc.gpt_step(torch.cat([f, f], 0), cache, cos[None, pos:pos + 1], sin[None, pos:pos + 1]).float()
            if it in check_it:
                out[it] = lg[1:] + 7.5 * (lg[:1] - lg[1:])
    return out


def job_llm_two_blocks_fp32():
    return _llm_two_blocks(torch.float32)


def job_llm_two_blocks_bf16():
    return _llm_two_blocks(torch.bfloat16)


# ---- test_gpu_fullsize.py::test_dit_multi_layer_step_full_shape_vs_oracle ----
def dit_3p3_inputs():
    from landiff_amd.config import PipelineConfig
    from landiff_amd.weights import dit_spec, init_state
    d3 = dataclasses.replace(PipelineConfig.full().dit, layers_main=3, layers_control=3)
    sd_main, sd_ctrl = init_state(dit_spec(d3, False), 1), init_state(dit_spec(d3, True), 2)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, d3.latent_frames, d3.in_channels, d3.latent_h, d3.latent_w, generator=g)
    ctx = torch.randn(1, d3.text_len, d3.text_dim, generator=g).to(torch.bfloat16).float()
    sem = (0.5 * torch.randn(d3.latent_frames, d3.in_channels, d3.latent_h, d3.latent_w, generator=g)).to(torch.bfloat16)
    return d3, sd_main, sd_ctrl, x, ctx, sem, 500


def job_dit_3p3_eps():
    """fp32 oracle noise prediction [2, ...] = [uncond, cond] of the 3 control + 3 main layer denoiser at the BASELINE shape."""
    from oracle.dit import ControlDiTOracle
    d3, sd_main, sd_ctrl, x, ctx, sem, timestep = dit_3p3_inputs()
    _threads(32)
    with torch.no_grad():
        return ControlDiTOracle(sd_main, sd_ctrl, d3, torch.float32)(
            torch.cat([x, x]), torch.full((2,), float(timestep)), torch.cat([torch.zeros_like(ctx), ctx]), sem.float()).float()


# ---- test_gpu_fullsize.py::test_dit_layer_full_shape_vs_oracle ----
def dit_layer_inputs():
    from landiff_amd.config import PipelineConfig
    from landiff_amd.weights import dit_spec, init_state
    d1 = dataclasses.replace(PipelineConfig.full().dit, layers_main=1, layers_control=1)
    sd_main = init_state(dit_spec(d1, False), 1)
    sd_ctrl = init_state(dit_spec(d1, True), 2)
    g = torch.Generator().manual_seed(5)
    h = torch.randn(2, d1.seq_len, d1.hidden, generator=g).to(torch.bfloat16)
    emb = torch.randn(2, d1.time_embed_dim, generator=g).to(torch.bfloat16)
    return d1, sd_main, sd_ctrl, h, emb


def job_dit_layer():
    """fp32 oracle output of one main AdaLN layer at the BASELINE shape (the call bench.py's cpu_baseline times)."""
    from oracle.dit import DiTOracle
    d1, sd_main, _, h, emb = dit_layer_inputs()
    _threads(32)
    with torch.no_grad():
        return DiTOracle(sd_main, d1, False, torch.float32).layer(0, h.float(), emb.float())


# ---- test_gpu_fullsize.py::test_vae_level0_resblock_full_resolution_vs_oracle ----
def vae_level0_inputs():
    from landiff_amd.config import VAEConfig
    from landiff_amd.weights import _res3d, init_state
    cfg = VAEConfig()
    C, T, H, W = 128, 4, 480, 720
    p = "decoder.up.0.block.1."
    sd = init_state(_res3d(p, C, C, cfg.z_channels), 31)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(1, C, T, H, W, generator=g).to(torch.bfloat16)
    zq = torch.randn(1, cfg.z_channels, 1, 60, 90, generator=g).to(torch.bfloat16)
    return cfg, sd, p, C, T, H, W, x, zq


def job_vae_level0():
    """(fp32, bf16) oracle outputs of one level-0 resblock at 480 x 720."""
    from oracle.vae import VAEDecoderOracle
    cfg, sd, p, C, T, H, W, x, zq = vae_level0_inputs()
    _threads(32)
    with torch.no_grad():
        ref32 = VAEDecoderOracle(sd, cfg, torch.float32).resblock(x.float(), zq.float(), p, C, C, True)[0]
        ref16 = VAEDecoderOracle(sd, cfg, torch.bfloat16).resblock(x, zq, p, C, C, True)[0].float()
    return ref32, ref16


# ---- test_facade.py::test_infer_video_entry_point_config0_frames_vs_oracle (started BY the entry-point test, with its own latent) ----
def job_config0_frames(path):
    """fp32 oracle (video [3, T, H, W] in [0, 1], uint8 frames) of the latent in `path` ({"vae_cfg", "vae_sd", "z32"}): PipelineOracle.frames
    of BASELINE configs[0] -- 29 frames at 512 x 512 through the full-width VAE, ~80 s of host time that no longer sits in a test."""
    from oracle.vae import VAEDecoderOracle, post_process, to_uint8_frames
    d = torch.load(path, weights_only=False)
    _threads(64)
    with torch.no_grad():
        rec = VAEDecoderOracle(d["vae_sd"], d["vae_cfg"], torch.float32).decode_latent(d["z32"].permute(0, 2, 1, 3, 4))
        video = post_process(rec)[0]
        return video.contiguous(), to_uint8_frames(video)


JOBS = {
    "vae_two_chunks": job_vae_two_chunks,
    "llm_two_blocks_fp32": job_llm_two_blocks_fp32,
    "llm_two_blocks_bf16": job_llm_two_blocks_bf16,
    "dit_3p3_eps": job_dit_3p3_eps,
    "dit_layer": job_dit_layer,
    "vae_level0": job_vae_level0,
    "config0_frames": job_config0_frames,
}
# jobs that a test starts itself (with an input file) instead of the session at collection time: name -> the test that starts it
LATE_JOBS = {"config0_frames": "test_infer_video_entry_point_config0"}
# which jobs a test (matched by the end of its node id) joins: conftest.py starts exactly these after collection
CONSUMERS = {
    "test_vae_full_resolution_two_chunks_vs_oracle": ["vae_two_chunks"],
    "test_llm_two_blocks_full_width_decode_at_real_context_lengths_vs_oracle": ["llm_two_blocks_fp32", "llm_two_blocks_bf16"],
    "test_dit_multi_layer_step_full_shape_vs_oracle": ["dit_3p3_eps"],
    "test_dit_layer_full_shape_vs_oracle": ["dit_layer"],
    "test_vae_level0_resblock_full_resolution_vs_oracle": ["vae_level0"],
    "test_infer_video_entry_point_config0_frames_vs_oracle": ["config0_frames"],
}


if __name__ == "__main__":
    name, out = sys.argv[1], sys.argv[2]
    t0 = time.perf_counter()
    res = JOBS[name](*sys.argv[3:4])
    torch.save({"result": res, "seconds": time.perf_counter() - t0}, out + ".tmp")
    os.replace(out + ".tmp", out)
    print(f"oracle job {name}: {time.perf_counter() - t0:.1f} s")

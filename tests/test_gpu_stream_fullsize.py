"""BASELINE configs[2] / configs[4] at full size (LanDiff 5B shapes, 49-frame 480 x 720 chunks, random-init weights): the streaming
long-video loop, bf16 and with MXFP8 DiT linears, cut to 2 chunks x 2 sampler steps so that it runs in about a minute.  No
reference loop exists for streaming (the reference ships the primitives), so the full-size checks are the properties
the loop must have whatever the weights are:
  * the pinned prefix: the first 7 latent frames of chunk 1 ARE the last 7 of chunk 0, bit for bit
    (landiff/diffusion/sgm/modules/diffusionmodules/sampling.py:800-835, fixed_frames);
  * the VAE continuation: the frames of decode(chunk 0, keep caches) + decode(6 new latent frames, continue) equal the one-call
    decode of the 19 concatenated latent frames (landiff/diffusion/vae_modules/cp_enc_dec.py:436-466);
  * determinism: a second run gives the same latents and the same frames (fixed-order reductions everywhere, the AR decode
    overlapped on a second stream included);
  * sanity: 73 frames of 480 x 720, finite latents, frames that are not constant.
The tiny-size runs against PipelineOracle.stream are in tests/test_gpu_stages.py and tests/test_gpu_fp8.py."""
import dataclasses

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fp8", [None, "mx"])
def test_streaming_two_chunks_full_size_properties(cuda, fp8):
    from landiff_amd.config import PipelineConfig
    from landiff_amd.pipeline import LanDiffPipeline, synthetic_inputs
    from landiff_amd.weights import init_pipeline_state
    cfg = PipelineConfig.full()
    cfg = dataclasses.replace(cfg, sampler=dataclasses.replace(cfg.sampler, num_steps=2)).check()
    states = init_pipeline_state(cfg, seed=1234, dtype=torch.bfloat16, device=cuda)
    T, prefix, chunks = cfg.dit.latent_frames, 7, 2
    new = T - prefix
    n_seg = -(-(T + (chunks - 1) * new) // cfg.llm.segment_length)
    pipe = LanDiffPipeline(cfg, states, cuda, max_llm_frames=n_seg * cfg.llm.segment_length, fp8_gemm=fp8)
    del states
    torch.cuda.empty_cache()
    inp = synthetic_inputs(cfg, cuda, n_text=64, seed=42)
    lat = []
    frames = pipe.generate_stream(inp, chunks, prefix_frames=prefix, latents_out=lat)
    torch.cuda.synchronize()
    n_frames = 4 * T - 3 + (chunks - 1) * 4 * new
    assert tuple(frames.shape) == (n_frames, 8 * cfg.dit.latent_h, 8 * cfg.dit.latent_w, 3) == (73, 480, 720, 3)
    assert frames.dtype == torch.uint8 and len(lat) == chunks
    assert all(torch.isfinite(z).all() for z in lat) and frames.float().std().item() > 1.0
    # the pinned prefix
    assert torch.equal(lat[1][:, :prefix], lat[0][:, T - prefix:])
    assert not torch.equal(lat[1][:, prefix:], lat[0][:, :new])                       # (and the new frames are new)
    # the VAE continuation against the one-call decode of all 19 latent frames
    whole = torch.cat([lat[0], lat[1][:, prefix:]], dim=1)
    assert whole.shape[1] == T + new == 19
    assert torch.equal(pipe.vae.decode(whole), frames)
    # run-to-run determinism
    lat2 = []
    frames2 = pipe.generate_stream(inp, chunks, prefix_frames=prefix, latents_out=lat2)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(lat, lat2)) and torch.equal(frames, frames2)


def test_generate_many_full_size_equals_per_prompt_runs(cuda):
    """The per-rank path of BASELINE configs[3] (a rank that owns several prompts: LanDiffPipeline.generate_many, the AR decode of
    prompt i + 1 on a second stream under the DiT loop of prompt i) at FULL size, 2 sampler steps: every prompt's frames must be
    exactly those of a plain per-prompt call.  The tiny-size twin (tests/test_gpu_stages.py) takes the plain attention kernel and never
    saw what this size does: before round 5 the overlapped decode sampled other tokens here."""
    from landiff_amd.config import PipelineConfig
    from landiff_amd.pipeline import LanDiffPipeline, synthetic_inputs
    from landiff_amd.weights import init_pipeline_state
    cfg = PipelineConfig.full()
    cfg = dataclasses.replace(cfg, sampler=dataclasses.replace(cfg.sampler, num_steps=2)).check()
    states = init_pipeline_state(cfg, seed=1234, dtype=torch.bfloat16, device=cuda)
    pipe = LanDiffPipeline(cfg, states, cuda)
    del states
    torch.cuda.empty_cache()
    base = synthetic_inputs(cfg, cuda, n_text=64, seed=42)
    inputs = [dataclasses.replace(base, seed=s) for s in (42, 43, 44)]
    want = [pipe(inp).clone() for inp in inputs]
    for rep in range(2):
        got = pipe.generate_many(inputs)
        torch.cuda.synchronize()
        assert len(got) == 3
        for i, (a, b) in enumerate(zip(got, want)):
            assert torch.equal(a, b), (rep, i, (a != b).float().mean().item())
    assert not torch.equal(want[0], want[1])

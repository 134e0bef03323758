"""Which stage makes two runs of the full-size streaming loop differ (2 chunks x 2 sampler steps)?  Prints, for a few settings,
whether tokens / chunk latents / frames of two consecutive runs are equal.  FP8=mx for the MXFP8 configuration."""
import dataclasses, os, sys, torch
sys.path.insert(0, ".")
from landiff_amd.config import PipelineConfig
from landiff_amd.pipeline import LanDiffPipeline, synthetic_inputs
from landiff_amd.weights import init_pipeline_state
dev = torch.device("cuda:0")
cfg = PipelineConfig.full()
cfg = dataclasses.replace(cfg, sampler=dataclasses.replace(cfg.sampler, num_steps=2)).check()
states = init_pipeline_state(cfg, seed=1234, dtype=torch.bfloat16, device=dev)
T, prefix, chunks = cfg.dit.latent_frames, 7, 2
n_seg = -(-(T + (chunks - 1) * (T - prefix)) // cfg.llm.segment_length)
pipe = LanDiffPipeline(cfg, states, dev, max_llm_frames=n_seg * cfg.llm.segment_length, fp8_gemm=os.environ.get("FP8") or None)
del states
inp = synthetic_inputs(cfg, dev, n_text=64, seed=42)
def run(**kw):
    lat = []
    fr = pipe.generate_stream(inp, chunks, prefix_frames=prefix, latents_out=lat, **kw)
    torch.cuda.synchronize()
    return pipe.llm.out_tokens[: n_seg * cfg.tok.num_latent_tokens].clone(), lat, fr
def cmp(a, b, tag):
    d = lambda x, y: "equal" if torch.equal(x, y) else f"DIFFER ({(x.float() - y.float()).abs().max().item():.3g} max, {(x != y).float().mean().item():.2e} of elements)"
    print(f"{tag}: tokens {d(a[0], b[0])}; latent0 {d(a[1][0], b[1][0])}; latent1 {d(a[1][1], b[1][1])}; frames {d(a[2], b[2])}", flush=True)
base = run()
cmp(base, run(), "default vs default")
cmp(base, run(), "default vs default (again)")
ser = run(overlap_decode=False)
cmp(base, ser, "default vs serial decode")
cmp(ser, run(overlap_decode=False), "serial decode vs serial decode")
pipe.dit.overlap = False
s2 = run(overlap_decode=False)
cmp(ser, s2, "serial decode: DiT chains overlapped vs serial")
cmp(s2, run(overlap_decode=False), "all serial vs all serial")

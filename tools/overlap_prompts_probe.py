"""Experiment: AR decode of prompt i+1 on a high-priority stream (own host thread) while prompt i is in the DiT loop + VAE.
Not a product path (the shared default RNG makes the two prompts' draws interleave); it measures what the hardware does with a
980 W HBM-bound stage next to a power-capped MFMA stage."""
import sys, threading, time, torch
sys.path.insert(0, ".")
from landiff_amd.config import PipelineConfig
from landiff_amd.pipeline import LanDiffPipeline, synthetic_inputs
from landiff_amd.weights import init_pipeline_state
dev = torch.device("cuda:0")
cfg = PipelineConfig.full().check()
pipe = LanDiffPipeline(cfg, init_pipeline_state(cfg, seed=1234, dtype=torch.bfloat16, device=dev), dev)
inp = synthetic_inputs(cfg, dev, n_text=64, seed=42)
d = cfg.dit
def rest(tokens):
    sem = pipe.detok.semantic_condition(tokens)
    pipe.dit.set_condition(inp.dit_context, sem)
    noise = torch.randn(1, d.latent_frames, d.in_channels, d.latent_h, d.latent_w, device=dev, dtype=torch.float32)
    z = pipe.sampler.run(pipe.dit.step, noise)
    return pipe.vae.decode(z.to(torch.bfloat16).float())
tokens = pipe.generate_tokens(inp); rest(tokens); torch.cuda.synchronize()          # warm-up
t0 = time.perf_counter(); tokens = pipe.generate_tokens(inp); torch.cuda.synchronize(); t_llm = time.perf_counter() - t0
t0 = time.perf_counter(); rest(tokens); torch.cuda.synchronize(); t_rest = time.perf_counter() - t0
print(f"sequential: AR decode {t_llm:.2f} s + detokenize/DiT/VAE {t_rest:.2f} s = {t_llm + t_rest:.2f} s per video", flush=True)
for prio in (-1, 0):
    s2 = torch.cuda.Stream(device=dev, priority=prio)
    res = {}
    def llm_thread():
        with torch.cuda.stream(s2):
            t = time.perf_counter()
            res["tok"] = pipe.generate_tokens(inp)
            s2.synchronize()
            res["t"] = time.perf_counter() - t
    th = threading.Thread(target=llm_thread)
    t0 = time.perf_counter()
    th.start()
    rest(tokens)
    torch.cuda.current_stream().synchronize()
    t_main = time.perf_counter() - t0
    th.join(); torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"overlapped (decode stream priority {prio}): detokenize/DiT/VAE {t_main:.2f} s, AR decode alongside {res['t']:.2f} s, both done after {t_all:.2f} s "
          f"(vs {t_llm + t_rest:.2f} s sequential)", flush=True)

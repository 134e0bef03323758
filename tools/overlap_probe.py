"""Does the HBM-bound AR decode of prompt i+1 hide behind the MFMA-bound DiT loop of prompt i?  Times 10 sampler steps
alone, ~1/5 of a decode alone (250 steps), and both on two HIP streams."""
import sys, time, torch
sys.path.insert(0, ".")
from landiff_amd.config import PipelineConfig
from landiff_amd.pipeline import LanDiffPipeline, synthetic_inputs
from landiff_amd.weights import init_pipeline_state
dev = torch.device("cuda:0")
cfg = PipelineConfig.full().check()
pipe = LanDiffPipeline(cfg, init_pipeline_state(cfg, seed=1234, dtype=torch.bfloat16, device=dev), dev)
inp = synthetic_inputs(cfg, dev, n_text=64, seed=42)
d = cfg.dit
tokens = torch.randint(0, 2048, (1218,), device=dev)
sem = pipe.detok.semantic_condition(tokens)
pipe.dit.set_condition(inp.dit_context, sem)
x = torch.randn(1, d.latent_frames, d.in_channels, d.latent_h, d.latent_w, device=dev)
out = torch.empty_like(x)
plan = pipe.sampler.plan
def dit_steps(n):
    for sp in plan[:n]:
        pipe.dit.step(x, sp.timestep, sp.c_out, sp.c_skip, sp.cfg_scale, out)
# LLM: prime a decode state and capture the step graph
llm = pipe.llm
llm.sample(inp.llm_text_emb, seed=42)          # warm
gen = torch.Generator(device=dev); gen.manual_seed(1)
graph = llm._capture(True, 7.5, 1.0, gen)
llm.pos.fill_(100); llm.out_count.zero_()
def llm_steps(n):
    for _ in range(n): graph.replay()
def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return time.perf_counter() - t0
dit_steps(2); llm_steps(10)
t_d = timed(lambda: dit_steps(10))
t_l = timed(lambda: llm_steps(250))
sl = torch.cuda.Stream()
def both():
    with torch.cuda.stream(sl):
        llm_steps(250)
    dit_steps(10)
llm.pos.fill_(100)
t_b = timed(both)
print(f"DiT 10 steps alone {t_d:.3f} s | LLM 250 steps alone {t_l:.3f} s | both concurrently {t_b:.3f} s (sum {t_d + t_l:.3f})")

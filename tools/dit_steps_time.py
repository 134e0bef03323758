"""Times a few full-size denoiser steps (control + main DiT, B = 2 CFG pair).  FP8=row|mx selects the fp8 linear modes."""
import os, sys, time, torch
sys.path.insert(0, ".")
from landiff_amd.config import PipelineConfig
from landiff_amd.dit import ControlDiTRunner
from landiff_amd.weights import init_pipeline_state
dev = torch.device("cuda:0")
cfg = PipelineConfig.full().check()
st = init_pipeline_state(cfg, seed=1234, dtype=torch.bfloat16, device=dev, parts=["dit_main", "dit_control"])
fp8 = os.environ.get("FP8") or False
run = ControlDiTRunner(st["dit_main"], st["dit_control"], cfg.dit, dev, fp8_gemm=fp8)
d = cfg.dit
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(1, d.latent_frames, d.in_channels, d.latent_h, d.latent_w, device=dev, generator=g)
run.set_condition(torch.randn(1, d.text_len, d.text_dim, device=dev, generator=g),
                  torch.randn(d.latent_frames, d.in_channels, d.latent_h, d.latent_w, device=dev, generator=g).to(torch.bfloat16))
out = torch.empty_like(x)
run.step(x, 500, -0.7, 0.7, 6.0, out); torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
t0 = time.perf_counter()
for i in range(n): run.step(x, 500 - i, -0.7, 0.7, 6.0, out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"fp8={fp8}: {dt * 1e3:.1f} ms per denoiser step, checksum {out.double().sum().item():.6f}")

"""Host-side launch rate of the decode step vs its GPU time: enqueue N steps on an idle queue (host clock), then sync."""
import sys, time, torch
sys.path.insert(0, ".")
from landiff_amd.config import LLMConfig
from landiff_amd.llm import LLMRunner
from landiff_amd.weights import init_state, llm_spec
dev = torch.device("cuda:0")
cfg = LLMConfig()
run = LLMRunner(init_state(llm_spec(cfg), 1, dtype=torch.bfloat16, device=dev), cfg, dev)
text = torch.randn(64, cfg.text_dim, device=dev)
run.sample(text, guidance_scale=7.5, seed=42)
run.pos.fill_(600)
for name, fn in (("native", run._decode_forward), ("per-op", run._decode_forward_per_op)):
    for n in (5, 20, 100):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{name:7s} {n:4d} steps: host enqueue {(t1 - t0) / n * 1e3:.3f} ms/step, until drained {(t2 - t0) / n * 1e3:.3f} ms/step")

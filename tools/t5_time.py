"""T5-XXL-size encoder (random weights) on the MI355X kernels: time per prompt at 64, 226 and 512 tokens."""
import sys, time, torch
sys.path.insert(0, ".")
from landiff_amd.t5 import T5Config, T5EncoderRunner, random_state
cfg = T5Config()
dev = torch.device("cuda:0")
run = T5EncoderRunner(random_state(cfg, 0, dev), cfg, dev)
inner = cfg.heads * cfg.d_kv
for n in (64, 226, 512):
    ids = torch.randint(0, cfg.vocab, (n,), device=dev)
    run.encode(ids); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): out = run.encode(ids)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    flops = 2 * n * cfg.layers * (4 * inner * cfg.d_model + 3 * cfg.d_ff * cfg.d_model)
    print(f"T5-XXL encoder n={n}: {dt*1e3:.1f} ms  ({flops/dt/1e12:.0f} TFLOP/s on the projections), finite={bool(torch.isfinite(out.float()).all())}", flush=True)

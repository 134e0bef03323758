"""T5-XXL-size encoder (random weights) on the MI355X kernels: time per prompt at 226 and 512 tokens."""
import sys, time, torch
sys.path.insert(0, ".")
from landiff_amd.t5 import T5Config, T5EncoderRunner
cfg = T5Config()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
def rnd(*s, sc=0.02): return (torch.randn(*s, device=dev, generator=g) * sc).to(torch.bfloat16)
sd = {"shared.weight": rnd(cfg.vocab, cfg.d_model, sc=1.0), "encoder.final_layer_norm.weight": torch.ones(cfg.d_model),
      "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight": rnd(cfg.num_buckets, cfg.heads, sc=1.0)}
inner = cfg.heads * cfg.d_kv
for i in range(cfg.layers):
    p = f"encoder.block.{i}.layer."
    for nm, shp in (("0.SelfAttention.q.weight", (inner, cfg.d_model)), ("0.SelfAttention.k.weight", (inner, cfg.d_model)),
                    ("0.SelfAttention.v.weight", (inner, cfg.d_model)), ("0.SelfAttention.o.weight", (cfg.d_model, inner)),
                    ("1.DenseReluDense.wi_0.weight", (cfg.d_ff, cfg.d_model)), ("1.DenseReluDense.wi_1.weight", (cfg.d_ff, cfg.d_model)),
                    ("1.DenseReluDense.wo.weight", (cfg.d_model, cfg.d_ff))):
        sd[p + nm] = rnd(*shp)
    sd[p + "0.layer_norm.weight"] = torch.ones(cfg.d_model); sd[p + "1.layer_norm.weight"] = torch.ones(cfg.d_model)
run = T5EncoderRunner(sd, cfg, dev)
for n in (64, 226, 512):
    ids = torch.randint(0, cfg.vocab, (n,), device=dev)
    run.encode(ids); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): out = run.encode(ids)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    flops = 2 * n * cfg.layers * (4 * inner * cfg.d_model + 3 * cfg.d_ff * cfg.d_model)
    print(f"T5-XXL encoder n={n}: {dt*1e3:.1f} ms  ({flops/dt/1e12:.0f} TFLOP/s on the projections), finite={bool(torch.isfinite(out.float()).all())}", flush=True)

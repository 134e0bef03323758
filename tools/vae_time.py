"""Times the full-size VAE decode (13 latent frames 60x90 -> 49 frames 480x720, chunk schedule 3,2,2,2,2,2)."""
import sys, time, torch
sys.path.insert(0, ".")
from landiff_amd.config import PipelineConfig
from landiff_amd.vae import VAEDecoder
from landiff_amd.weights import init_pipeline_state
dev = torch.device("cuda:0")
cfg = PipelineConfig.full().check()
st = init_pipeline_state(cfg, seed=1234, dtype=torch.bfloat16, device=dev, parts=["vae"])
vae = VAEDecoder(st["vae"], cfg.vae, dev)
d = cfg.dit
lat = torch.randn(1, d.latent_frames, d.in_channels, d.latent_h, d.latent_w, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
out = vae.decode(lat); torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
t0 = time.perf_counter()
for _ in range(n): out = vae.decode(lat)
torch.cuda.synchronize()
print(f"VAE decode: {(time.perf_counter() - t0) / n * 1e3:.1f} ms per video, checksum {out.double().sum().item():.3f}")

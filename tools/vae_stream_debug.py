import sys, torch
sys.path.insert(0, ".")
from landiff_amd.config import PipelineConfig
from landiff_amd.weights import init_pipeline_state
from landiff_amd.vae import VAEDecoder
cuda = torch.device("cuda:0")
cfg = PipelineConfig.tiny(num_steps=3).check()
st = init_pipeline_state(cfg, seed=1234)
d = cfg.dit
g = torch.Generator().manual_seed(2)
latent = torch.randn(1, 5, d.in_channels, 8, 12, generator=g).to(torch.bfloat16).float()
vae = VAEDecoder(st["vae"], cfg.vae, cuda)
f1 = vae.decode(latent.to(cuda))
f2 = vae.decode(latent.to(cuda))
print("run-to-run", (f1.int() - f2.int()).abs().amax(dim=(1, 2, 3)).tolist())
fa = vae.decode(latent[:, :3].to(cuda), stream_keep=True)
print("cache keys", len(vae.cache))
fb = vae.decode(latent[:, 3:5].to(cuda), stream_continue=True)
fs = torch.cat([fa, fb], 0)
print("stream vs whole per frame", (fs.int() - f1.int()).abs().amax(dim=(1, 2, 3)).tolist())

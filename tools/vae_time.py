"""Times the full-size VAE decode (13 latent frames 60x90 -> 49 frames 480x720, chunk schedule 3,2,2,2,2,2).
usage: python tools/vae_time.py [decodes per arm] [ab | rcp | m512]   -- `ab`: GroupNorm statistics from the conv epilogues vs the
separate pass (VAEDecoder.fuse_gn_stats); `rcp`: the sigmoid's reciprocal in GroupNorm apply as v_rcp_f32 vs the IEEE division
(LD_GN_FAST_RCP); `m512`: the 512 x 128 tile of the 8-phase loop vs the 128 x 128 tiles (LD_GEMM_M512); arms alternated in one process, frames compared."""
import os, sys, time, torch
os.environ.setdefault("LD_TUNING", "1")          # the library re-reads its knobs per call
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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


def timed():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        o = vae.decode(lat)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, o


if len(sys.argv) > 2 and sys.argv[2] == "ab":
    outs = {}
    for rnd in range(3):
        for fuse in (True, False):
            vae.fuse_gn_stats = fuse
            vae.decode(lat)
            ms, o = timed()
            outs[fuse] = o
            print(f"round {rnd}: GroupNorm statistics {'from the conv epilogues' if fuse else 'as a separate pass   '}: {ms:.1f} ms per video")
    a, b = outs[True].int(), outs[False].int()
    diff = (a - b).abs()
    print(f"frames: {(diff > 0).float().mean().item():.4f} of the uint8 values differ, max {diff.max().item()} grey levels")
elif len(sys.argv) > 2 and sys.argv[2] in ("rcp", "m512"):
    knob, names = {"rcp": ("LD_GN_FAST_RCP", ("sigmoid reciprocal v_rcp_f32    ", "sigmoid reciprocal IEEE division")),
                   "m512": ("LD_GEMM_M512", ("Cin = Cout = 128 convs on 512 x 128 8-phase tiles ", "Cin = Cout = 128 convs on 128 x 128 two-stage tiles"))}[sys.argv[2]]
    outs = {}
    for rnd in range(3):
        for fast in ("1", "0"):
            os.environ[knob] = fast
            vae.decode(lat)
            ms, o = timed()
            outs[fast] = o
            print(f"round {rnd}: {names[0] if fast == '1' else names[1]}: {ms:.1f} ms per video")
    a, b = outs["1"].int(), outs["0"].int()
    diff = (a - b).abs()
    print(f"frames: {(diff > 0).float().mean().item():.4f} of the uint8 values differ, max {diff.max().item()} grey levels")
else:
    ms, out = timed()
    print(f"VAE decode: {ms:.1f} ms per video, checksum {out.double().sum().item():.3f}")

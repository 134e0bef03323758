import sys, time, torch, collections
sys.path.insert(0, ".")
from landiff_amd import ops
from landiff_amd.config import PipelineConfig
from landiff_amd.vae import VAEDecoder
from landiff_amd.weights import init_pipeline_state
dev = torch.device("cuda:0")
cfg = PipelineConfig.full().check()
st = init_pipeline_state(cfg, seed=1234, dtype=torch.bfloat16, device=dev, parts=["vae"])
vae = VAEDecoder(st["vae"], cfg.vae, dev)
d = cfg.dit
lat = torch.randn(1, d.latent_frames, d.in_channels, d.latent_h, d.latent_w, device=dev)
vae.decode(lat); torch.cuda.synchronize()
orig = ops.conv_cl
stats = collections.OrderedDict()
def timed(xp, w, T, H, W, out=None, **epi):
    Cout, kT, kH, kW, Cin = w.shape
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig(xp, w, T, H, W, out=out, **epi); e1.record(); torch.cuda.synchronize()
    key = (T, H, W, Cin, Cout, kT, kH, kW)
    fl = 2.0 * T * H * W * Cout * kT * kH * kW * Cin
    s = stats.setdefault(key, [0, 0.0, 0.0]); s[0] += 1; s[1] += e0.elapsed_time(e1); s[2] += fl
    return r
ops.conv_cl = timed
import landiff_amd.vae as V
V.ops.conv_cl = timed
vae.decode(lat)
tot_ms = sum(s[1] for s in stats.values()); tot_fl = sum(s[2] for s in stats.values())
print(f"convs: {tot_ms:.1f} ms, {tot_fl/1e12:.1f} TFLOP -> {tot_fl/tot_ms/1e9:.0f} TFLOP/s")
for k, s in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    print(f"T{k[0]} {k[1]}x{k[2]} Cin{k[3]} Cout{k[4]} k{k[5]}{k[6]}{k[7]}: {s[0]:3d} calls {s[1]:7.1f} ms  {s[2]/s[1]/1e9:6.0f} TFLOP/s  ({s[1]/s[0]*1e3:.0f} us each)")

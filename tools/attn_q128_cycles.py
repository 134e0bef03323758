"""Shader cycles vs wall time of the q128 attention workgroups (timing builds of ld_attn_q128.hip, LD_Q128_ABLATE bit 256):
separates what a variant costs in ISSUE CYCLES from what it costs in CLOCK (the chip runs these kernels under its power cap).
  LANDIFF_HIP_LIB=landiff_amd/variants/libq128_abl<bits>.so python tools/attn_q128_cycles.py
Prints per variant: launch ms, median workgroup cycles, cycles per 32-key half, effective shader clock (cycles / 100 MHz ticks)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from landiff_amd import ops
os.environ["LD_ATTN_Q128"] = "1"
B, H, N = 2, 30, 17776
Npad = (N + 127) // 128 * 128
q = torch.randn(B, H, Npad, 64, device="cuda").to(torch.bfloat16)
k = torch.randn(B, H, Npad, 64, device="cuda").to(torch.bfloat16)
vt = torch.randn(B, H, 64, Npad, device="cuda").to(torch.bfloat16)
out = torch.empty(B, N, H * 64, device="cuda", dtype=torch.bfloat16)
nwg = B * H * ((Npad + 511) // 512)
dbg = torch.zeros(nwg * 2, device="cuda", dtype=torch.int64)
for _ in range(3):
    ops.attn_fwd(q, k, vt, out, N, N, 0.125, kt_min=dbg.view(torch.int32))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.attn_fwd(q, k, vt, out, N, N, 0.125, kt_min=dbg.view(torch.int32))
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
d = dbg.view(-1, 2).cpu().double()
cyc, real = d[:, 0], d[:, 1]
halves = 2 * ((N + 63) // 64) + 4
mhz = (cyc / real * 100.0)
print(f"{os.environ.get('LANDIFF_HIP_LIB', 'default').split('/')[-1]}: {ms:.3f} ms; workgroup cycles median {cyc.median():.0f} (min {cyc.min():.0f}, max {cyc.max():.0f}); "
      f"{cyc.median() / halves:.0f} cycles per half; shader clock median {mhz.median():.0f} MHz (min {mhz.min():.0f}, max {mhz.max():.0f}); "
      f"sum of workgroup wall / 256 CUs = {real.sum() / 100.0 / 256 / 1000:.3f} ms")

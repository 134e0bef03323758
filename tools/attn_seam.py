"""Per-block fixed cost of the DiT attention launch: B=2,H=30,Nq=17776 with the key count varied -- the launch time is
blocks x (seam + key tiles x period); the intercept of the line through the points is the seam (prologue + O stores + queue pop)."""
import sys, torch
sys.path.insert(0, ".")
from landiff_amd import ops
B, H, Nq = 2, 30, 17776
Npad = (Nq + 127) // 128 * 128
q = torch.randn(B, H, Npad, 64, device="cuda").to(torch.bfloat16)
k = torch.randn(B, H, Npad, 64, device="cuda").to(torch.bfloat16)
vt = torch.randn(B, H, 64, Npad, device="cuda").to(torch.bfloat16)
out = torch.empty(B, Nq, H * 64, device="cuda", dtype=torch.bfloat16)
pts = []
for rep in range(2):
    for Nk in (17776, 13312, 8960, 4480, 2304, 1152):
        for _ in range(3):
            ops.attn_fwd(q, k, vt, out, Nq, Nk, 0.125)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 10
        e0.record()
        for _ in range(it):
            ops.attn_fwd(q, k, vt, out, Nq, Nk, 0.125)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / it
        pts.append((Nk, ms))
        print(f"Nk={Nk:6d}: {ms:.4f} ms  {4*B*H*Nq*Nk*64/ms/1e9:.0f} TF  kernel={ops.attn_last_kernel() if hasattr(ops,'attn_last_kernel') else ''}", flush=True)
import numpy as np
x = np.array([p[0] for p in pts], float); y = np.array([p[1] for p in pts])
a, b = np.polyfit(x, y, 1)
print(f"fit: ms = {b:.4f} + {a*128:.6f} per key tile of 128;  intercept = {b/ (a*17776+b)*100:.1f} % of the full launch = {b/(a*128):.1f} key-tile periods per block")

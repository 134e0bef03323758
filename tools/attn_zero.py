"""DVFS probe for the DiT attention launch: random vs zero vs constant inputs (same instruction stream)."""
import sys, os, torch
sys.path.insert(0, ".")
from landiff_amd import ops
B, H, N = 2, 30, 17776
Npad = (N + 127) // 128 * 128
out = torch.empty(B, N, H * 64, device="cuda", dtype=torch.bfloat16)
for fill in ("randn", "zeros", "const"):
    def mk(*s):
        if fill == "randn": return torch.randn(*s, device="cuda").to(torch.bfloat16)
        if fill == "zeros": return torch.zeros(*s, device="cuda", dtype=torch.bfloat16)
        return torch.full(s, 0.25, device="cuda", dtype=torch.bfloat16)
    q, k, vt = mk(B, H, Npad, 64), mk(B, H, Npad, 64), mk(B, H, 64, Npad)
    for _ in range(3): ops.attn_fwd(q, k, vt, out, N, N, 0.125)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.attn_fwd(q, k, vt, out, N, N, 0.125)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"fill={fill}: {ms:.3f} ms  {4*B*H*N*N*64/ms/1e9:.0f} TF", flush=True)

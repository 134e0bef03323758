"""A few DiT-size attention launches (target for rocprofv3 --pmc)."""
import sys, os, torch
sys.path.insert(0, ".")
from landiff_amd import ops
B, H, N = 2, 30, 17776
Npad = (N + 127) // 128 * 128
q = torch.randn(B, H, Npad, 64, device="cuda").to(torch.bfloat16)
k = torch.randn(B, H, Npad, 64, device="cuda").to(torch.bfloat16)
vt = torch.randn(B, H, 64, Npad, device="cuda").to(torch.bfloat16)
out = torch.empty(B, N, H * 64, device="cuda", dtype=torch.bfloat16)
for _ in range(3): ops.attn_fwd(q, k, vt, out, N, N, 0.125)
torch.cuda.synchronize()

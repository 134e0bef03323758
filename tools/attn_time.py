"""Times the DiT-size attention launch (B=2,H=30,N=17776) for the variant in LD_ATTN_VARIANT."""
import sys, os, torch
sys.path.insert(0, ".")
from landiff_amd import ops
B, H, N = 2, 30, int(os.environ.get("N", "17776"))
Npad = (N + 127) // 128 * 128
q = torch.randn(B, H, Npad, 64, device="cuda").to(torch.bfloat16)
k = torch.randn(B, H, Npad, 64, device="cuda").to(torch.bfloat16)
vt = torch.randn(B, H, 64, Npad, device="cuda").to(torch.bfloat16)
out = torch.empty(B, N, H * 64, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.attn_fwd(q, k, vt, out, N, N, 0.125)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
it = 10
e0.record()
for _ in range(it):
    ops.attn_fwd(q, k, vt, out, N, N, 0.125)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / it
print(f"variant={os.environ.get('LD_ATTN_VARIANT','0')} N={N}: {ms:.3f} ms  {4*B*H*N*N*64/ms/1e9:.0f} TF", flush=True)

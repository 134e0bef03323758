"""DiT attention launch at different head counts: how much the partial last round of workgroups costs (2 WGs per CU, 512 slots)."""
import os, sys, torch
sys.path.insert(0, ".")
from landiff_amd import ops
N = 17776; Npad = (N + 127) // 128 * 128
def run(B, H):
    q = torch.randn(B, H, Npad, 64, device="cuda").to(torch.bfloat16); k = torch.randn_like(q)
    vt = torch.randn(B, H, 64, Npad, device="cuda").to(torch.bfloat16)
    out = torch.empty(B, N, H * 64, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): ops.attn_fwd(q, k, vt, out, N, N, 0.125)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.attn_fwd(q, k, vt, out, N, N, 0.125)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    rows = 128 if os.environ.get("LD_ATTN_Q64") == "0" else 256
    wgs = B * H * ((Npad + rows - 1) // rows)
    print(f"B*H={B*H:3d}: {wgs} workgroups = {wgs/512:6.3f} rounds  {ms:.3f} ms  {4*B*H*N*N*64/ms/1e9:.0f} TF  ({ms/(wgs/512)*1e3:.1f} us per round-equivalent)", flush=True)
for rep in range(2):
    for B, H in ((1, 51), (1, 55), (1, 58), (1, 59), (2, 30), (1, 62), (1, 64), (1, 66), (1, 73)): run(B, H)

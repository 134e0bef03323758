"""Debug aid for ld_attn_q128: error map of the fast pass and the forced safe pass against torch fp32 (small problems)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from landiff_amd import ops, _lib
torch.manual_seed(0)
def run(B, H, N, knob):
    os.environ["LD_ATTN_Q128"] = knob
    q = torch.randn(B, H, N, 64).cuda().bfloat16(); k = torch.randn(B, H, N, 64).cuda().bfloat16(); v = torch.randn(B, H, N, 64).cuda().bfloat16()
    Npad = (N + 127) // 128 * 128
    def pack(x):
        o = torch.zeros(B, H, Npad, 64, device="cuda", dtype=x.dtype); o[:, :, :N] = x; return o
    out = torch.zeros(B, N, H * 64, device="cuda", dtype=torch.bfloat16)
    ops.attn_fwd(pack(q), pack(k), pack(v).transpose(2, 3).contiguous(), out, N, N, 0.125)
    name = _lib.load().ld_attn_last_kernel().decode()
    s = (q.float() @ k.float().transpose(-1, -2)) * 0.125
    ref = (torch.softmax(s, -1) @ v.float()).permute(0, 2, 1, 3).reshape(B, N, H * 64)
    o = out.float()
    nan = torch.isnan(o)
    err = (o - ref).abs()
    err[nan] = 9.0
    print(f"[{name}] N={N}: nan frac {nan.float().mean().item():.4f}, max err {err.max().item():.4f}, mean err {err.mean().item():.5f}")
    if err.max() > 0.05:
        rows = (err.max(dim=2).values[0] > 0.05).nonzero().flatten().tolist()
        cols = (err.max(dim=1).values[0] > 0.05).nonzero().flatten().tolist()
        print("  bad rows:", len(rows), rows[:40], "...", rows[-8:])
        print("  bad cols:", len(cols), cols[:70])
        r0 = rows[0]
        print("  row", r0, "got", o[0, r0, :8].tolist(), "ref", ref[0, r0, :8].tolist())
for N in (1152, 2048, 1122):
    run(1, 1, N, "0")
    run(1, 1, N, "2")

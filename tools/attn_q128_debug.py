"""Debug aid for ld_attn_q128: error map against torch fp32 on the overflow-fallback input (LD_ATTN_SAFE=1: forced safe pass)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from landiff_amd import ops, _lib
def run(B, H, N, knob, q_scale, spike):
    os.environ["LD_ATTN_Q128"] = knob
    g = torch.Generator().manual_seed(0)
    q = (torch.randn(B, H, N, 64, generator=g) * q_scale).cuda().bfloat16(); k = torch.randn(B, H, N, 64, generator=g).cuda().bfloat16(); v = torch.randn(B, H, N, 64, generator=g).cuda().bfloat16()
    if spike:
        k[:, :, N - 3] = q[:, :, 5] * 4
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
    err = (o - ref).abs(); err[nan] = 9.0
    print(f"[{name}] N={N} q_scale={q_scale} spike={spike}: nan frac {nan.float().mean().item():.4f}, max err {err.max().item():.4f} (ref absmax {ref.abs().max().item():.2f})")
    if nan.any():
        rows = nan.any(dim=2)[0].nonzero().flatten().tolist()
        cols = nan.any(dim=1)[0].nonzero().flatten().tolist()
        print("  nan rows:", len(rows), rows[:24], "...", rows[-6:], " nan cols:", len(cols), cols[:12])
for qs, sp in ((6.0, True), (6.0, False), (1.0, True)):
    run(1, 2, 1122, "0", qs, sp)
    run(1, 2, 1122, "2", qs, sp)

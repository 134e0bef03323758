"""Correctness of the attention variant in LD_ATTN_VARIANT vs an fp32 reference, including a forced running-max rebase."""
import sys, os, torch
sys.path.insert(0, ".")
from landiff_amd import ops
torch.manual_seed(0)
def run(B, H, N, spike=False, scale_q=1.0):
    q = (torch.randn(B, H, N, 64) * scale_q).cuda().bfloat16(); k = torch.randn(B, H, N, 64).cuda().bfloat16(); v = torch.randn(B, H, N, 64).cuda().bfloat16()
    if spike:    # one key row far larger than the rest, late in the sequence: forces the rebase branch mid-stream
        k[:, :, N * 2 // 3] *= 12.0
        k[:, :, 5] *= 6.0
    Npad = (N + 127) // 128 * 128
    def pack(x):
        o = torch.zeros(B, H, Npad, 64, device="cuda", dtype=x.dtype); o[:, :, :N] = x; return o
    out = torch.zeros(B, N, H * 64, device="cuda", dtype=torch.bfloat16)
    ops.attn_fwd(pack(q), pack(k), pack(v).transpose(2, 3).contiguous(), out, N, N, 0.125)
    s = (q.float() @ k.float().transpose(-1, -2)) * 0.125
    ref = (torch.softmax(s, -1) @ v.float()).permute(0, 2, 1, 3).reshape(B, N, H * 64)
    err = (out.float() - ref).abs()
    print(f"B={B} H={H} N={N} spike={spike} sq={scale_q}: max err {err.max().item():.4f} mean {err.mean().item():.6f} ref absmax {ref.abs().max().item():.2f}", flush=True)
for N in (1152 - 30, 1400, 2176 - 1, 4000):
    run(1, 2, N)
run(2, 3, 1152 - 30, spike=True)
run(1, 2, 1400, spike=True, scale_q=3.0)

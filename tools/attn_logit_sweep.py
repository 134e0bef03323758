"""Prices the attention kernel's max-free fast pass against the logit range: the DiT shape (B 2, H 30, N 17 776,
D 64) with LayerNorm-like q, k ~ N(0, 1) plus ONE SINK KEY per head (key 0: a text token) whose logit q.k/8 against the queries of a
chosen fraction of the 256-row query blocks is T.  Per (T, fraction): launch ms, the fraction of workgroups whose fast pass left
the window (recomputed here from the data: a row's denominator sum_k 2^(log2e q.k/8) outside [2^-80, 2^110]), and the output against
torch fp32 softmax on sampled heads.  Also the forced running-max pass (LD_ATTN_SAFE=1) on benign data: the fallback's own cost (a
workgroup that falls back has already spent one fast pass: 1 + that factor).

usage: python tools/attn_logit_sweep.py [reps]          (LD_TUNING=1 is set here: the knob is re-read per call)"""
import math, os, sys
os.environ.setdefault("LD_TUNING", "1")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from landiff_amd import ops

B, H, N, D = 2, 30, 17776, 64
Npad = (N + 127) // 128 * 128
ROWS = 256                                   # query rows per workgroup (ld_attn_q64)
NBLK = (Npad + ROWS - 1) // ROWS
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(3)
q0 = torch.zeros(B, H, Npad, D, device=dev, dtype=torch.bfloat16); k0 = torch.zeros_like(q0)
vt = torch.zeros(B, H, D, Npad, device=dev, dtype=torch.bfloat16)
q0[:, :, :N] = torch.randn(B, H, N, D, device=dev, generator=g).to(torch.bfloat16)
k0[:, :, :N] = torch.randn(B, H, N, D, device=dev, generator=g).to(torch.bfloat16)
vt[:, :, :, :N] = torch.randn(B, H, D, N, device=dev, generator=g).to(torch.bfloat16)
out = torch.empty(B, N, H * D, device=dev, dtype=torch.bfloat16)
# which query blocks see the sink, per fraction: a fixed pseudo-random order of the B * H * NBLK workgroups
order = torch.rand(B, H, NBLK, device=dev, generator=g)


def make(T, frac):
    """q, k with key 0 of every head = (beta, 0, ...) and coordinate 0 of the selected blocks' queries = gamma, beta gamma / 8 = T."""
    q, k = q0.clone(), k0.clone()
    if T is None:
        return q, k, torch.zeros(B, H, NBLK, dtype=torch.bool, device=dev)
    beta = gamma = math.sqrt(8.0 * abs(T))
    if T > 0:
        k[:, :, 0, :] = 0
        k[:, :, 0, 0] = beta
    else:                                                           # T < 0: EVERY key's logit is ~T (the underflow side of the window)
        k[:, :, :N, 0] = beta
    sel = order < frac                                              # [B, H, NBLK]
    rows = sel[:, :, :, None].expand(B, H, NBLK, ROWS).reshape(B, H, NBLK * ROWS)[:, :, :Npad]
    q[:, :, :, 0] = torch.where(rows, torch.full_like(q[:, :, :, 0], gamma if T > 0 else -gamma), q[:, :, :, 0])
    q[:, :, N:] = 0
    return q, k, sel


def left_window(q, k):
    """Fraction of workgroups with a row whose fast-pass denominator leaves [2^-80, 2^110] (and the largest log2 denominator)."""
    c = 0.125 * math.log2(math.e)
    bad_blocks, total, top = 0, 0, -1e30
    for b in range(B):
        for h in range(H):
            qs = (q[b, h, :N].float() * c).to(torch.bfloat16).float()          # the kernel pre-scales q and rounds it to bf16
            kk = k[b, h, :N].float()
            lse = torch.empty(N, device=dev)
            for r0 in range(0, N, 4096):
                s = qs[r0:r0 + 4096] @ kk.T                                     # log2-domain scores
                lse[r0:r0 + 4096] = torch.logsumexp(s * math.log(2.0), dim=1) / math.log(2.0)
            bad = (lse > 110.0) | (lse < -80.0)
            pad = torch.zeros(NBLK * ROWS, dtype=torch.bool, device=dev); pad[:N] = bad
            bad_blocks += int(pad.view(NBLK, ROWS).any(1).sum()); total += (N + ROWS - 1) // ROWS
            top = max(top, float(lse.max()))
    return bad_blocks / total, top


def error_vs_fp32(q, k, heads=((0, 0), (1, 17))):
    worst = 0.0
    for b, h in heads:
        s = (q[b, h, :N].float() @ k[b, h, :N].float().T) * 0.125
        ref = torch.softmax(s, dim=1) @ vt[b, h, :, :N].float().T               # [N, D]
        got = out[b, :, h * D:(h + 1) * D].float()
        worst = max(worst, float((got - ref).abs().max() / ref.abs().max()))
    return worst


def time_launch(q, k, exact=False):
    for _ in range(2):
        ops.attn_fwd(q, k, vt, out, N, N, 0.125, exact=exact)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.attn_fwd(q, k, vt, out, N, N, 0.125, exact=exact); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


print(f"# ld_attn_fwd_bf16, B {B} H {H} N {N} D {D}, {B * H * ((N + ROWS - 1) // ROWS)} workgroups of {ROWS} query rows; median of {reps} launches")
q, k, _ = make(None, 0)
base = time_launch(q, k)
print(f"benign (row max of q.k/8 ~ 4.5)            : {base:.3f} ms   error vs fp32 {error_vs_fp32(q, k):.4f}")
os.environ["LD_ATTN_SAFE"] = "1"
safe = time_launch(q, k)
os.environ["LD_ATTN_SAFE"] = "0"
print(f"benign, running-max pass forced            : {safe:.3f} ms = {safe / base:.2f} x the fast pass   error vs fp32 {error_vs_fp32(q, k):.4f}")
ex = time_launch(q, k, exact=True)
print(f"benign, ld_attn_fwd_bf16_exact (two passes): {ex:.3f} ms = {ex / base:.2f} x the fast pass   error vs fp32 {error_vs_fp32(q, k):.4f}")
for T, frac in [(10, 1.0), (30, 1.0), (50, 1.0), (70, 1.0), (76, 1.0), (80, 1.0), (90, 1.0), (110, 1.0), (90, 0.01), (90, 0.10), (90, 0.5), (-70, 1.0)]:
    q, k, sel = make(T, frac)
    ms = time_launch(q, k)
    fb, top = left_window(q, k)
    err = error_vs_fp32(q, k)
    ex = time_launch(q, k, exact=True)
    err_ex = error_vs_fp32(q, k)
    print(f"sink logit {T:4d} for {frac * 100:5.1f} % of the blocks : {ms:.3f} ms = {ms / base:.2f} x   workgroups that re-ran {fb * 100:5.1f} %   "
          f"largest log2 denominator {top:6.1f}   error vs fp32 {err:.4f}   | exact form {ex:.3f} ms = {ex / base:.2f} x, error {err_ex:.4f}", flush=True)

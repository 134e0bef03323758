"""Correctness (vs torch fp32) and timing of the GEMM main loop selected by LD_GEMM_TILE on the DiT shapes."""
import sys, os, torch
sys.path.insert(0, ".")
from landiff_amd import ops
dev = "cuda"
def rel(a, b): return ((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-6)).item()
torch.manual_seed(0)
for (M, N, K) in [(256, 256, 128), (300, 200, 256), (1000, 1920, 1920), (513, 5760, 384), (4444, 1920, 7680)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    a[:, 0] += torch.arange(M, device=dev).to(torch.bfloat16) * 0.01
    ref = a.float() @ w.float().t()
    worst = 0.0
    for rep in range(5):
        out = ops.gemm(a, w)
        worst = max(worst, rel(out, ref))
    print(f"M={M} N={N} K={K}: rel err {worst:.2e}", flush=True)
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
M, D = 35552, 1920
def rnd(*s, sc=1.0): return (torch.randn(*s, device=dev) * sc).to(torch.bfloat16)
x = rnd(M, D); x4 = rnd(M, 4 * D)
for name, a, w in [("qkv", x, rnd(3 * D, D, sc=0.02)), ("proj", x, rnd(D, D, sc=0.02)), ("ff1", x, rnd(4 * D, D, sc=0.02)), ("ff2", x4, rnd(D, 4 * D, sc=0.02))]:
    N, K = w.shape
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bias = rnd(N)
    ms = timeit(lambda: ops.gemm(a, w, out=out, bias=bias))
    print(f"{name} N={N} K={K}: {ms:.3f} ms {2*M*N*K/ms/1e9:.0f} TF", flush=True)

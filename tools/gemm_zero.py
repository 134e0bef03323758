"""DVFS probe: the DiT GEMM shapes on zero-filled vs random operands (same kernels, same instruction stream)."""
import sys, os, torch
sys.path.insert(0, ".")
from landiff_amd import ops
dev = "cuda"
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
M, D = 35552, 1920
for fill in ("randn", "zeros", "const"):
    def mk(*s, sc=1.0):
        if fill == "randn": return (torch.randn(*s, device=dev) * sc).to(torch.bfloat16)
        if fill == "zeros": return torch.zeros(*s, device=dev, dtype=torch.bfloat16)
        return torch.full(s, 0.5 * sc, device=dev, dtype=torch.bfloat16)
    res = []
    for name, (N, K) in (("qkv", (3 * D, D)), ("ff1", (4 * D, D)), ("ff2", (D, 4 * D))):
        a = mk(M, K); w = mk(N, K, sc=0.02); out = torch.empty(M, N, device=dev, dtype=torch.bfloat16); bias = mk(N)
        ms = timeit(lambda: ops.gemm(a, w, out=out, bias=bias))
        res.append(f"{name} {ms:.3f}ms {2*M*N*K/ms/1e9:.0f}TF")
    print(f"TILE={os.environ.get('LD_GEMM_TILE')} fill={fill}: " + " | ".join(res), flush=True)

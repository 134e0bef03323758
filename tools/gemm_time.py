"""Times the four DiT GEMM shapes (bias epilogue) for the main loop selected by LD_GEMM_TILE / LD_GEMM_DBG."""
import sys, os, torch
sys.path.insert(0, ".")
from landiff_amd import ops
dev = "cuda"
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
res = []
for name, a, w in [("qkv", x, rnd(3 * D, D, sc=0.02)), ("proj", x, rnd(D, D, sc=0.02)), ("ff1", x, rnd(4 * D, D, sc=0.02)), ("ff2", x4, rnd(D, 4 * D, sc=0.02))]:
    N, K = w.shape
    if os.environ.get("SHAPES") and name not in os.environ["SHAPES"]: continue
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bias = rnd(N)
    ms = timeit(lambda: ops.gemm(a, w, out=out, bias=bias))
    res.append(f"{name} {ms:.3f}ms {2*M*N*K/ms/1e9:.0f}TF")
print(f"TILE={os.environ.get('LD_GEMM_TILE')} DBG={os.environ.get('LD_GEMM_DBG')}: " + " | ".join(res), flush=True)

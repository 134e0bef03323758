"""Per-tile fixed cost of the GEMM kernels: time at K and 2K (same M, N) -> intercept = prologue + epilogue share."""
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
M, N = 35552, 7680
for fill in ("randn", "zeros"):
    ts = {}
    for K in (1920, 3840, 7680):
        a = (torch.randn(M, K, device=dev) if fill == "randn" else torch.zeros(M, K, device=dev)).to(torch.bfloat16)
        w = ((torch.randn(N, K, device=dev) * 0.02) if fill == "randn" else torch.zeros(N, K, device=dev)).to(torch.bfloat16)
        bias = torch.zeros(N, device=dev, dtype=torch.bfloat16); out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ts[K] = timeit(lambda: ops.gemm(a, w, out=out, bias=bias))
        tl = timeit(lambda: torch.addmm(bias, a, w.t(), out=out))
        print(f"TILE={os.environ.get('LD_GEMM_TILE')} {fill} K={K}: own {ts[K]:.3f} ms ({2*M*N*K/ts[K]/1e9:.0f} TF) | lib {tl:.3f} ms ({2*M*N*K/tl/1e9:.0f} TF)", flush=True)
    fixed = 2 * ts[1920] - ts[3840]
    print(f"   fixed cost at K=1920: {fixed:.3f} ms = {100*fixed/ts[1920]:.0f} % ; slope-rate {2*M*N*1920/(ts[3840]-ts[1920])/1e9:.0f} TF")

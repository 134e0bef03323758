import sys, torch
sys.path.insert(0, ".")
from landiff_amd import ops
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
M, N = 35552, 5760
import os
for K in [int(k) for k in os.environ.get('KS', '64,128,640,1920').split(',')]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ms = timeit(lambda: ops.gemm(a, w, out=out))
    print(f"K={K}: {ms:.3f} ms {2*M*N*K/ms/1e9:.0f} TF")

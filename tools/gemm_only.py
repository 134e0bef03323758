import sys, os, torch
sys.path.insert(0, ".")
from landiff_amd import ops
M, N, K = 35552, 7680, int(os.environ.get("K", "1920"))
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.gemm(a, w, out=out)
torch.cuda.synchronize()

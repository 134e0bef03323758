"""One DiT-shape GEMM (SHAPE=qkv|proj|ff1|ff2) launched a few times -- target for rocprofv3 --pmc runs."""
import sys, os, torch
sys.path.insert(0, ".")
from landiff_amd import ops
M, D = 35552, 1920
shapes = {"qkv": (3 * D, D), "proj": (D, D), "ff1": (4 * D, D), "ff2": (D, 4 * D)}
N, K = shapes[os.environ.get("SHAPE", "ff1")]
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(int(os.environ.get("IT", "3"))):
    ops.gemm(a, w, out=out)
torch.cuda.synchronize()

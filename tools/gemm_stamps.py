"""Per-segment cycle sums (s_memtime) of the ping-pong GEMM main loop: LD_GEMM_TILE=7 LD_GEMM_DBG=72|73|74|75."""
import sys, os, torch
sys.path.insert(0, ".")
from landiff_amd import ops
M, D = 35552, 1920
shapes = {"qkv": (3 * D, D), "proj": (D, D), "ff1": (4 * D, D), "ff2": (D, 4 * D)}
N, K = shapes[os.environ.get("SHAPE", "ff1")]
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(3): ops.gemm(a, w, out=out)
torch.cuda.synchronize()
st = out.view(-1)[:8 * 4 * 4].view(torch.int64).cpu().reshape(8, 4)
nph = K // 16 - 1
print(f"DBG={os.environ.get('LD_GEMM_DBG')} K={K} phases={nph}: per-phase cycles [load, barrier1, mfma, barrier2] per wave")
for w_ in range(8): print("  wave", w_, [round(x / nph) for x in st[w_].tolist()], "sum", round(st[w_].sum().item() / nph))

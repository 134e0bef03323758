"""rocprofv3 target: the four DiT GEMM shapes on MXFP8 operands (bias-only epilogues), three launches each."""
import sys, torch
sys.path.insert(0, ".")
from landiff_amd import ops
M, D = 35552, 1920
rnd = lambda *s, sc=1.0: (torch.randn(*s, device="cuda") * sc).to(torch.bfloat16)
x, x4 = rnd(M, D), rnd(M, 4 * D)
for name, a, N, K in (("qkv", x, 3 * D, D), ("proj", x, D, D), ("ff1", x, 4 * D, D), ("ff2", x4, D, 4 * D)):
    w = rnd(N, K, sc=0.02); b = rnd(N); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    a8, sa = ops.quantize_mxfp8(a); w8, sw = ops.quantize_mxfp8(w)
    for _ in range(3): ops.gemm_mxfp8(a8, sa, w8, sw, out=out, bias=b)
torch.cuda.synchronize()

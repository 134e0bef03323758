import sys, torch
sys.path.insert(0, ".")
from landiff_amd import ops
BF=torch.bfloat16
g = torch.Generator().manual_seed(4)
M, N, K = 4096, 1920, 1920
for outl in (1.0, 30.0, 300.0):
    a = torch.randn(M, K, generator=g).to("cuda", BF)
    a[:, ::97] *= outl
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to("cuda", BF)
    out16 = ops.gemm(a, w).float()
    a8, sa = ops.quantize_mxfp8(a); w8, sw = ops.quantize_mxfp8(w)
    rel_mx = ((ops.gemm_mxfp8(a8, sa, w8, sw).float() - out16).norm() / out16.norm()).item()
    q8, s8 = ops.quantize_fp8(a); v8, t8 = ops.quantize_fp8(w)
    rel_row = ((ops.gemm_fp8(q8, s8, v8, t8).float() - out16).norm() / out16.norm()).item()
    print(f"outlier x{outl}: MX {rel_mx:.4f}   per-row {rel_row:.4f}")

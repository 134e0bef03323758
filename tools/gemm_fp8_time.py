"""DiT linear shapes: bf16 GEMM vs quantise + fp8 GEMM (random data)."""
import sys, torch
sys.path.insert(0, ".")
from landiff_amd import ops
dev = "cuda"
M = 35552
def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for name, N, K in (("qkv", 5760, 1920), ("proj", 1920, 1920), ("fc1", 7680, 1920), ("fc2", 1920, 7680)):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    w8, sw = ops.quantize_fp8(w)
    a8, sa = ops.quantize_fp8(a)
    fl = 2.0 * M * N * K
    t16 = t(lambda: ops.gemm(a, w, out=out, bias=bias))
    t8 = t(lambda: ops.gemm_fp8(a8, sa, w8, sw, out=out, bias=bias))
    tq = t(lambda: ops.quantize_fp8(a, a8, sa))
    m8, ms = ops.quantize_mxfp8(a); v8, vs = ops.quantize_mxfp8(w)
    tm = t(lambda: ops.gemm_mxfp8(m8, ms, v8, vs, out=out, bias=bias))
    tmq = t(lambda: ops.quantize_mxfp8(a, m8, ms))
    print(f"{name:5s} M={M} N={N} K={K}: bf16 {t16*1e3:7.1f} us ({fl/t16/1e9:6.0f} TF)   fp8 row {t8*1e3:7.1f} us ({fl/t8/1e9:6.0f} TF) + quantise {tq*1e3:6.1f} us"
          f"   MXFP8 {tm*1e3:7.1f} us ({fl/tm/1e9:6.0f} TF) + quantise {tmq*1e3:6.1f} us")

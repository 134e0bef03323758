"""Calibration: the DiT GEMM shapes through torch.matmul (hipBLASLt / rocBLAS on ROCm) on the same random operands as
tools/gemm_dit_shapes.py.  Not used by the product; answers "is ~1.0-1.1 PFLOP/s the practical ceiling on this data?"."""
import sys, torch
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
for fill in ("randn", "zeros"):
    def mk(*s, sc=1.0):
        return (torch.randn(*s, device=dev) * sc).to(torch.bfloat16) if fill == "randn" else torch.zeros(*s, device=dev, dtype=torch.bfloat16)
    for name, (N, K) in (("qkv", (3 * D, D)), ("proj", (D, D)), ("ff1", (4 * D, D)), ("ff2", (D, 4 * D))):
        a = mk(M, K); w = mk(N, K, sc=0.02); bias = mk(N)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t_lib = timeit(lambda: torch.addmm(bias, a, w.t(), out=out))
        t_own = timeit(lambda: ops.gemm(a, w, out=out, bias=bias))
        f = 2 * M * N * K / 1e9
        print(f"{fill:5s} {name:4s} N={N} K={K}: torch.addmm {t_lib:.3f} ms {f/t_lib:.0f} TF | ld_gemm_bf16 {t_own:.3f} ms {f/t_own:.0f} TF", flush=True)

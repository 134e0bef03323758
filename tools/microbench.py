"""Kernel micro-benchmarks at the full LanDiff shapes (run on the GPU box)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from landiff_amd import ops


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def bench_gemm(M, N, K):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ms = timeit(lambda: ops.gemm(a, w, out=out))
    ms_t = timeit(lambda: torch.matmul(a, w.t()))
    print(f"gemm M={M} N={N} K={K}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.0f} TF | torch(hipblaslt) {ms_t:.3f} ms {2*M*N*K/ms_t/1e9:.0f} TF")


def bench_attn(B, H, N):
    Npad = (N + 127) // 128 * 128
    q = torch.randn(B, H, Npad, 64, device="cuda").to(torch.bfloat16)
    k = torch.randn(B, H, Npad, 64, device="cuda").to(torch.bfloat16)
    vt = torch.randn(B, H, 64, Npad, device="cuda").to(torch.bfloat16)
    out = torch.empty(B, N, H * 64, device="cuda", dtype=torch.bfloat16)
    ms = timeit(lambda: ops.attn_fwd(q, k, vt, out, N, N, 0.125), iters=5)
    fl = 4.0 * B * H * N * N * 64
    qq, kk, vv = q[:, :, :N], k[:, :, :N], vt.transpose(2, 3)[:, :, :N].contiguous()
    try:
        ms_t = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(qq, kk, vv), iters=3)
    except Exception as e:  # noqa
        ms_t = float("nan")
    print(f"attn B={B} H={H} N={N}: {ms:.3f} ms {fl/ms/1e9:.0f} TF | torch sdpa {ms_t:.3f} ms {fl/ms_t/1e9:.0f} TF")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "gemm"):
        M = 2 * 17776
        bench_gemm(M, 5760, 1920)
        bench_gemm(M, 1920, 1920)
        bench_gemm(M, 7680, 1920)
        bench_gemm(M, 1920, 7680)
        bench_gemm(4096, 4096, 4096)
        bench_gemm(8192, 8192, 8192)
    if which in ("all", "attn"):
        bench_attn(2, 30, 17776)
        bench_attn(1, 12, 18768)

"""Epilogue / tile-overhead probe: the DiT GEMM shapes at K = 64 (one K-tile: launch + prologue + epilogue only) and at full K."""
import sys, torch
sys.path.insert(0, ".")
from landiff_amd import ops
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
M, D = 35552, 1920
dev = "cuda"
def rnd(*s, sc=1.0): return (torch.randn(*s, device=dev) * sc).to(torch.bfloat16)
gate = rnd(2, 12 * D); resid = rnd(M, D)
for name, N, epi in (("ff1", 4 * D, dict(act="gelu_tanh")), ("plain7680", 4 * D, dict()), ("proj/ff2", D, dict(resid=resid, gate=gate, gate_bstride=12 * D, gate_off_img=2 * D, gate_off_txt=8 * D, rows_per_batch=M // 2, text_len=226)), ("plain1920", D, dict())):
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bias = rnd(N)
    for K in (64, 128, 1920):
        a, w = rnd(M, K), rnd(N, K, sc=0.02)
        e = dict(epi)
        if name != "plain7680" and name != "plain1920": e["bias"] = bias
        ms = timeit(lambda: ops.gemm(a, w, out=out, **e))
        print(f"{name:10s} N={N} K={K:5d}: {ms*1e3:8.1f} us  ({2*M*N*K/ms/1e9:6.0f} TF)  out {M*N*2/1e6:.0f} MB -> {M*N*2/ms/1e6:.0f} GB/s", flush=True)

"""Times the four DiT layer GEMMs at full size with their real epilogues (and with none) -- epilogue cost probe."""
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
M, D = 35552, 1920
dev = "cuda"
def rnd(*s, sc=1.0): return (torch.randn(*s, device=dev) * sc).to(torch.bfloat16)
x = rnd(M, D); x4 = rnd(M, 4 * D)
gate = rnd(2, 12 * D)
resid = rnd(M, D)
cases = [
    ("qkv  N=5760 K=1920", x, rnd(3 * D, D, sc=0.02), [dict(), dict(bias=rnd(3 * D))]),
    ("proj N=1920 K=1920", x, rnd(D, D, sc=0.02), [dict(), dict(bias=rnd(D)), dict(bias=rnd(D), resid=resid, gate=gate, gate_bstride=12 * D, gate_off_img=2 * D, gate_off_txt=8 * D, rows_per_batch=M // 2, text_len=226)]),
    ("ff1  N=7680 K=1920", x, rnd(4 * D, D, sc=0.02), [dict(), dict(bias=rnd(4 * D)), dict(bias=rnd(4 * D), act="gelu_tanh")]),
    ("ff2  N=1920 K=7680", x4, rnd(D, 4 * D, sc=0.02), [dict(), dict(bias=rnd(D), resid=resid, gate=gate, gate_bstride=12 * D, gate_off_img=5 * D, gate_off_txt=11 * D, rows_per_batch=M // 2, text_len=226)]),
]
for name, a, w, epis in cases:
    N, K = w.shape
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for e in epis:
        ms = timeit(lambda: ops.gemm(a, w, out=out, **e))
        print(f"{name} epi={sorted(e.keys())}: {ms:.3f} ms {2*M*N*K/ms/1e9:.0f} TF", flush=True)

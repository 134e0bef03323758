"""Does the Infinity Cache (MALL, 256 MB) serve a weight matrix faster than HBM?  The decode GEMVs on a matrix that was read
just before (same matrix every launch) against a rotation over enough copies to exceed the cache."""
import sys, torch
sys.path.insert(0, ".")
from landiff_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16

def timeit(fn, iters=40, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

for name, N, K, gated in (("qkv", 6144, 2048, False), ("wo", 2048, 2048, False), ("w1w3", 11008, 2048, True), ("w2", 2048, 11008, False)):
    byts = N * K * 2 * (2 if gated else 1)
    ncopy = max(2, int(600e6 // byts) + 1)                  # > 2 x the cache in flight between two uses of a copy
    ws = [torch.randn(N, K, device=dev).to(BF) for _ in range(ncopy)]
    w2s = [torch.randn(N, K, device=dev).to(BF) for _ in range(ncopy)] if gated else None
    x = torch.randn(2, K, device=dev).to(BF)
    out = torch.empty(2, N, device=dev, dtype=BF)
    for mode in ("rotate", "same"):
        it = [0]
        def f():
            i = (it[0] % ncopy) if mode == "rotate" else 0
            it[0] += 1
            ops.gemv(x, ws[i], out, w2=w2s[i] if gated else None, act="gelu_tanh" if gated else None)
        ms = timeit(f)
        print(f"gemv {name:5s} {byts/1e6:5.1f} MB {mode:6s}: {ms*1e3:6.1f} us  {byts/ms/1e9:5.2f} TB/s")

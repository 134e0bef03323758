"""Reference points for the GEMV work: achievable HBM read bandwidth (torch reduction over 4 GB; and over 24 x 90 MB
separate buffers inside one HIP graph, i.e. the same launch granularity as the LLM's gated GEMV)."""
import torch
dev = "cuda"
big = torch.ones(2 * 1024 ** 3, device=dev, dtype=torch.bfloat16)      # 4 GiB
def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
ms = t(lambda: big.view(torch.int32).sum())
print(f"torch int32 sum over 4 GiB: {ms:.3f} ms, {big.numel() * 2 / ms / 1e9:.2f} TB/s")
ms = t(lambda: big.view(torch.float32).max())
print(f"torch fp32 max over 4 GiB: {ms:.3f} ms, {big.numel() * 2 / ms / 1e9:.2f} TB/s")
for mb in (8, 25, 45, 90):
    n = mb * 1024 * 1024 // 4
    bufs = [big.view(torch.int32)[i * n:(i + 1) * n] for i in range(24)]
    outs = torch.zeros(24, device=dev, dtype=torch.int64)
    def run():
        for i in range(24): torch.sum(bufs[i], dim=0, out=outs[i])
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s): run()
    ms = t(g.replay)
    print(f"24 x {mb} MiB separate sums in a graph: {ms / 24 * 1e3:.2f} us each, {mb * 1.048576e6 / (ms / 24 * 1e-3) / 1e12:.2f} TB/s")

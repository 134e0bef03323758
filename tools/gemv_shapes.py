"""LLM decode GEMV shapes (B=2) over 24 distinct weight sets (no cache reuse), wall-timed inside a HIP graph.
SHAPE=qkv|wo|gated|w2|all"""
import sys, os, torch
sys.path.insert(0, ".")
from landiff_amd import ops
dev = "cuda"
H, F, L = 2048, 11008, 24
def rnd(*s): return (torch.randn(*s, device=dev) * 0.02).to(torch.bfloat16)
which = os.environ.get("SHAPE", "all")
x = rnd(2, H); xf = rnd(2, F); nw = torch.ones(H, device=dev)
qkv = torch.empty(2, 3 * H, device=dev, dtype=torch.bfloat16); o = torch.empty(2, H, device=dev, dtype=torch.bfloat16)
g = torch.empty(2, F, device=dev, dtype=torch.bfloat16)
Ws = {"qkv": [rnd(3 * H, H) for _ in range(L)], "wo": [rnd(H, H) for _ in range(L)],
      "w1": [rnd(F, H) for _ in range(L)], "w3": [rnd(F, H) for _ in range(L)], "w2": [rnd(H, F) for _ in range(L)]}
def run(shape):
    for i in range(L):
        if shape in ("qkv", "all"): ops.gemv(x, Ws["qkv"][i], qkv, norm_w=nw, norm_eps=1e-5)
        if shape in ("wo", "all"): ops.gemv(x, Ws["wo"][i], o, resid=o)
        if shape in ("gated", "all"): ops.gemv(x, Ws["w1"][i], g, w2=Ws["w3"][i], act="gelu_tanh", norm_w=nw, norm_eps=1e-5)
        if shape in ("w2", "all"): ops.gemv(xf, Ws["w2"][i], o, resid=o)
mb = {"qkv": 3 * H * H * 2, "wo": H * H * 2, "gated": 2 * F * H * 2, "w2": F * H * 2}
mb["all"] = sum(mb.values())
for shape in ([which] if which != "each" else ["qkv", "wo", "gated", "w2", "all"]):
    run(shape); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run(shape)
        torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=s):
            run(shape)
    torch.cuda.synchronize()
    for _ in range(3): gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 20
    e0.record()
    for _ in range(it): gr.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / it / L * 1e3
    print(f"{shape:6s}: {us:7.2f} us per layer-op, {mb[shape] / us / 1e6:6.2f} TB/s", flush=True)

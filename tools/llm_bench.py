"""LLM decode micro-benchmark: GEMV bandwidth per shape and the full 1244-step decode."""
import sys, time
import torch
sys.path.insert(0, ".")
from landiff_amd import ops
from landiff_amd.config import LLMConfig
from landiff_amd.llm import LLMRunner
from landiff_amd.weights import init_state, llm_spec


def timeit(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

dev = torch.device("cuda:0")
BF = torch.bfloat16
for name, N, K, gated in (("qkv", 6144, 2048, False), ("wo", 2048, 2048, False), ("w1w3", 11008, 2048, True), ("w2", 2048, 11008, False)):
    # rotate over 8 weight copies so the matrix is not L2/MALL resident between launches
    ws = [torch.randn(N, K, device=dev).to(BF) for _ in range(8)]
    w2s = [torch.randn(N, K, device=dev).to(BF) for _ in range(8)] if gated else None
    x = torch.randn(2, K, device=dev).to(BF)
    out = torch.empty(2, N, device=dev, dtype=BF)
    it = [0]
    def f():
        i = it[0] % 8; it[0] += 1
        ops.gemv(x, ws[i], out, w2=w2s[i] if gated else None, act="gelu_tanh" if gated else None)
    ms = timeit(f)
    byts = N * K * 2 * (2 if gated else 1)
    print(f"gemv {name} N={N} K={K}: {ms*1e3:.1f} us  {byts/ms/1e9:.2f} TB/s")

cfg = LLMConfig()
sd = init_state(llm_spec(cfg), 1, dtype=BF, device=dev)
run = LLMRunner(sd, cfg, dev)
text = torch.randn(64, cfg.text_dim, device=dev)
for use_graph in (True, False, True, False):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    toks = run.sample(text, guidance_scale=7.5, seed=42, use_graph=use_graph)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"decode 1244 steps graph={use_graph}: {dt:.3f} s  ({dt/1244*1e3:.3f} ms/step), host enqueue {run.host_enqueue_s:.3f} s")

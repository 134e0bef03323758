"""Does the AR decode give the same tokens when other kernels share the GPU?  Full-size LLMRunner; the decode runs in a helper thread on
its own stream (PRIO=-1|0) while the main thread keeps the chip busy with DiT-sized GEMMs / attention (LOAD=gemm|attn|both|none).
Prints the first step at which the CFG logits of two runs differ, how much, and the token agreement."""
import os, sys, threading, torch
sys.path.insert(0, ".")
from landiff_amd import ops
from landiff_amd.config import LLMConfig
from landiff_amd.llm import LLMRunner
from landiff_amd.weights import init_state, llm_spec
dev = torch.device("cuda:0")
cfg = LLMConfig()
run = LLMRunner(init_state(llm_spec(cfg), 1, dtype=torch.bfloat16, device=dev), cfg, dev, max_frames=26)
text = torch.randn(64, cfg.text_dim, device=dev)
prio = int(os.environ.get("PRIO", "-1")); load = os.environ.get("LOAD", "both"); frames = int(os.environ.get("FRAMES", "4"))
M, D = 35552, 1920
rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev) * sc).to(torch.bfloat16)
x, w, b = rnd(M, D), rnd(4 * D, D, sc=0.02), rnd(4 * D)
o = torch.empty(M, 4 * D, device=dev, dtype=torch.bfloat16)
B, H, N = 2, 30, 17776; Npad = (N + 127) // 128 * 128
q = rnd(B, H, Npad, 64); k = rnd(B, H, Npad, 64); vt = rnd(B, H, 64, Npad); ao = torch.empty(B, N, H * 64, device=dev, dtype=torch.bfloat16)
side = torch.cuda.Stream(device=dev, priority=prio)
def decode(with_load):
    log, res, stop = [], {}, threading.Event()
    def work():
        torch.cuda.set_device(dev)
        with torch.cuda.stream(side):
            res["ids"] = run.sample(text, guidance_scale=7.5, seed=42, num_frames=frames, logits_log=log, mode="chain").clone()
            side.synchronize()
        stop.set()
    side.wait_stream(torch.cuda.current_stream(dev))
    th = threading.Thread(target=work); th.start()
    n = 0
    while with_load and not stop.is_set():
        if load in ("gemm", "both"): ops.gemm(x, w, out=o, bias=b, act="gelu_tanh")
        if load in ("attn", "both"): ops.attn_fwd(q, k, vt, ao, N, N, 0.125)
        n += 1
        if n % 8 == 0: torch.cuda.current_stream(dev).synchronize()
    th.join(); torch.cuda.synchronize()
    return res["ids"], torch.cat(log, 0)
ops.attn_fwd(q, k, vt, ao, N, N, 0.125); torch.cuda.synchronize()
ao_ref = ao.clone()
ref_ids, ref_log = decode(False)
ids2, log2 = decode(False)
print("quiet vs quiet:", "equal" if torch.equal(ref_ids, ids2) and torch.equal(ref_log, log2) else "DIFFER", flush=True)
for i in range(3):
    ids, log = decode(True)
    d = (log - ref_log).abs().amax(dim=1)
    nz = torch.nonzero(d > 0)
    first = int(nz[0]) if len(nz) else -1
    if load in ("attn", "both"):
        print("   attention output under the decode:", "equal" if torch.equal(ao, ao_ref) else f"DIFFERS ({(ao != ao_ref).float().mean().item():.2e} of elements)")
    print(f"loaded run {i} (LOAD={load}, PRIO={prio}) vs quiet: tokens equal {torch.equal(ids, ref_ids)} ({(ids == ref_ids).float().mean().item():.4f}); "
          f"first differing step {first} of {log.shape[0]}" + (f", |dlogit| there {d[first].item():.4g} (|logit|max {ref_log[first].abs().max().item():.3g}), "
          f"steps that differ {int((d > 0).sum())}" if first >= 0 else ""), flush=True)

"""The persistent one-launch decode step (ld_llm_fused.hip) against the per-operation chain at the full 24 x 2048 size:
token ids, final logits and the KV cache must be bit-identical; wall time of the 1244-step decode for both forms.
usage: python tools/llm_fused_check.py [num_frames]   (run under `timeout`: a wrong grid barrier would otherwise wait ~1 s per step)"""
import sys, time
import torch
sys.path.insert(0, ".")
from landiff_amd.config import LLMConfig
from landiff_amd.llm import LLMRunner
from landiff_amd.weights import init_state, llm_spec

dev = torch.device("cuda:0")
BF = torch.bfloat16
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 13
cfg = LLMConfig()
sd = init_state(llm_spec(cfg), 1, dtype=BF, device=dev)
run = LLMRunner(sd, cfg, dev)
assert run.fused_supported
text = torch.randn(64, cfg.text_dim, device=dev)
res = {}
for fused in (False, True, False, True):
    logs = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    toks = run.sample(text, guidance_scale=7.5, seed=42, num_frames=frames, fused=fused).clone()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    n = toks.numel()
    state = (toks, run.logits.clone(), run.kc[0].clone(), run.vc[-1].clone(), run.x.clone())
    print(f"fused={fused}: {dt:.3f} s for {n} visual tokens, host enqueue {run.host_enqueue_s:.3f} s, ctl {run.fused_ctl[:2].tolist()}", flush=True)
    if fused in res:
        continue
    res[fused] = state
a, b = res[False], res[True]
names = ("tokens", "logits", "k_cache[0]", "v_cache[-1]", "x")
ok = True
for nm, u, v in zip(names, a, b):
    same = torch.equal(u, v)
    ok &= same
    print(f"{nm}: {'identical' if same else 'DIFFERENT'}" + ("" if same else f" ({(u != v).sum().item()} of {u.numel()} elements)"))
print("RESULT", "bit-identical" if ok else "MISMATCH")
sys.exit(0 if ok else 1)

"""The dependent-launch form (two streams) and the persistent one-launch form (ld_llm_fused.hip) of a decode step against the
per-operation chain at the full 24 x 2048 size: token ids, final logits and the KV cache must be bit-identical; wall time of the
1244-step decode for every form.
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
assert run.fused_supported and run.chained_supported
text = torch.randn(64, cfg.text_dim, device=dev)
res = {}
for mode in ("chain", "chained", "fused", "chain", "chained", "fused"):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    toks = run.sample(text, guidance_scale=7.5, seed=42, num_frames=frames, mode=mode).clone()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    n = toks.numel()
    state = (toks, run.logits.clone(), run.kc[0].clone(), run.vc[-1].clone())
    print(f"{mode:8s}: {dt:.3f} s for {n} visual tokens, host enqueue {run.host_enqueue_s:.3f} s, "
          f"errors fused {int(run.fused_ctl[1])} chained {int(run.chain_ctl[0])}", flush=True)
    res.setdefault(mode, state)
names = ("tokens", "logits", "k_cache[0]", "v_cache[-1]")
ok = True
for mode in ("chained", "fused"):
    for nm, u, v in zip(names, res["chain"], res[mode]):
        same = torch.equal(u, v)
        ok &= same
        print(f"{mode} {nm}: {'identical' if same else 'DIFFERENT'}" + ("" if same else f" ({(u != v).sum().item()} of {u.numel()} elements)"))
print("RESULT", "bit-identical" if ok else "MISMATCH")
sys.exit(0 if ok else 1)

"""One decode step: per-operation chain vs the persistent launch, layer by layer and buffer by buffer (debug aid)."""
import sys
import torch
sys.path.insert(0, ".")
from landiff_amd import ops
from landiff_amd.config import LLMConfig
from landiff_amd.llm import LLMRunner
from landiff_amd.weights import init_state, llm_spec

dev = torch.device("cuda:0")
cfg = LLMConfig()
sd = init_state(llm_spec(cfg), 1, dtype=torch.bfloat16, device=dev)
run = LLMRunner(sd, cfg, dev)
text = torch.randn(64, cfg.text_dim, device=dev)
run.sample(text, guidance_scale=7.5, seed=42, num_frames=1, mode="chain")       # leaves a filled KV cache and a position
c = cfg
pos0 = run.pos.clone()
kc0 = [k.clone() for k in run.kc]; vc0 = [v.clone() for v in run.vc]
NAMES = ("qkv", "ws", "att", "gate", "x", "kc", "vc")


def step(layers, x_in, fused):
    tab = ops.llm_layer_table([run.blocks[i] for i in layers], [run.kc[i] for i in layers], [run.vc[i] for i in layers])
    run.x.copy_(x_in); run.pos.copy_(pos0)
    for i in layers:
        run.kc[i].copy_(kc0[i]); run.vc[i].copy_(vc0[i])
    run.fused_ctl.zero_()
    args = (None, run.token, run.pos, run.x, run.qkv, run.att, run.gate, run.attn_ws, run.cos, run.sin, run.ln_w, run.ln_b,
            run.lnf, run.head, run.logits, c.heads, run.Lmax, run.nsplit, c.rms_eps, c.ln_eps)
    if fused:
        ops.llm_decode_forward_fused(ops.llm_layer_table_device(tab, dev), len(layers), *args, run.fused_ctl)
    else:
        ops.llm_decode_forward(tab, *args)
    torch.cuda.synchronize()
    assert run.fused_ctl[1].item() == 0
    l = layers[-1]
    return dict(qkv=run.qkv.clone(), ws=run.attn_ws.clone(), att=run.att.clone(), gate=run.gate.clone(), x=run.x.clone(),
                kc=run.kc[l].clone(), vc=run.vc[l].clone())


found = 0
for seed in range(40):
    g = torch.Generator(device=dev); g.manual_seed(1000 + seed)
    x0 = torch.randn(2, c.hidden, device=dev, generator=g).to(torch.bfloat16)
    a, b = step(list(range(24)), x0, False), step(list(range(24)), x0, True)
    if all(torch.equal(a[k], b[k]) for k in NAMES):
        continue
    found += 1
    print(f"seed {seed}: 24-layer step differs; walking the layers on the chain's activations")
    x = x0
    for l in range(24):
        a, b = step([l], x, False), step([l], x, True)
        bad = [k for k in NAMES if not torch.equal(a[k], b[k])]
        if bad:
            print(f"  layer {l}: first difference in {bad}")
            for k in bad:
                u, v = a[k].reshape(-1), b[k].reshape(-1)
                idx = (u != v).nonzero().reshape(-1)
                print(f"    {k}: {idx.numel()} of {u.numel()} differ, idx {idx[:8].tolist()} chain {u[idx[:4]].tolist()} fused {v[idx[:4]].tolist()}")
            break
        x = a["x"]
    else:
        print("  every single layer matches on its own: the difference needs the layers in one launch")
    if found >= 3:
        break
print("seeds with a difference:", found)

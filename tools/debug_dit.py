import sys; sys.path.insert(0, ".")
import torch
from landiff_amd.config import PipelineConfig
from landiff_amd.weights import init_pipeline_state
from landiff_amd.dit import ControlDiTRunner
from landiff_amd import ops
from oracle.dit import ControlDiTOracle
import torch.nn.functional as F

def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-6)).item()

cuda = torch.device("cuda:0")
cfg = PipelineConfig.tiny(3)
st = init_pipeline_state(cfg, 1234)
d = cfg.dit
g = torch.Generator().manual_seed(0)
x = torch.randn(1, d.latent_frames, d.in_channels, d.latent_h, d.latent_w, generator=g)
ctx = torch.randn(1, d.text_len, d.text_dim, generator=g)
sem = (torch.randn(1, d.latent_frames, d.in_channels, d.latent_h, d.latent_w, generator=g) * 0.5).to(torch.bfloat16)
orc = ControlDiTOracle(st["dit_main"], st["dit_control"], d, torch.bfloat16)
run = ControlDiTRunner(st["dit_main"], st["dit_control"], d, cuda)
run.set_condition(ctx, sem[0])
t = 979.0
x2 = torch.cat([x, x]); ctx2 = torch.cat([torch.zeros_like(ctx), ctx])
ts = torch.full((2,), t)
# oracle control
oc = orc.ctrl
xc = x2.to(torch.bfloat16) + sem
emb = oc.time_emb(ts)
h = oc.embed(xc, ctx2.to(torch.bfloat16))
run._time_emb(run.ctrl, t)
print("emb", rel(run.emb, emb))
run._embed(run.ctrl, x.to(cuda), run.hc, run.txt_ctrl, run.sem)
print("embed", rel(run.hc.view(2, -1, d.hidden), h))
# layer 0 pieces
lw = run.ctrl.layers[0]
ada_ref = F.linear(F.silu(emb), oc.s["mixins.adaln_layer.adaLN_modulations.0.1.weight"].bfloat16(), oc.s["mixins.adaln_layer.adaLN_modulations.0.1.bias"].bfloat16())
ops.gemv(run.emb, lw["ada_w"], run.ada, bias=lw["ada_b"], in_act="silu")
print("ada", rel(run.ada, ada_ref))
h1 = oc.layer(0, h, emb)
run._layer(run.ctrl, 0, run.hc, run.hc)
ops.gemm(run.hc, lw["zero_w"], out=run.ctrl_out[0])
print("layer0+zero", rel(run.ctrl_out[0].view(2, -1, d.hidden), h1))
# finer: attention input
from oracle.dit import modulate
from oracle.common import layer_norm
# ---- full control + main, per-layer errors, and noise floor (bf16 oracle vs fp32 oracle)
orc32 = ControlDiTOracle(st["dit_main"], st["dit_control"], d, torch.float32)
run.set_condition(ctx, sem[0])
outs_bf = orc.ctrl.forward(x2, ts, ctx2, semantic_feature=sem)
outs_32 = orc32.ctrl.forward(x2, ts, ctx2, semantic_feature=sem.float())
run._time_emb(run.ctrl, t)
run._embed(run.ctrl, x.to(cuda), run.hc, run.txt_ctrl, run.sem)
h_in = run.hc
for i in range(d.layers_control):
    run._layer(run.ctrl, i, h_in, run.hc)
    ops.gemm(run.hc, run.ctrl.layers[i]["zero_w"], out=run.ctrl_out[i])
    h_in = run.ctrl_out[i]
    print(f"ctrl layer {i}: gpu-vs-bf16 {rel(run.ctrl_out[i].view(2,-1,d.hidden), outs_bf[i]):.4f}  bf16-vs-fp32 {rel(outs_bf[i], outs_32[i]):.4f} gpu-vs-fp32 {rel(run.ctrl_out[i].view(2,-1,d.hidden), outs_32[i]):.4f}")
eps_bf = orc.main.forward(x2, ts, ctx2, control_outputs=outs_bf)
eps_32 = orc32.main.forward(x2, ts, ctx2, control_outputs=outs_32)
out = torch.empty(1, d.latent_frames, d.in_channels, d.latent_h, d.latent_w, device=cuda)
run.step(x.to(cuda), t, -1.0, 0.0, 1.0, out)   # c_out=-1, c_skip=0, scale=1 -> out = -eps_cond
print("eps cond: gpu-vs-bf16", rel(-out[0], eps_bf[1]), "bf16-vs-fp32", rel(eps_bf[1], eps_32[1]), "gpu-vs-fp32", rel(-out[0], eps_32[1]))

"""The attention policy of ControlDiTRunner on a checkpoint whose logits leave the fast pass's window in SOME layers: full-size denoiser
steps (15 control + 30 main layers, B = 2) with the QK-LayerNorm gains of every third layer scaled by 5 (row maxima of q.k/8 ~ 100),
for attn_exact = False (in-kernel fallback) / "auto" (per-layer switch to ld_attn_fwd_bf16_exact) / True (every layer exact), and the
benign checkpoint under False / "auto" (what the bookkeeping costs when nothing leaves the window)."""
import sys, time, torch
sys.path.insert(0, ".")
from landiff_amd.config import PipelineConfig
from landiff_amd.dit import ControlDiTRunner
from landiff_amd.weights import init_pipeline_state
dev = torch.device("cuda:0")
cfg = PipelineConfig.full().check()
d = cfg.dit
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(1, d.latent_frames, d.in_channels, d.latent_h, d.latent_w, device=dev, generator=g)
ctx = torch.randn(1, d.text_len, d.text_dim, device=dev, generator=g)
sem = torch.randn(d.latent_frames, d.in_channels, d.latent_h, d.latent_w, device=dev, generator=g).to(torch.bfloat16)
out = torch.empty_like(x)
for label, hot in (("benign checkpoint", False), ("gains x 5 in every third layer", True)):
    st = init_pipeline_state(cfg, seed=1234, dtype=torch.bfloat16, device=dev, parts=["dit_main", "dit_control"])
    if hot:
        for part, L in (("dit_main", d.layers_main), ("dit_control", d.layers_control)):
            for i in range(0, L, 3):
                for nm in ("query", "key"):
                    st[part][f"mixins.adaln_layer.{nm}_layernorm_list.{i}.weight"] *= 5.0
    for mode in ((False, "auto", True) if hot else (False, "auto")):
        run = ControlDiTRunner(st["dit_main"], st["dit_control"], d, dev, attn_exact=mode)
        run.set_condition(ctx, sem)
        for i in range(3): run.step(x, 500 - i, -0.7, 0.7, 6.0, out)       # (auto: the switch happens at the end of step 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 4
        for i in range(n): run.step(x, 490 - i, -0.7, 0.7, 6.0, out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{label:32s} attn_exact={mode!s:5s}: {dt * 1e3:7.1f} ms per denoiser step   layers on the exact form: {len(run.exact_layers) if mode == 'auto' else ('all' if mode else 0)}", flush=True)
        del run
    del st

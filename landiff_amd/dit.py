"""Control + main diffusion transformer on MI355X (host orchestration over the HIP kernels).

Mirrors ControlDiffWarp.forward -> ControlDiffusionTransformer.forward / DiffusionTransformer.forward
(landiff/diffusion/dit_video_concat.py:872-1027,1196-1200) with the AdaLN layer of :540-629 / :1260-1372 and
sat's transformer internals as recorded in SURVEY.md 8c.  Every tensor op below is a kernel from
liblandiff_hip.so; torch only owns the buffers.

HBM layout: the joint sequence is a row-major [B*(text+T*h*w)][hidden] bf16 matrix (B = 2: uncond, cond);
q/k live as [B][H][Npad][64], v transposed as [B][H][64][Npad]; all workspaces are allocated once.
"""
from __future__ import annotations

import os

import torch

from . import ops
from .config import DiTConfig

BF = torch.bfloat16


def _dev(t, device, dtype=BF):
    return t.detach().to(device=device, dtype=dtype).contiguous()


class _Branch:
    """Packed weights of one DiffusionTransformer (control or main)."""

    def __init__(self, sd: dict, cfg: DiTConfig, control: bool, device, fp8_gemm: bool = False):
        d = cfg.hidden
        self.control = control
        self.L = cfg.layers_control if control else cfg.layers_main
        g = lambda k: _dev(sd[k], device)
        self.te0_w, self.te0_b = g("time_embed.0.weight"), g("time_embed.0.bias")
        self.te2_w, self.te2_b = g("time_embed.2.weight"), g("time_embed.2.bias")
        self.pos = g("mixins.pos_embed.pos_embedding")[0][: cfg.seq_len].contiguous()    # [seq, d]: the first text_len + seq_length rows (:227-231)
        self.patch_w = g("mixins.patch_embed.proj.weight").reshape(d, -1).contiguous()   # [d, C*p*p]
        self.patch_b = g("mixins.patch_embed.proj.bias")
        self.text_w, self.text_b = g("mixins.patch_embed.text_proj.weight"), g("mixins.patch_embed.text_proj.bias")
        self.layers = []
        for i in range(self.L):
            p = f"transformer.layers.{i}."
            a = "mixins.adaln_layer."
            lw = dict(
                ln1_w=g(p + "input_layernorm.weight"), ln1_b=g(p + "input_layernorm.bias"),
                qkv_w=g(p + "attention.query_key_value.weight"), qkv_b=g(p + "attention.query_key_value.bias"),
                dense_w=g(p + "attention.dense.weight"), dense_b=g(p + "attention.dense.bias"),
                ln2_w=g(p + "post_attention_layernorm.weight"), ln2_b=g(p + "post_attention_layernorm.bias"),
                h4_w=g(p + "mlp.dense_h_to_4h.weight"), h4_b=g(p + "mlp.dense_h_to_4h.bias"),
                h1_w=g(p + "mlp.dense_4h_to_h.weight"), h1_b=g(p + "mlp.dense_4h_to_h.bias"),
                ada_w=g(a + f"adaLN_modulations.{i}.1.weight"), ada_b=g(a + f"adaLN_modulations.{i}.1.bias"),
                qln=(g(a + f"query_layernorm_list.{i}.weight"), g(a + f"query_layernorm_list.{i}.bias"),
                     g(a + f"key_layernorm_list.{i}.weight"), g(a + f"key_layernorm_list.{i}.bias")),
            )
            if control:
                lw["zero_w"] = g(a + f"zero_linears.{i}.weight")
            if fp8_gemm:      # e4m3 weights (one scale per output channel, or MX blocks of 32 along K); the bf16 copies go
                quant = ops.quantize_mxfp8 if fp8_gemm == "mx" else ops.quantize_fp8
                for name in ("qkv", "dense", "h4", "h1"):
                    lw[name + "_w8"], lw[name + "_s"] = quant(lw.pop(name + "_w"))
            self.layers.append(lw)
        # all layers' adaLN modulation Linear stacked: the time embedding is the same for every layer of a step, so ONE
        # weight-streaming GEMV per branch and step produces all 12 x hidden modulation vectors (the per-layer tensors are
        # dropped: views into the stack would only keep two copies alive)
        self.ada_w_all = torch.cat([lw.pop("ada_w") for lw in self.layers], 0).contiguous()      # [L * 12 d, time_embed_dim]
        self.ada_b_all = torch.cat([lw.pop("ada_b") for lw in self.layers], 0).contiguous()
        if not control:
            f = "mixins.final_layer."
            self.fln_w, self.fln_b = g("transformer.final_layernorm.weight"), g("transformer.final_layernorm.bias")
            self.nf_w, self.nf_b = g(f + "norm_final.weight"), g(f + "norm_final.bias")
            self.lin_w, self.lin_b = g(f + "linear.weight"), g(f + "linear.bias")
            self.fada_w, self.fada_b = g(f + "adaLN_modulation.1.weight"), g(f + "adaLN_modulation.1.bias")


class ControlDiTRunner:
    """Evaluates the CFG pair (uncond, cond) of the control+main DiT for one sampler step."""

    B = 2

    def __init__(self, main_sd: dict, control_sd: dict, cfg: DiTConfig, device, fp8_gemm: bool = False,
                 attn_exact: "bool | str | None" = None):
        """fp8_gemm: BASELINE configs[4] -- the qkv / dense / 4h / 4h->h linears run on e4m3 operands; attention, norms,
        residual stream and every other layer stay bf16.  True / "row": weights quantised once per output channel,
        activations per row by a pass in front of each GEMM.  "mx": MXFP8 (one power-of-two scale per 32 K elements, applied
        by the MFMA); LayerNorm+modulate and the GELU epilogue write MXFP8 directly, only the attention output still takes
        a quantise pass.  Off for the headline metric and for every parity claim of the bf16 path."""
        self.cfg, self.dev, self.fp8 = cfg, device, fp8_gemm
        # attn_exact=True (or LD_DIT_ATTN_EXACT=1): every attention launch through ld_attn_fwd_bf16_exact -- for checkpoints whose
        # QK-LayerNorm gains push q.k/8 beyond the fast pass's window (~76): a fixed 1.55 x instead of data-dependent re-runs
        # (the default launch is exact inside the window and falls back by itself outside it, at up to 2.6 x).
        # "auto" (or LD_DIT_ATTN_EXACT=auto): per layer -- every default launch reports how many of its 256-row blocks fell back
        # (ld_attn_last_fallbacks); a layer where more than AUTO_EXACT_FRACTION of them did in a denoiser step takes the exact form
        # from the step after next on (the counts of step s are read while step s + 1 runs: no host wait on the GPU's critical
        # path, and the decision does not depend on timing).  The same checkpoint visits the same layers 50 times per video,
        # so the first two steps pay for the measurement; the choice sticks to the runner (reset_attn_auto()).
        # The default since it measured free on a checkpoint that never leaves the window (279.1 vs 279.4 ms per full-size step) and
        # 312.6 vs 377.0 ms on one that does in a third of its layers (profiles/r06_dit_attn_auto_policy.txt); LD_DIT_ATTN_EXACT=0 /
        # attn_exact=False: always the default launch, =1 / True: always the exact form.
        if attn_exact is None:
            attn_exact = {"1": True, "0": False}.get(os.environ.get("LD_DIT_ATTN_EXACT", "auto"), "auto")
        self.attn_auto = attn_exact == "auto"
        self.attn_exact = (not self.attn_auto) and bool(attn_exact)
        self.main = _Branch(main_sd, cfg, False, device, fp8_gemm)
        self.ctrl = _Branch(control_sd, cfg, True, device, fp8_gemm)
        c, B = cfg, self.B
        d, N = c.hidden, c.seq_len
        M = B * N
        self.N, self.M = N, M
        self.Npad = (N + 127) // 128 * 128
        e = lambda *s, dt=BF: torch.empty(*s, device=device, dtype=dt)
        self.h = e(M, d)                        # main residual stream
        self.hc = e(M, d)                       # control working stream
        self.ctrl_out = [e(M, d) for _ in range(c.layers_control)]
        # default (LD_DIT_OVERLAP=0 selects the serial step), see _step_overlapped(): the control branch runs on a second stream, one
        # layer ahead of the main branch, and needs its own copy of every per-layer workspace
        # (round 6: also with e4m3 linears -- the quantised-activation buffers are part of the per-chain workspace set)
        self.overlap = os.environ.get("LD_DIT_OVERLAP", "1") == "1"
        self._sets = []
        for _ in range(2 if self.overlap else 1):
            ws = dict(ln=e(M, d), qkv=e(M, 3 * d),
                      q=torch.zeros(B, c.heads, self.Npad, 64, device=device, dtype=BF),
                      k=torch.zeros(B, c.heads, self.Npad, 64, device=device, dtype=BF),
                      vt=torch.zeros(B, c.heads, 64, self.Npad, device=device, dtype=BF),
                      attn=e(B, N, d), mlp=e(M, 4 * d), patches=e(c.n_img, c.in_channels * c.patch * c.patch),
                      temb=e(B, d), emb_h=e(B, c.time_embed_dim), emb=e(B, c.time_embed_dim), tvec=e(B, dt=torch.float32))
            if fp8_gemm:
                ws["a8"] = torch.empty(M, 4 * d, device=device, dtype=torch.uint8)    # quantised GEMM input (largest K)
                ws["sa"] = e(M, dt=torch.float32)
            if fp8_gemm == "mx":
                ws["a8d"] = torch.empty(M, d, device=device, dtype=torch.uint8)        # MXFP8 activations of width d ...
                ws["s8d"] = torch.empty(d // 128, M, 4, device=device, dtype=torch.uint8)       # scales, K-tile-major
                ws["s8m"] = torch.empty(4 * d // 128, M, 4, device=device, dtype=torch.uint8)   # ... and 4d (codes in a8)
            self._sets.append(ws)
        self._use(0)
        self._side = torch.cuda.Stream(device=device) if self.overlap else None
        self._ctrl_done = [torch.cuda.Event() for _ in range(c.layers_control)] if self.overlap else None
        # modulation vectors of every layer of a branch for the current step: [B][L][12 d]
        self.ada_main = e(B, self.main.L * 12 * d)
        self.ada_ctrl = e(B, self.ctrl.L * 12 * d)
        self.fada = e(B, 2 * d)
        self.lin = e(B, c.n_img, c.patch * c.patch * c.out_channels)
        self.txt_main = e(B, c.text_len, d)
        self.txt_ctrl = e(B, c.text_len, d)
        self.sem = None                         # [T, C, H, W] bf16, set per video
        # qkv Linear + QK-LayerNorm + head split + V transpose in one launch (ld_gemm_qkv_heads) where its shape rules hold;
        # LD_DIT_FUSE_QKV=0 keeps the two-launch form (A/B timing)
        # (round 6: also on MXFP8 operands, ld_gemm_qkv_heads_mxfp8; the row-scaled fp8 form keeps the two launches)
        self.fuse_qkv = (self.fp8 in (False, None, "mx") and self.N % 8 == 0 and self.N >= 256 and c.head_dim == 64
                         and os.environ.get("LD_DIT_FUSE_QKV", "1") != "0")
        self._solo = True
        self._attn_slot = 0                     # which layer's attention is being launched: control layers first, then main
        self.exact_layers = set()               # slots switched to ld_attn_fwd_bf16_exact by the "auto" policy
        if self.attn_auto:
            L = c.layers_control + c.layers_main
            self._fb = torch.zeros(L, device=device, dtype=torch.int32)
            self._fb_host = [torch.zeros(L, dtype=torch.int32).pin_memory() for _ in range(2)]
            self._fb_ev = [torch.cuda.Event(), torch.cuda.Event()]
            self._fb_pending = [False, False]
            self._fb_par = 0
            self._attn_blocks = B * c.heads * ((self.Npad + 255) // 256)
        self.attn_events = None                 # bench.py: list of (start, end, solo) HIP events around every attention launch
        self.gemm_events = None                 # bench.py: list of (start, end, flops) around the large linears (qkv, dense, 4h, 4h->h, zero)

    def _use(self, i: int):
        """Select the workspace set the layer methods read through self.ln / self.q / ... (set 1 exists only when the two chains overlap)."""
        for k, v in self._sets[i].items():
            setattr(self, k, v)

    # ---- per-video setup -------------------------------------------------------------------
    def set_condition(self, context: torch.Tensor, semantic_feature: torch.Tensor):
        """context [1, text_len, text_dim] (T5 states of the prompt; the uncond branch uses zeros,
        dif_infer.py:214-218), semantic_feature [T, C, H, W] bf16 (SemanticCond output, cached per video)."""
        c = self.cfg
        ctx = torch.zeros(self.B, c.text_len, c.text_dim, device=self.dev, dtype=BF)
        ctx[1] = context.to(self.dev, BF)[0]
        for br, dst in ((self.main, self.txt_main), (self.ctrl, self.txt_ctrl)):
            for b in range(self.B):   # text_proj + position rows (zero in the shipped table, added all the same)
                ops.gemm(ctx[b], br.text_w, out=dst[b], bias=br.text_b, add2=br.pos[: c.text_len])
        self.sem = semantic_feature.to(self.dev, BF).contiguous()

    # ---- pieces ----------------------------------------------------------------------------
    def _time_emb(self, br: _Branch, timestep: float):
        self.tvec.fill_(float(timestep))
        ops.timestep_embedding(self.tvec, self.temb)
        ops.gemv(self.temb, br.te0_w, self.emb_h, bias=br.te0_b)
        ops.gemv(self.emb_h, br.te2_w, self.emb, bias=br.te2_b, in_act="silu")
        self._modulations(br)

    def _modulations(self, br: _Branch):
        """adaLN_modulations[l] = Linear(SiLU(emb)) for every layer l of the branch (dit_video_concat.py:540-566), one launch:
        self.emb [B, time_embed_dim] -> [B, L * 12 hidden]."""
        ops.gemv(self.emb, br.ada_w_all, self._ada_all(br), bias=br.ada_b_all, in_act="silu")

    def _ada_all(self, br: _Branch):
        return self.ada_ctrl if br is self.ctrl else self.ada_main

    def _ada(self, br: _Branch, i: int):
        """(view of layer i's 12 x hidden modulation vectors [B, 12 d], batch stride in elements)"""
        a, d12 = self._ada_all(br), 12 * self.cfg.hidden
        return a[:, i * d12:(i + 1) * d12], a.stride(0)

    def _embed(self, br: _Branch, x: torch.Tensor, h: torch.Tensor, txt: torch.Tensor, sem):
        c = self.cfg
        ops.patchify(x, sem, self.patches, c.patch)
        hv = h.view(self.B, self.N, c.hidden)
        for b in range(self.B):      # the image half is identical for uncond/cond; text rows differ
            ops.gemm(self.patches, br.patch_w, out=hv[b, c.text_len:], bias=br.patch_b, add2=br.pos[c.text_len:])
            hv[b, :c.text_len].copy_(txt[b])

    def _timed(self, flops, fn, *a, **kw):
        """fn(*a, **kw), bracketed by HIP events on the current stream when bench.py asked for GEMM timings.  flops: the launch's
        2 M N K, or (2 M N K, "mx") for an e4m3 linear (bench.py prices those against the MX-fp8 peak)."""
        if self.gemm_events is None:
            return fn(*a, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **kw)
        e1.record()
        self.gemm_events.append((e0, e1, flops))
        return r

    def _linear(self, x: torch.Tensor, lw: dict, name: str, out: torch.Tensor, **epi):
        """One of the four large nn.Linear sites of a block: bf16 MFMA GEMM, or (fp8_gemm) quantise + e4m3 GEMM."""
        if not self.fp8:
            w = lw[name + "_w"]
            return self._timed(2.0 * x.shape[0] * w.shape[0] * w.shape[1], ops.gemm, x, w, out=out, bias=lw[name + "_b"], **epi)
        K = x.shape[1]
        a8 = self.a8.view(-1)[: x.shape[0] * K].view(x.shape[0], K)
        ops.quantize_fp8(x, a8, self.sa)
        return ops.gemm_fp8(a8, self.sa, lw[name + "_w8"], lw[name + "_s"], out=out, bias=lw[name + "_b"], **epi)

    AUTO_EXACT_FRACTION = 0.25      # profiles/r06_attn_logit_sweep.txt: beyond ~25 % of the blocks falling back, the exact form is cheaper

    def _attention(self):
        c, N = self.cfg, self.N
        exact = self.attn_exact or (self._attn_slot in self.exact_layers)
        if self.attn_events is not None:       # bench.py: HIP events around every attention launch (roofline.achieved)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.attn_fwd(self.q, self.k, self.vt, self.attn, N, N, c.head_dim ** -0.5, exact=exact)
            e1.record()
            self.attn_events.append((e0, e1, self._solo))       # _solo: no other stream has work queued next to this launch
        else:
            ops.attn_fwd(self.q, self.k, self.vt, self.attn, N, N, c.head_dim ** -0.5, exact=exact)
        if self.attn_auto and not exact:
            ops.attn_last_fallbacks(self._fb[self._attn_slot:self._attn_slot + 1])     # (on the stream of the launch)

    def _attn_auto_end_of_step(self):
        """attn_exact="auto": hand this step's per-layer fallback counts to the host (asynchronously) and act on the previous step's."""
        par = self._fb_par
        self._fb_host[par].copy_(self._fb, non_blocking=True)
        self._fb_ev[par].record()
        self._fb_pending[par] = True
        prev = 1 - par
        if self._fb_pending[prev]:
            self._fb_ev[prev].synchronize()          # the step before this one: done, or about to be -- the GPU stays one step ahead
            self._fb_pending[prev] = False
            limit = self.AUTO_EXACT_FRACTION * self._attn_blocks
            self.exact_layers.update(i for i, n in enumerate(self._fb_host[prev].tolist()) if n > limit)
        self._fb_par = prev

    def reset_attn_auto(self):
        """Forget which layers the "auto" policy switched to the exact attention form (a new checkpoint in the same runner)."""
        self.exact_layers.clear()
        if self.attn_auto:
            self._fb.zero_()
            self._fb_pending = [False, False]

    def _layer_mx(self, br: _Branch, i: int, h_in: torch.Tensor, h_out: torch.Tensor, control_add=None):
        """_layer with MXFP8 operands on the four large linears (fp8_gemm="mx")."""
        c, lw = self.cfg, br.layers[i]
        d, N = c.hidden, self.N
        ada, ada_bs = self._ada(br, i)
        mod = dict(mod=ada, mod_bstride=ada_bs, rows_per_batch=N, text_len=c.text_len)
        gate = dict(gate=ada, gate_bstride=ada_bs, rows_per_batch=N, text_len=c.text_len)
        ops.layernorm_mxfp8(h_in, lw["ln1_w"], lw["ln1_b"], self.a8d, self.s8d, c.block_ln_eps, shift_img=0, scale_img=d,
                            shift_txt=6 * d, scale_txt=7 * d, **mod)
        fl = 2.0 * self.M * d * d                   # (bench.py: HIP events around the e4m3 linears too, flops tagged "mx")
        if self.fuse_qkv:
            self._timed((3 * fl, "mx"), ops.gemm_qkv_heads_mxfp8, self.a8d, self.s8d, lw["qkv_w8"], lw["qkv_s"], lw["qkv_b"], self.q, self.k, self.vt,
                        self.B, N, c.heads, self.Npad, lw["qln"], eps=c.qk_ln_eps)
        else:
            self._timed((3 * fl, "mx"), ops.gemm_mxfp8, self.a8d, self.s8d, lw["qkv_w8"], lw["qkv_s"], out=self.qkv, bias=lw["qkv_b"])
            ops.qkv_split(self.qkv, self.q, self.k, self.vt, self.B, N, c.heads, self.Npad, ln=lw["qln"], eps=c.qk_ln_eps)
        self._attention()
        ops.quantize_mxfp8(self.attn.view(-1, d), self.a8d, self.s8d)
        self._timed((fl, "mx"), ops.gemm_mxfp8, self.a8d, self.s8d, lw["dense_w8"], lw["dense_s"], out=h_out, bias=lw["dense_b"], resid=h_in,
                    gate_off_img=2 * d, gate_off_txt=8 * d, **gate)
        ops.layernorm_mxfp8(h_out, lw["ln2_w"], lw["ln2_b"], self.a8d, self.s8d, c.block_ln_eps, shift_img=3 * d, scale_img=4 * d,
                            shift_txt=9 * d, scale_txt=10 * d, **mod)
        a8m = self.a8.view(self.M, 4 * d)
        self._timed((4 * fl, "mx"), ops.gemm_mxfp8, self.a8d, self.s8d, lw["h4_w8"], lw["h4_s"], out=a8m, out_scales=self.s8m, bias=lw["h4_b"], act="gelu_tanh")
        self._timed((4 * fl, "mx"), ops.gemm_mxfp8, a8m, self.s8m, lw["h1_w8"], lw["h1_s"], out=h_out, bias=lw["h1_b"], resid=h_out, gate_off_img=5 * d,
                    gate_off_txt=11 * d, add2=control_add, **gate)

    def _layer(self, br: _Branch, i: int, h_in: torch.Tensor, h_out: torch.Tensor, control_add=None):
        # with the two chains overlapped, only the main layers behind the last control state run with the GPU to themselves
        self._solo = (not self.overlap) or (br is self.main and i > self.cfg.layers_control)
        self._attn_slot = i if br is self.ctrl else self.cfg.layers_control + i
        if self.fp8 == "mx":
            return self._layer_mx(br, i, h_in, h_out, control_add)
        c, lw = self.cfg, br.layers[i]
        d, N = c.hidden, self.N
        ada, ada_bs = self._ada(br, i)
        mod = dict(mod=ada, mod_bstride=ada_bs, rows_per_batch=N, text_len=c.text_len)
        ops.layernorm(h_in, lw["ln1_w"], lw["ln1_b"], self.ln, c.block_ln_eps, shift_img=0, scale_img=d,
                      shift_txt=6 * d, scale_txt=7 * d, **mod)
        if self.fuse_qkv:
            # qkv Linear + head split + QK-LayerNorm + V transpose in one launch (q / k / vt padding rows stay zero from allocation)
            self._timed(2.0 * self.M * 3 * d * d, ops.gemm_qkv_heads, self.ln, lw["qkv_w"], lw["qkv_b"], self.q, self.k, self.vt,
                        self.B, N, c.heads, self.Npad, lw["qln"], eps=c.qk_ln_eps)
        else:
            self._linear(self.ln, lw, "qkv", self.qkv)
            ops.qkv_split(self.qkv, self.q, self.k, self.vt, self.B, N, c.heads, self.Npad, ln=lw["qln"], eps=c.qk_ln_eps)
        self._attention()
        gate = dict(gate=ada, gate_bstride=ada_bs, rows_per_batch=N, text_len=c.text_len)
        self._linear(self.attn.view(-1, d), lw, "dense", h_out, resid=h_in, gate_off_img=2 * d, gate_off_txt=8 * d, **gate)
        ops.layernorm(h_out, lw["ln2_w"], lw["ln2_b"], self.ln, c.block_ln_eps, shift_img=3 * d, scale_img=4 * d,
                      shift_txt=9 * d, scale_txt=10 * d, **mod)
        self._linear(self.ln, lw, "h4", self.mlp, act="gelu_tanh")
        self._linear(self.mlp, lw, "h1", h_out, resid=h_out, gate_off_img=5 * d, gate_off_txt=11 * d, add2=control_add, **gate)

    # ---- one denoiser evaluation -----------------------------------------------------------
    def _control_chain(self, h_in: torch.Tensor, n_layers: int | None = None):
        """Control branch: layer l+1 consumes zero_linear_l(layer_l(.)) (SURVEY Appendix C.6,
        dit_video_concat.py:1224-1238).  h_in [B*N, d] -> self.ctrl_out[0 .. n_layers)."""
        c = self.cfg
        for i in range(c.layers_control if n_layers is None else n_layers):
            self._layer(self.ctrl, i, h_in, self.hc)
            self._timed(2.0 * self.M * c.hidden * c.hidden, ops.gemm, self.hc, self.ctrl.layers[i]["zero_w"], out=self.ctrl_out[i])
            h_in = self.ctrl_out[i]

    def _main_chain(self, n_layers: int | None = None):
        """Main branch on self.h in place; layer i < layers_control adds the control state i (dit_video_concat.py:1357-1370)."""
        c = self.cfg
        for i in range(c.layers_main if n_layers is None else n_layers):
            self._layer(self.main, i, self.h, self.h, self.ctrl_out[i] if i < c.layers_control else None)

    def _final(self, h: torch.Tensor, x: torch.Tensor, c_out: float, c_skip: float, cfg_scale: float, out: torch.Tensor,
               sat_final_layernorm: bool = True):
        """sat's transformer.final_layernorm, then FinalLayerMixin.final_forward (dit_video_concat.py:442-456: adaLN modulation of
        norm_final over the image tokens, Linear, unpatchify), the denoiser scaling and the CFG combine.  h [B*N, d] bf16;
        self.emb holds the step's time embedding.  sat_final_layernorm=False starts at final_forward (the seam the reference's
        own code owns; tests/test_gpu_golden.py feeds it the golden final_in)."""
        c, m, d, N = self.cfg, self.main, self.cfg.hidden, self.N
        if sat_final_layernorm:
            ops.layernorm(h, m.fln_w, m.fln_b, self.ln, c.block_ln_eps)
            h = self.ln
        ops.gemv(self.emb, m.fada_w, self.fada, bias=m.fada_b, in_act="silu")
        ops.layernorm(h, m.nf_w, m.nf_b, self.ln, c.final_ln_eps, mod=self.fada, mod_bstride=2 * d, shift_img=0,
                      scale_img=d, shift_txt=0, scale_txt=d, rows_per_batch=N, text_len=0)
        lnv = self.ln.view(self.B, N, d)
        for b in range(self.B):
            ops.gemm(lnv[b, c.text_len:], m.lin_w, out=self.lin[b], bias=m.lin_b)
        ops.unpatchify_cfg(self.lin, x, out, c.patch, c_out, c_skip, cfg_scale)
        return out

    def _step_overlapped(self, x, timestep, c_out, c_skip, cfg_scale, out):
        """The same launches as step(), the control branch on a second stream: control layer l + 1 needs only control layer l,
        main layer l needs main layer l - 1 and control state l, so the two chains run one layer apart and the partial last round
        of one kernel (attention: 4200 workgroups on 512 slots; the GEMMs' M-split tail launches) fills with the other chain's
        workgroups.  Bit-identical results (no arithmetic changes).  Measured: -1.8 ... -3 % per denoiser step
        (profiles/r04_dit_overlap_ab.txt).  HIP-event durations of launches that have a partner include the partner's work, so
        bench.py takes a kernel's own duration from the main layers behind the last control state (`_solo`), which run alone."""
        c = self.cfg
        main_s = torch.cuda.current_stream(self.dev)
        self._side.wait_stream(main_s)                    # x (the sampler's update) and last step's reads of ctrl_out are ordered
        with torch.cuda.stream(self._side):
            self._use(1)
            self._time_emb(self.ctrl, timestep)
            self._embed(self.ctrl, x, self.hc, self.txt_ctrl, self.sem)
            h_in = self.hc
            for i in range(c.layers_control):
                self._layer(self.ctrl, i, h_in, self.hc)
                self._timed(2.0 * self.M * c.hidden * c.hidden, ops.gemm, self.hc, self.ctrl.layers[i]["zero_w"], out=self.ctrl_out[i])
                self._ctrl_done[i].record(self._side)
                h_in = self.ctrl_out[i]
        self._use(0)
        self._time_emb(self.main, timestep)
        self._embed(self.main, x, self.h, self.txt_main, None)
        for i in range(c.layers_main):
            if i < c.layers_control:
                main_s.wait_event(self._ctrl_done[i])
            self._layer(self.main, i, self.h, self.h, self.ctrl_out[i] if i < c.layers_control else None)
        return self._final(self.h, x, c_out, c_skip, cfg_scale, out)

    def step(self, x: torch.Tensor, timestep: int, c_out: float, c_skip: float, cfg_scale: float, out: torch.Tensor):
        """x, out: [1, T, C, H, W] fp32.  out = CFG(denoised_uncond, denoised_cond)."""
        r = self._step(x, timestep, c_out, c_skip, cfg_scale, out)
        if self.attn_auto:
            self._attn_auto_end_of_step()
        return r

    def _step(self, x, timestep, c_out, c_skip, cfg_scale, out):
        if self.overlap:
            return self._step_overlapped(x, timestep, c_out, c_skip, cfg_scale, out)
        self._time_emb(self.ctrl, timestep)
        self._embed(self.ctrl, x, self.hc, self.txt_ctrl, self.sem)
        self._control_chain(self.hc)
        self._time_emb(self.main, timestep)
        self._embed(self.main, x, self.h, self.txt_main, None)
        self._main_chain()
        return self._final(self.h, x, c_out, c_skip, cfg_scale, out)

"""Autoregressive semantic-token decode on MI355X.

Mirrors Semantic1DLM.tokenize / .sample (landiff/llm/models/lm_model.py:175-516), GPT.sample
(landiff/llm/models/transformer.py:91-119), LlamaTransformerBlock with the KV-cache path
(landiff/llm/modules/transformer_blocks.py:128-236), TextCond / MicroConditioner
(landiff/llm/modules/conditioner.py:90-170,230-323) and Rope1DPosEmb (landiff/modules/pos_emb.py:73-123).

What changes is mechanism only: weights are cast to bf16 once (the reference re-casts 2 B parameters every
forward under autocast), the KV cache is preallocated in HBM and appended in place, position / current token /
forced-token schedule live in device memory, and the per-token step is captured in a HIP graph and replayed.
torch.multinomial stays on PyTorch-ROCm so the RNG stream is the reference's (one draw per step, forced or not).
"""
from __future__ import annotations

import math

import os
import time

import torch

from . import _lib, ops
from .config import LLMConfig

BF = torch.bfloat16


def forced_token_schedule(cfg: LLMConfig, S: int, num_frames: int):
    """Position bookkeeping of lm_model.py:323-396 (END tokens enabled): returns (full_len, forced {pos: id},
    restricted {pos: [ids]}, n_visual).  S is the position of the first START_OF_IFrame."""
    I, P, seg, stride = cfg.iframe_len, cfg.pframe_len, cfg.segment_length, cfg.segment_stride
    code_len = 0
    for off in range(0, num_frames, stride):
        fl = min(off + seg, num_frames) - off
        code_len += I + (fl - 1) * P + 2 * fl
    full_len = S + code_len + 1
    block = I + (seg - 1) * P + 2 * seg
    start_i, end_i, start_p, end_p, eos_ok = set(), set(), set(), set(), set()
    for index in range(S, full_len - 1, block):          # one block per segment
        start_i.add(index)
        pos = index + 1 + I                               # END_OF_IFrame slot
        end_i.add(pos)
        pos += 1
        if index > S:
            eos_ok.add(pos)
        p_end = min(full_len - 1, pos - 1 + P * (seg - 1) + 2 * (seg - 1))
        for j in range(pos, p_end, P + 2):                # START_P, P tokens, END_P
            start_p.add(j)
            end_p.add(j + P + 1)
            if index > S:
                eos_ok.add(j + P + 2)
    forced, restricted = {}, {}
    for i in range(S + 1, full_len):
        allowed = [t for t, st in ((cfg.START_I, start_i), (cfg.START_P, start_p), (cfg.EOS, eos_ok)) if i in st]
        if allowed:
            restricted[i] = allowed
        for t, st in ((cfg.START_I, start_i), (cfg.END_I, end_i), (cfg.START_P, start_p), (cfg.END_P, end_p)):
            if i in st:
                forced[i] = t
                break
        else:
            if i == full_len - 1:
                forced[i] = cfg.EOS
    n_visual = (full_len - S - 1) - len(forced)
    return full_len, forced, restricted, n_visual


class LLMRunner:
    B = 2   # (cond, uncond)

    def __init__(self, sd: dict, cfg: LLMConfig, device, max_text: int = 512, max_frames: int = 13):
        self.cfg, self.dev = cfg, device
        c = cfg
        g = lambda k, dt=BF: sd[k].detach().to(device=device, dtype=dt).contiguous()
        self.blocks = []
        for i in range(c.num_layers):
            p = f"transformer.blocks.{i}."
            self.blocks.append(dict(n0=g(p + "norm0.weight", torch.float32), n1=g(p + "norm1.weight", torch.float32),
                                    wqkv=g(p + "wqkv.weight"), wo=g(p + "wo.weight"), w1=g(p + "mlp.w1.weight"),
                                    w3=g(p + "mlp.w3.weight"), w2=g(p + "mlp.w2.weight")))
        self.ln_w, self.ln_b = g("transformer.layer_norm.weight", torch.float32), g("transformer.layer_norm.bias", torch.float32)
        self.head = g("transformer.head.weight", torch.float32)
        self.emb = g("visual_embedding_model.tok_emb_code.weight", torch.float32)
        self.fc0 = (g("cond_model.embeddings.fc0.weight"), g("cond_model.embeddings.fc0.bias"))
        self.fc1 = (g("cond_model.embeddings.fc1.weight"), g("cond_model.embeddings.fc1.bias"))
        self.null_text = g("cond_model.null_text_embedding")
        self.micro = {k: (g(f"micro_condition.mlps.{k}.0.weight"), g(f"micro_condition.mlps.{k}.0.bias"),
                          g(f"micro_condition.mlps.{k}.2.weight"), g(f"micro_condition.mlps.{k}.2.bias"))
                      for k in ("frames", "motion_score")}
        # RoPE table (pos_emb.py:49-70)
        n_seg = -(-max_frames // c.segment_stride)
        self.Lmax = max_text + 4 + n_seg * (c.iframe_len + (c.segment_length - 1) * c.pframe_len + 2 * c.segment_length) + 2
        freqs = 1.0 / (c.rope_theta ** (torch.arange(0, c.head_dim, 2)[: c.head_dim // 2].float() / c.head_dim))
        ang = torch.outer(torch.arange(self.Lmax).float(), freqs).float()
        cis = torch.polar(torch.ones_like(ang), ang)
        self.cos, self.sin = cis.real.contiguous().to(device), cis.imag.contiguous().to(device)
        # state in HBM
        B, H, D = self.B, c.heads, c.head_dim
        self.kc = [torch.zeros(B, self.Lmax, H, D, device=device, dtype=BF) for _ in range(c.num_layers)]
        self.vc = [torch.zeros(B, self.Lmax, H, D, device=device, dtype=BF) for _ in range(c.num_layers)]
        self.pos = torch.zeros(1, device=device, dtype=torch.int32)
        self.pos0 = torch.zeros(1, device=device, dtype=torch.int32)
        self.token = torch.zeros(1, device=device, dtype=torch.int64)
        self.sampled = torch.zeros(1, 1, device=device, dtype=torch.int64)
        self.out_tokens = torch.zeros(self.Lmax, device=device, dtype=torch.int64)
        self.out_count = torch.zeros(1, device=device, dtype=torch.int32)
        self.forced = torch.full((self.Lmax + 2,), -1, device=device, dtype=torch.int32)
        self.allowed = torch.zeros(self.Lmax + 2, 4, device=device, dtype=torch.int32)
        e = lambda *s, dt=BF: torch.empty(*s, device=device, dtype=dt)
        self.x = e(B, c.hidden)
        self.xn = e(B, c.hidden)
        self.qkv = e(B, 3 * c.hidden)
        self.qr = e(B, c.hidden)
        self.att = e(B, c.hidden)
        self.gate = e(B, c.mlp)
        self.lnf = e(B, c.hidden, dt=torch.float32)
        self.logits = e(B, c.vocab, dt=torch.float32)
        self.probs = e(1, c.vocab, dt=torch.float32)
        self.cfg_logits = e(1, c.vocab, dt=torch.float32)
        self.noise = e(1, c.vocab, dt=torch.float32)       # Exp(1) draws of the step's torch.multinomial (see _sample_and_advance)
        # LD_LLM_FUSED_TAIL=0 (read once): sampling through torch.multinomial + ld_llm_decode_advance + ld_llm_embed (13 + 3 launches)
        self.fused_tail = os.environ.get("LD_LLM_FUSED_TAIL", "1") != "0"
        self._x_from_tail = False                          # the sampling launch has left the next token's embedding in self.x
        # split-K decode attention: >= one workgroup per CU, and at most 256 keys per split (ld_llm_kv_attn)
        self.nsplit = max(1, 256 // (B * H), -(-self.Lmax // 256))
        self.top_k, self.top_p = None, None
        # partial results [B*H][nsplit][130] + B*H arrival counters (zero between launches: the last split to arrive merges)
        self.attn_ws = torch.zeros(B * H * (self.nsplit * 130 + 1), device=device, dtype=torch.float32)
        self._graph = None
        self._layer_table = None
        # Three forms of a decode step's blocks, identical bits (tests/variants/variant_cases.py):
        #   "chain"    one launch per operation on the current stream (5 per block) -- what the shipped library has;
        #   "chained"  the same launches alternating between two streams with device-side dependencies (every launch requests
        #              its first weight rows while its predecessor is still running; ld_llm_decode_blocks_chained);
        #   "fused"    all blocks in one persistent launch with grid barriers (ld_llm_fused.hip).
        # The last two were measured slower and live in the VARIANTS build of the library only (landiff_amd/_lib.py:
        # VARIANTS_LIB_PATH, selected with LANDIFF_HIP_LIB): asking for them on the shipped library raises.
        # "chained" and "fused" need the GPU to themselves (their workgroups wait for other workgroups): callers that decode
        # UNDER another stream's kernels (generate_many, the streaming loop) pass mode="chain".  LD_LLM_DECODE overrides the default.
        self.decode_mode = os.environ.get("LD_LLM_DECODE", "chain")
        assert self.decode_mode in ("chain", "chained", "fused"), self.decode_mode
        self.fused_supported = (B == 2 and c.head_dim == 128 and c.hidden <= 2048 and c.mlp <= 12288 and self.nsplit >= 2
                                and -(-self.Lmax // self.nsplit) <= 256)
        self.chained_supported = (B == 2 and c.head_dim == 128 and c.hidden <= 4096 and c.mlp <= 12288 and self.nsplit >= 2
                                  and -(-self.Lmax // self.nsplit) <= 256 and 5 * c.num_layers <= 256)
        self.fused_ctl = torch.zeros(ops.LLM_FUSED_CTL_WORDS, device=device, dtype=torch.int32)
        self.chain_ctl = torch.zeros(ops.LLM_CHAIN_CTL_WORDS, device=device, dtype=torch.int32)
        self._mode = "chain"               # the form the running decode uses
        self._layer_table_dev = None
        self._side = None                  # second stream + event of the chained form
        self._chain_ev = None
        self._chain_epoch = 0
        self._pos_host = -1                # host mirror of *pos, valid only inside sample()'s loop; -1: the kernels read *pos
        self._capturing = False

    # ---- conditioning ------------------------------------------------------------------------
    def _micro_cond(self, frames: float, motion_score: float):
        c, dev = self.cfg, self.dev
        out = torch.empty(2, c.hidden, device=dev, dtype=BF)
        t = torch.empty(1, device=dev, dtype=torch.float32)
        te = torch.empty(1, c.freq_dim, device=dev, dtype=BF)
        hid = torch.empty(1, c.micro_hidden, device=dev, dtype=BF)
        for r, (key, val) in enumerate((("frames", frames), ("motion_score", motion_score))):
            w0, b0, w2, b2 = self.micro[key]
            t.fill_(float(val))
            ops.timestep_embedding(t, te)
            ops.gemv(te, w0, hid, bias=b0)
            ops.gemv(hid, w2, out[r:r + 1], bias=b2, in_act="silu")
        return out

    def prefix_features(self, text_emb: torch.Tensor, frames: float, motion_score: float):
        """[BOS][frames][motion][text x n][START_I] for (cond, uncond) -> bf16 [2, n+4, hidden]."""
        c, dev = self.cfg, self.dev
        n = text_emb.shape[0]
        t = text_emb.to(dev, BF).contiguous()
        h = ops.gemm(t, self.fc0[0], bias=self.fc0[1], act="gelu_tanh")
        cond = ops.gemm(h, self.fc1[0], bias=self.fc1[1])
        micro = self._micro_cond(frames, motion_score)
        feats = torch.empty(2, n + 4, c.hidden, device=dev, dtype=BF)
        feats[:, 0] = self.emb[c.BOS].to(BF)
        feats[:, 1:3] = micro
        feats[0, 3:3 + n] = cond
        feats[1, 3:3 + n] = self.null_text
        feats[:, 3 + n] = self.emb[c.START_I].to(BF)
        return feats

    # ---- transformer ---------------------------------------------------------------------------
    def _prefill(self, feats: torch.Tensor):
        c = self.cfg
        B, m, d = feats.shape
        M = B * m
        x = feats.reshape(M, d).contiguous()
        xn = torch.empty_like(x)
        qkv = torch.empty(M, 3 * d, device=self.dev, dtype=BF)
        qr = torch.empty(M, d, device=self.dev, dtype=BF)
        att = torch.empty(M, d, device=self.dev, dtype=BF)
        h3 = torch.empty(M, c.mlp, device=self.dev, dtype=BF)
        gate = torch.empty(M, c.mlp, device=self.dev, dtype=BF)
        self.pos0.zero_()
        for i, w in enumerate(self.blocks):
            ops.rmsnorm(x, w["n0"], xn, c.rms_eps)
            ops.gemm(xn, w["wqkv"], out=qkv)
            ops.llm_rope_append(qkv, self.cos, self.sin, self.pos0, qr, self.kc[i], self.vc[i], B, m, c.heads, self.Lmax)
            ops.llm_kv_attn(qr, self.kc[i], self.vc[i], self.pos0, att, B, m, c.heads, self.Lmax)
            ops.gemm(att, w["wo"], out=x, resid=x)
            ops.rmsnorm(x, w["n1"], xn, c.rms_eps)
            ops.gemm(xn, w["w3"], out=h3)
            ops.gemm(xn, w["w1"], out=gate, act="gelu_tanh", mul=h3)
            ops.gemm(gate, w["w2"], out=x, resid=x)
        last = x.view(B, m, d)[:, -1]                       # rows (b, m-1), stride m*d
        ops.layernorm_bf16_to_f32(last, self.ln_w, self.ln_b, self.lnf, c.ln_eps)
        ops.gemv(self.lnf, self.head, self.logits)

    def _decode_forward(self):
        """One token: embedding of *token at position *pos -> logits [2, V].  All sizes are static and every per-step scalar
        lives on the device; the ~125 launches are queued by ONE native call (ld_llm_decode_forward) -- issued one by one
        from Python the step was bound by the interpreter (1.3 ms of host time per step), and replaying it as a HIP graph
        costs 1.2 ms of host time per launch of the 164-node graph."""
        c = self.cfg
        if self._layer_table is None:
            self._layer_table = ops.llm_layer_table(self.blocks, self.kc, self.vc)
        if self._mode == "chained":
            s0 = torch.cuda.current_stream(self.dev)
            if not self._x_from_tail:
                ops.llm_embed(self.emb, self.token, self.x)
            ops.llm_decode_blocks_chained(self._layer_table, self._pos_host, self.x, self.qkv, self.att, self.gate, self.attn_ws,
                                          self.cos, self.sin, c.heads, self.Lmax, self.nsplit, c.rms_eps, self.chain_ctl,
                                          self._chain_epoch, s0, self._side)
            self._chain_epoch += 1
            self._chain_ev.record(self._side)              # the step's last operation may sit on either stream: the tail follows both
            s0.wait_event(self._chain_ev)
            ops.layernorm_bf16_to_f32(self.x, self.ln_w, self.ln_b, self.lnf, c.ln_eps)
            ops.gemv(self.lnf, self.head, self.logits)
            return
        if self._mode == "fused":
            if self._layer_table_dev is None:
                self._layer_table_dev = ops.llm_layer_table_device(self._layer_table, self.dev)
            ops.llm_decode_forward_fused(self._layer_table_dev, len(self.blocks), None if self._x_from_tail else self.emb, self.token,
                                         self.pos, self.x, self.qkv, self.att, self.gate, self.attn_ws, self.cos, self.sin,
                                         self.ln_w, self.ln_b, self.lnf, self.head, self.logits, c.heads, self.Lmax, self.nsplit,
                                         c.rms_eps, c.ln_eps, self.fused_ctl)
            return
        ops.llm_decode_forward(self._layer_table, None if self._x_from_tail else self.emb, self.token, self.pos, self.x, self.qkv, self.att, self.gate,
                               self.attn_ws, self.cos, self.sin, self.ln_w, self.ln_b, self.lnf, self.head, self.logits,
                               c.heads, self.Lmax, self.nsplit, c.rms_eps, c.ln_eps,
                               pos_value=-1 if self._capturing else self._pos_host)   # (a captured graph must read *pos itself)

    def _decode_forward_per_op(self):
        """The same step issued op by op through the C-ABI (kept for tests: must equal _decode_forward bit for bit)."""
        c, B = self.cfg, self.B
        ops.llm_embed(self.emb, self.token, self.x)
        for i, w in enumerate(self.blocks):      # 6 launches per layer: RMSNorm rides in the GEMVs, RoPE/append in attention
            ops.gemv(self.x, w["wqkv"], self.qkv, norm_w=w["n0"], norm_eps=c.rms_eps)
            ops.llm_kv_attn(None, self.kc[i], self.vc[i], self.pos, self.att, B, 1, c.heads, self.Lmax,
                            workspace=self.attn_ws, nsplit=self.nsplit, qkv_fused=self.qkv, cos_t=self.cos, sin_t=self.sin)
            ops.gemv(self.att, w["wo"], self.x, resid=self.x)
            ops.gemv(self.x, w["w1"], self.gate, w2=w["w3"], act="gelu_tanh", norm_w=w["n1"], norm_eps=c.rms_eps)
            ops.gemv(self.gate, w["w2"], self.x, resid=self.x)
        ops.layernorm_bf16_to_f32(self.x, self.ln_w, self.ln_b, self.lnf, c.ln_eps)
        ops.gemv(self.lnf, self.head, self.logits)

    def _sample_and_advance(self, guided, scale, temperature, generator):
        """lm_model.py:417-508: CFG / temperature / restriction / filters -> probabilities -> one multinomial draw -> forced-token
        schedule, token record, position advance.  torch.multinomial(p, 1, generator) is argmax(p / q) with
        q = empty_like(p).exponential_(1, generator) (ATen's own formula; its other ten launches are validity checks), so the
        draw is taken inside ld_llm_sample_advance from q drawn here with the same generator: the same token from the same
        generator state (tests/test_gpu_stages.py::test_fused_sampling_equals_torch_multinomial), in 2 launches instead of 16."""
        if self._x_from_tail:
            self.noise.exponential_(1.0, generator=generator)
            ops.llm_sample_advance(self.logits, self.probs, self.cfg_logits, guided, scale, temperature, self.pos, self.allowed,
                                   self.noise, self.forced, self.token, self.out_tokens, self.out_count, self.sampled, self.emb,
                                   self.x, top_k=self.top_k, top_p=self.top_p)
            return
        ops.llm_logits_to_probs(self.logits, self.probs, self.cfg_logits, guided, scale, temperature, self.pos, self.allowed,
                                top_k=self.top_k, top_p=self.top_p)
        torch.multinomial(self.probs, num_samples=1, generator=generator, out=self.sampled)
        ops.llm_decode_advance(self.sampled, self.forced, self.pos, self.token, self.out_tokens, self.out_count)

    # ---- decode loop -----------------------------------------------------------------------------
    @torch.no_grad()
    def sample(self, text_emb: torch.Tensor, *, motion_score: float = 0.1, num_frames: int = 13, guidance_scale: float = 7.5,
               temperature: float = 1.0, seed: int | None = None, generator=None, use_graph: bool = False,
               teacher_fed=None, logits_log=None, top_k: int | None = None, top_p: float | None = None,
               first_frame_tokens: torch.Tensor | None = None, on_segment=None, segment_tokens: int | None = None,
               mode: str | None = None) -> torch.Tensor:
        """Returns the clamped visual token ids, int64 [n_visual] on the device (lm_model.py:509-516).
        top_k / top_p filter the unrestricted positions inside the sampling kernel (lm_model.py:441-447).
        first_frame_tokens (int64 [iframe_len], e.g. from TokenizerEncoder.encode_to_index): use_gt_first_frame of the
        reference (lm_model.py:332-352) -- the given I-frame tokens, END_OF_IFrame and the first START_OF_PFrame join the
        prefilled prefix, sampling (and the RNG stream) starts at the first P token, the result begins with the given ids.
        on_segment(s): called on the host right after the step that emits the last of every `segment_tokens` visual tokens
        has been QUEUED (tokens [s * segment_tokens, (s + 1) * segment_tokens) of self.out_tokens are then final in stream
        order) -- lets a streaming caller start on segment s while later segments are still being decoded."""
        c, dev = self.cfg, self.dev
        self.top_k, self.top_p = top_k, top_p
        # mode: the form of a decode step's blocks ("chain" / "chained" / "fused", see __init__; None: the runner's default);
        # an unsupported shape falls back to "chain"
        mode = self.decode_mode if mode is None else mode
        assert mode in ("chain", "chained", "fused"), mode
        if mode != "chain" and not _lib.has_variants():
            raise _lib.LandiffHipError(f"decode mode {mode!r} needs the variants build of the library (LD_BUILD_VARIANTS=1 "
                                       f"landiff_amd/csrc/build.sh, then LANDIFF_HIP_LIB={_lib.VARIANTS_LIB_PATH}); the shipped library "
                                       "has the per-operation chain only")
        if (mode == "fused" and not self.fused_supported) or (mode == "chained" and (not self.chained_supported or use_graph)):
            mode = "chain"
        self._mode = mode
        guided = guidance_scale > 0 and guidance_scale != 1
        # unguided (ARSampleCfg's dataclass default cfg=0.0, lm_model.py:311-319): the conditional row is independent of the
        # second row, so the resident two-row buffers are kept and the sampling kernel reads row 0 only.
        feats = self.prefix_features(text_emb, float(num_frames), motion_score)
        S = feats.shape[1] - 1
        full_len, forced, restricted, n_visual = forced_token_schedule(c, S, num_frames)
        if full_len > self.Lmax:
            raise ValueError(f"LLM decode of {num_frames} frames after {text_emb.shape[-2]} text tokens needs {full_len} positions; this runner "
                             f"was built for {self.Lmax} (max_text / max_frames of LLMRunner; the reference's T5 tokenizer truncates at 512)")
        S_last = S                                         # position of the last prefilled token
        if first_frame_tokens is not None:
            assert num_frames > 1 and first_frame_tokens.numel() == c.iframe_len, "first_frame_tokens: one I frame, more frames to sample"
            ids = torch.cat([first_frame_tokens.reshape(-1).to(dev, torch.int64),
                             torch.tensor([c.END_I, c.START_P], device=dev, dtype=torch.int64)])
            assert forced[S + 1 + c.iframe_len] == c.END_I and forced[S + 2 + c.iframe_len] == c.START_P
            feats = torch.cat([feats, self.emb[ids].to(BF)[None].expand(2, -1, -1)], 1)
            S_last = S + c.iframe_len + 2
            n_visual -= c.iframe_len
        ft = torch.full((self.Lmax + 2,), -1, dtype=torch.int32)
        al = torch.zeros(self.Lmax + 2, 4, dtype=torch.int32)
        for p, t in forced.items():
            ft[p] = t
        for p, ids in restricted.items():
            al[p, 0] = len(ids)
            al[p, 1:1 + len(ids)] = torch.tensor(ids, dtype=torch.int32)
        self.forced.copy_(ft); self.allowed.copy_(al)
        self._x_from_tail = self.fused_tail and teacher_fed is None          # (a teacher-fed token is written by the host: re-embed it)
        self.out_count.zero_()
        if generator is None and seed:
            generator = torch.Generator(device=dev)
            generator.manual_seed(seed)                     # lm_model.py:398-402
        emitted = 0
        def note_position(pos):            # the token of position pos has just been queued
            nonlocal emitted
            if on_segment is None or pos in forced:
                return
            emitted += 1
            if emitted % segment_tokens == 0:
                on_segment(emitted // segment_tokens - 1)
        assert on_segment is None or (segment_tokens and first_frame_tokens is None)
        self._prefill(feats)
        self.pos.fill_(S_last)
        self._sample_and_advance(guided, guidance_scale, temperature, generator)
        try:                               # from here on the host-side position is live: the finally below always retires it
            self._pos_host = S_last + 1
            if self._mode == "fused":
                self.fused_ctl.zero_()
            elif self._mode == "chained":
                if self._side is None:
                    self._side = torch.cuda.Stream(device=dev)
                    self._chain_ev = torch.cuda.Event()
                self.chain_ctl.zero_()
                self._chain_epoch = 0
                self._side.wait_stream(torch.cuda.current_stream(dev))      # the prefill's KV cache and the zeroed counters
            note_position(S_last + 1)
            if logits_log is not None:
                logits_log.append(self.cfg_logits.clone())
            steps = full_len - (S_last + 1) - 1
            debug = teacher_fed is not None or logits_log is not None
            graph = None
            if use_graph and not debug and steps > 4:
                graph = self._capture(guided, guidance_scale, temperature, generator)
            t_enq = time.perf_counter()
            for it in range(steps):
                if teacher_fed is not None:
                    self.token.copy_(teacher_fed[it].reshape(1))
                if graph is not None:
                    graph.replay()
                else:
                    self._decode_forward()
                    self._sample_and_advance(guided, guidance_scale, temperature, generator)
                    self._pos_host += 1
                note_position(S_last + 2 + it)
                if logits_log is not None:
                    logits_log.append(self.cfg_logits.clone())
                if self._mode != "chain" and (it & 63) == 63:
                    self._raise_on_wait_timeout()          # opt-in forms only: one host sync per 64 steps instead of ~1244 steps on garbage
        finally:
            self._pos_host = -1            # any other caller of _decode_forward gets the device-side position
        self.host_enqueue_s = time.perf_counter() - t_enq      # host time to enqueue the loop (< wall time when the GPU is the bound)
        self._raise_on_wait_timeout()
        assert int(self.out_count.item()) == n_visual, (int(self.out_count.item()), n_visual)
        out = self.out_tokens[:n_visual]
        if first_frame_tokens is not None:
            out = torch.cat([first_frame_tokens.reshape(-1).to(dev, torch.int64), out])
        return out.clamp(0, c.visual_vocab - 1)

    def _raise_on_wait_timeout(self):
        """The 'fused' and 'chained' forms spin-wait on other workgroups without a cooperative launch; a wait that gives up sets a
        device flag.  After that the barrier counters are out of step and the KV cache holds rows computed from stale inputs: the
        runner must not be used for another decode step before the next sample() (which re-zeroes the control words and prefills)."""
        if (self._mode == "fused" and int(self.fused_ctl[1].item()) != 0) or (self._mode == "chained" and int(self.chain_ctl[0].item()) != 0):
            raise RuntimeError(f"LLM decode ({self._mode}): a device-side wait timed out (the decode did not have the GPU to itself?); "
                               "the KV cache of this decode is invalid; rerun with mode='chain' / LD_LLM_DECODE=chain")

    def _capture(self, guided, scale, temperature, generator):
        """Capture one decode step (forward + sampling + advance) into a HIP graph."""
        g = torch.cuda.CUDAGraph()
        if generator is not None:
            g.register_generator_state(generator)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        self._capturing = True
        try:
            with torch.cuda.stream(s):
                with torch.cuda.graph(g, stream=s):
                    self._decode_forward()
                    self._sample_and_advance(guided, scale, temperature, generator)
        finally:
            self._capturing = False
        torch.cuda.current_stream().wait_stream(s)
        return g

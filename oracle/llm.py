"""Oracle: AR semantic-token decode (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Follows landiff/llm/models/lm_model.py:175-516 (tokenize, sample), landiff/llm/models/transformer.py:91-119
(GPT.sample), landiff/llm/modules/transformer_blocks.py:22-40,67-88,128-236 (RMSNorm, LlamaMLP2, KV-cache
block), landiff/llm/modules/conditioner.py:90-170,230-323 (MicroConditioner, TextCond),
landiff/modules/pos_emb.py:16-123 (RoPE 1D).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .common import layer_norm, linear, timestep_embedding


def rope_table(dim: int, n: int, theta: float = 10000.0):
    """pos_emb.py:49-70: cis(t * theta^(-2i/dim)) as (cos, sin) fp32 [n, dim/2]."""
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
    t = torch.arange(n).float()
    f = torch.outer(t, freqs).float()
    cis = torch.polar(torch.ones_like(f), f)      # same op as the reference (bit-identical table)
    return cis.real.contiguous(), cis.imag.contiguous()


def apply_rope(x, cos, sin):
    """pos_emb.py:16-46: interleaved-pair complex rotation in fp32, cast back.
    x [..., heads, D]; cos/sin [..., D/2] (broadcast over heads)."""
    xf = x.float().reshape(*x.shape[:-1], -1, 2)
    a, b = xf[..., 0], xf[..., 1]
    c, s = cos.unsqueeze(-2), sin.unsqueeze(-2)
    out = torch.stack([a * c - b * s, a * s + b * c], dim=-1).flatten(-2)
    return out.type_as(x)


def rmsnorm(x, w, eps):
    """transformer_blocks.py:22-40 (fp32 inside, cast back to x.dtype)."""
    xf = x.float()
    out = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return (out * w.float()).type_as(x)


def forced_schedule(cfg, S: int, num_frames: int):
    """Position bookkeeping of Semantic1DLM.sample (lm_model.py:323-396) with END tokens enabled.

    S = index of the first START_OF_IFrame.  Returns (full_len, forced {pos: token id},
    restricted {pos: [allowed ids]}, n_visual)."""
    I, P, seg, stride = cfg.iframe_len, cfg.pframe_len, cfg.segment_length, cfg.segment_stride
    code_len = 0
    for off in range(0, num_frames, stride):       # video_frames_to_code_len :278-291
        fl = min(off + seg, num_frames) - off
        code_len += I + (fl - 1) * P + 2 * fl
    full_len = S + code_len + 1
    block = I + (seg - 1) * P + seg * 2
    start_i, end_i, start_p, end_p, eos_ok = set(), set(), set(), set(), set()
    n_i = n_p = 0
    for index in range(S, full_len - 1, block):
        mv = index
        start_i.add(mv)
        mv += 1
        n_i += I
        mv += I
        end_i.add(mv)
        mv += 1
        if index > S:
            eos_ok.add(mv)
        p_end = min(full_len - 1, mv - 1 + P * (seg - 1) + 2 * (seg - 1))
        for j in range(mv, p_end, P + 2):
            start_p.add(j)
            mv += 1
            n_p += P
            mv += P
            end_p.add(j + P + 1)
            mv += 1
            if index > S:
                eos_ok.add(mv)
    forced, restricted = {}, {}
    for i in range(S + 1, full_len):
        allowed = []
        if i in start_i: allowed.append(cfg.START_I)
        if i in start_p: allowed.append(cfg.START_P)
        if i in eos_ok: allowed.append(cfg.EOS)
        if allowed:
            restricted[i] = allowed
        # elif-chain of lm_model.py:464-505
        if i in start_i: forced[i] = cfg.START_I
        elif i in end_i: forced[i] = cfg.END_I
        elif i in start_p: forced[i] = cfg.START_P
        elif i in end_p: forced[i] = cfg.END_P
        elif i == full_len - 1: forced[i] = cfg.EOS
    # the reference's iframe/pframe sets count tokens; clip P tokens to what fits
    n_visual = sum(1 for i in range(S + 1, full_len) if i not in forced)
    return full_len, forced, restricted, n_visual


class LLMOracle:
    def __init__(self, state: dict, cfg, dtype=torch.bfloat16):
        self.s, self.cfg, self.dtype = state, cfg, dtype

    # ---- conditioning (tokenize) ----
    def text_cond(self, text_emb):
        """TextCond.forward_with_precomputed_embedding (conditioner.py:279-307): MLP2 with GELU-tanh."""
        s, dt = self.s, self.dtype
        x = text_emb.to(dt)
        x = linear(x, s["cond_model.embeddings.fc0.weight"], s["cond_model.embeddings.fc0.bias"], dt)
        x = F.gelu(x, approximate="tanh")
        return linear(x, s["cond_model.embeddings.fc1.weight"], s["cond_model.embeddings.fc1.bias"], dt)

    def micro_cond(self, frames: float, motion_score: float):
        """MicroConditioner.forward (conditioner.py:90-170), keys sorted: frames, motion_score."""
        s, dt, c = self.s, self.dtype, self.cfg
        outs = []
        for key, val in (("frames", frames), ("motion_score", motion_score)):
            emb = timestep_embedding(torch.full((1,), float(val)), c.freq_dim).to(dt)
            p = f"micro_condition.mlps.{key}."
            h = linear(emb, s[p + "0.weight"], s[p + "0.bias"], dt)
            h = F.silu(h)
            outs.append(linear(h, s[p + "2.weight"], s[p + "2.bias"], dt))
        return torch.cat(outs, dim=0)  # [2, hidden]

    def prefix_features(self, text_emb, frames, motion_score, with_guidance=True):
        """Semantic1DLM.tokenize (lm_model.py:175-276): [BOS][frames][motion][text x n][START_I];
        batch order [cond, uncond] (:195-200)."""
        s, c = self.s, self.cfg
        emb = s["visual_embedding_model.tok_emb_code.weight"]
        cond = self.text_cond(text_emb)
        micro = self.micro_cond(frames, motion_score)
        seqs = [cond]
        if with_guidance:
            null = s["cond_model.null_text_embedding"].to(self.dtype)
            seqs.append(null[None].expand(cond.shape[0], -1))
        feats = []
        for t in seqs:
            # mixed fp32 embedding rows + bf16 conditions concatenate to fp32 (torch promotion)
            feats.append(torch.cat([emb[c.BOS][None].float(), micro.float(), t.float(), emb[c.START_I][None].float()], 0))
        return torch.stack(feats, 0)  # [B, n+4, hidden] fp32

    # ---- transformer ----
    def block(self, i, x, cache, cos, sin):
        """TransformerBlock.forward + local_kvcache_inference (transformer_blocks.py:128-236)."""
        s, c, dt = self.s, self.cfg, self.dtype
        p = f"transformer.blocks.{i}."
        B, m, _ = x.shape
        h = rmsnorm(x, s[p + "norm0.weight"], c.rms_eps)
        qkv = linear(h, s[p + "wqkv.weight"], None, dt).view(B, m, 3, c.heads, c.head_dim)
        q, k, v = qkv.unbind(2)
        q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
        if cache[i] is not None:
            k = torch.cat([cache[i][0], k], 1)
            v = torch.cat([cache[i][1], v], 1)
        cache[i] = (k, v)
        scores = torch.einsum("nqhd,nkhd->nhqk", q, k) / (c.head_dim ** 0.5)
        if m > 1:
            mask = torch.triu(torch.ones_like(scores, dtype=torch.bool), diagonal=1)
            scores = scores.masked_fill(mask, -torch.finfo(scores.dtype).max)
        attn = F.softmax(scores.float(), dim=-1).to(scores.dtype)
        out = torch.einsum("nhql,nlhd->nqhd", attn, v).flatten(2)
        x = x + linear(out, s[p + "wo.weight"], None, dt)
        h = rmsnorm(x, s[p + "norm1.weight"], c.rms_eps)
        a = F.gelu(linear(h, s[p + "mlp.w1.weight"], None, dt), approximate="tanh") * linear(h, s[p + "mlp.w3.weight"], None, dt)
        return x + linear(a, s[p + "mlp.w2.weight"], None, dt)

    def gpt_step(self, feats, cache, cos, sin):
        """GPT.sample (transformer.py:91-119): blocks in fwd dtype, fp32 LayerNorm + fp32 head on the last token."""
        s, c = self.s, self.cfg
        x = feats.to(self.dtype)
        for i in range(c.num_layers):
            x = self.block(i, x, cache, cos, sin)
        x = layer_norm(x.float(), s["transformer.layer_norm.weight"], s["transformer.layer_norm.bias"], c.ln_eps)
        x = x[:, -1].contiguous()
        return F.linear(x, s["transformer.head.weight"].float())

    # ---- decode loop ----
    @torch.no_grad()
    def sample(self, text_emb, *, motion_score=0.1, num_frames=13, guidance_scale=7.5, temperature=1.0,
               top_k=None, top_p=None, multinomial_fn=None, generator=None, teacher_tokens=None,
               return_logits=False, first_frame_tokens=None):
        """Semantic1DLM.sample (lm_model.py:293-516).  `multinomial_fn(probs[1,V]) -> LongTensor[1,1]`
        lets a test draw from the same RNG stream as the device under test; `teacher_tokens` (full
        token fed back after every loop iteration, forced positions included) overrides the fed-back
        token while still recording what was sampled."""
        c = self.cfg
        with_guidance = guidance_scale > 0 and guidance_scale != 1
        feats = self.prefix_features(text_emb, float(num_frames), motion_score, with_guidance)
        S = feats.shape[1] - 1
        full_len, forced, restricted, n_visual = forced_schedule(c, S, num_frames)
        cos_all, sin_all = rope_table(c.head_dim, full_len, c.rope_theta)
        if multinomial_fn is None:
            multinomial_fn = lambda p: torch.multinomial(p, num_samples=1, generator=generator)
        emb = self.s["visual_embedding_model.tok_emb_code.weight"]
        cache = [None] * c.num_layers
        prefix_len = S + 1
        last = None
        sampled, all_logits = [], []
        if first_frame_tokens is not None:
            # use_gt_first_frame (lm_model.py:332-352): the prefix runs through [I tokens][END_OF_IFrame][START_OF_PFrame]
            # (prefix_len = start_of_visual + 1) and the returned codes start with the given I tokens
            ids = torch.cat([first_frame_tokens.reshape(-1).long(), torch.tensor([c.END_I, c.START_P])])
            extra = emb[ids].to(feats.dtype)[None].expand(feats.shape[0], -1, -1)
            feats = torch.cat([feats, extra], 1)
            prefix_len = S + 1 + c.iframe_len + 2
            sampled.append(first_frame_tokens.reshape(1, -1).long())
        for i in range(prefix_len, full_len):
            if last is not None:
                f = emb[last].float()  # [1,1,hidden]
                feats = torch.cat([f, f], 0) if with_guidance else f
                cs, sn = cos_all[None, i - 1:i], sin_all[None, i - 1:i]
            else:
                cs, sn = cos_all[None, :prefix_len], sin_all[None, :prefix_len]
            logits = self.gpt_step(feats, cache, cs, sn).float()
            if with_guidance:
                lc, lu = logits[:1], logits[1:]
                logits = lu + guidance_scale * (lc - lu)
            if return_logits:
                all_logits.append(logits.clone())
            logits = logits / temperature
            if i not in restricted:
                if top_k is not None:
                    v, _ = torch.topk(logits, top_k)
                    logits = logits.masked_fill(logits < v[:, [-1]], -float("inf"))
                probs = F.softmax(logits, dim=-1)
                if top_p is not None:
                    probs = top_p_probability(top_p, probs)
            else:
                mask = torch.full_like(logits, -float("inf"))
                mask[0, restricted[i]] = 0
                probs = F.softmax(logits + mask, dim=-1)
            last = multinomial_fn(probs)        # RNG is consumed at every step, forced or not
            if i in forced:
                last = torch.tensor([[forced[i]]], dtype=torch.long)
            else:
                sampled.append(last)
            if teacher_tokens is not None:
                last = teacher_tokens[i - prefix_len].reshape(1, 1)
        codes = torch.cat(sampled, dim=1)
        assert codes.shape[1] == n_visual, (codes.shape, n_visual)
        codes = codes.clamp(min=0, max=c.visual_vocab - 1)   # lm_model.py:515
        if return_logits:
            return codes, torch.cat(all_logits, 0)
        return codes


def top_p_probability(top_p: float, probs):
    """landiff/utils.py:345-359."""
    sp, si = torch.sort(probs, dim=-1, descending=True)
    cum = torch.cumsum(sp, dim=-1)
    rem = cum >= top_p
    rem[..., 1:] = rem[..., :-1].clone()
    rem[..., 0] = False
    to_remove = rem.scatter(-1, si, rem)
    probs = probs.masked_fill(to_remove, 0.0)
    return probs / probs.sum(-1, keepdim=True)

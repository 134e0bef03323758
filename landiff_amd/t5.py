"""T5 encoders on the MI355X kernels: FLAN-T5-XXL states for the LLM condition
(landiff/llm/modules/text_encoder.py:16-146) and T5-v1.1-XXL states padded to 226 tokens for the DiT context
(landiff/diffusion/sgm/modules/encoders/modules.py:246-295).  Both are HF transformers `T5EncoderModel` in bf16 in the
reference; this runner consumes the same state dict (`encoder.block.{i}.layer.{0,1}...`, `shared.weight`) and restates
transformers/models/t5/modeling_t5.py (pinned 4.47.1 in the reference's uv.lock):

    x = shared[ids]
    per block:  n = T5LayerNorm(x);  q,k,v = n Wq^T, n Wk^T, n Wv^T  (no bias, no 1/sqrt(d) scaling)
                s = q k^T + bias[bucket(j - i)]  (bias table of block 0, shared by all blocks);  p = softmax_fp32(s)
                x = x + (p v) Wo^T
                n = T5LayerNorm(x);  x = x + (gelu_new(n Wi0^T) * (n Wi1^T)) Wo2^T        (gated-gelu, v1.1 / FLAN)
    out = T5LayerNorm_final(x)

One prompt at a time, without padding: identical to the batched, masked HF call for the kept positions (masked keys get
-inf there).  The DiT path pads to 226 tokens and does NOT mask (FrozenT5Embedder.forward passes no attention_mask) --
`encode(ids)` on the padded ids reproduces that.  Tokenisation stays with sentencepiece / HF T5Tokenizer on the host.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

from . import ops

BF = torch.bfloat16


@dataclass(frozen=True)
class T5Config:
    vocab: int = 32128
    d_model: int = 4096
    d_kv: int = 64
    heads: int = 64
    d_ff: int = 10240
    layers: int = 24
    num_buckets: int = 32
    max_distance: int = 128
    eps: float = 1e-6

    @staticmethod
    def tiny() -> "T5Config":
        return T5Config(vocab=100, d_model=256, d_kv=64, heads=4, d_ff=512, layers=2)

    def hf(self):
        """The equivalent transformers config (tests / weight loading)."""
        from transformers import T5Config as HF
        return HF(vocab_size=self.vocab, d_model=self.d_model, d_kv=self.d_kv, d_ff=self.d_ff, num_layers=self.layers,
                  num_heads=self.heads, relative_attention_num_buckets=self.num_buckets,
                  relative_attention_max_distance=self.max_distance, feed_forward_proj="gated-gelu",
                  layer_norm_epsilon=self.eps, dropout_rate=0.0)


def relative_buckets(n: int, num_buckets: int = 32, max_distance: int = 128) -> torch.Tensor:
    """T5Attention._relative_position_bucket (bidirectional) for relative positions -(n-1) .. n-1 -> int32 [2n-1];
    entry (j - i + n - 1) is the bucket of key j seen from query i."""
    rel = torch.arange(-(n - 1), n, dtype=torch.long)
    nb = num_buckets // 2
    ret = (rel > 0).long() * nb
    a = rel.abs()
    max_exact = nb // 2
    is_small = a < max_exact
    large = max_exact + (torch.log(a.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return (ret + torch.where(is_small, a, large)).to(torch.int32)


def random_state(cfg: T5Config, seed: int, device="cpu") -> dict:
    """Seeded random weights in the HF T5EncoderModel layout at cfg's shapes (bench.py --with-t5, tools/t5_time.py: there is no
    network for the released encoders): projections ~ N(0, 0.02), gains 1, embedding and bias table ~ N(0, 1)."""
    g = torch.Generator(device=device).manual_seed(seed)
    rnd = lambda *shape, sc=0.02: (torch.randn(*shape, device=device, generator=g) * sc).to(BF)
    inner = cfg.heads * cfg.d_kv
    sd = {"shared.weight": rnd(cfg.vocab, cfg.d_model, sc=1.0), "encoder.final_layer_norm.weight": torch.ones(cfg.d_model, device=device),
          "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight": rnd(cfg.num_buckets, cfg.heads, sc=1.0)}
    for i in range(cfg.layers):
        p = f"encoder.block.{i}.layer."
        for nm, shp in (("0.SelfAttention.q.weight", (inner, cfg.d_model)), ("0.SelfAttention.k.weight", (inner, cfg.d_model)),
                        ("0.SelfAttention.v.weight", (inner, cfg.d_model)), ("0.SelfAttention.o.weight", (cfg.d_model, inner)),
                        ("1.DenseReluDense.wi_0.weight", (cfg.d_ff, cfg.d_model)), ("1.DenseReluDense.wi_1.weight", (cfg.d_ff, cfg.d_model)),
                        ("1.DenseReluDense.wo.weight", (cfg.d_model, cfg.d_ff))):
            sd[p + nm] = rnd(*shp)
        sd[p + "0.layer_norm.weight"] = torch.ones(cfg.d_model, device=device)
        sd[p + "1.layer_norm.weight"] = torch.ones(cfg.d_model, device=device)
    return sd


class T5EncoderRunner:
    def __init__(self, sd: dict, cfg: T5Config, device):
        assert cfg.d_kv == 64, "ld_t5_attn is written for head_dim 64 (T5-XXL / FLAN-T5-XXL)"
        self.cfg, self.dev = cfg, torch.device(device)
        g = lambda k: sd[k].detach().to(device=self.dev, dtype=BF).contiguous()
        self.emb = g("shared.weight")
        self.blocks = []
        for i in range(cfg.layers):
            p = f"encoder.block.{i}.layer."
            a = p + "0.SelfAttention."
            self.blocks.append(dict(
                ln0=g(p + "0.layer_norm.weight"),
                wqkv=torch.cat([g(a + "q.weight"), g(a + "k.weight"), g(a + "v.weight")], dim=0).contiguous(),  # one GEMM
                wo=g(a + "o.weight"), ln1=g(p + "1.layer_norm.weight"),
                wi0=g(p + "1.DenseReluDense.wi_0.weight"), wi1=g(p + "1.DenseReluDense.wi_1.weight"),
                wo2=g(p + "1.DenseReluDense.wo.weight")))
        self.bias_table = g("encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight")   # [buckets, H]
        self.final_ln = g("encoder.final_layer_norm.weight")
        self._buckets = {}

    def _bucket(self, n):
        if n not in self._buckets:
            self._buckets[n] = relative_buckets(n, self.cfg.num_buckets, self.cfg.max_distance).to(self.dev)
        return self._buckets[n]

    @torch.no_grad()
    def encode(self, ids: torch.Tensor) -> torch.Tensor:
        """ids int64 [n] (n <= 512) -> last_hidden_state [n, d_model] bf16."""
        c = self.cfg
        n = int(ids.numel())
        inner = c.heads * c.d_kv
        x = self.emb.index_select(0, ids.to(self.dev).reshape(-1))           # embedding gather (plumbing)
        nrm = torch.empty_like(x)
        qkv = torch.empty(n, 3 * inner, device=self.dev, dtype=BF)
        att = torch.empty(n, 3 * inner, device=self.dev, dtype=BF)[:, :inner]    # same row stride as the q/k/v views
        lin = torch.empty(n, c.d_ff, device=self.dev, dtype=BF)
        gate = torch.empty(n, c.d_ff, device=self.dev, dtype=BF)
        bucket = self._bucket(n)
        for b in self.blocks:
            ops.t5_rmsnorm(x, b["ln0"], nrm, c.eps)
            ops.gemm(nrm, b["wqkv"], out=qkv)
            ops.t5_attn(qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:], att, self.bias_table, bucket, c.heads)
            ops.gemm(att, b["wo"], out=x, resid=x)
            ops.t5_rmsnorm(x, b["ln1"], nrm, c.eps)
            ops.gemm(nrm, b["wi1"], out=lin)
            ops.gemm(nrm, b["wi0"], out=gate, act="gelu_tanh", mul=lin)       # gelu_new(wi_0 x) * (wi_1 x)
            ops.gemm(gate, b["wo2"], out=x, resid=x)
        out = torch.empty_like(x)
        ops.t5_rmsnorm(x, self.final_ln, out, c.eps)
        return out

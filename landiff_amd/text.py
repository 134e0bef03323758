"""Text encoders at the boundary: FLAN-T5-XXL for the LLM condition
(landiff/llm/modules/text_encoder.py:16-146) and T5-v1.1-XXL padded to 226 tokens for the DiT
(landiff/diffusion/sgm/modules/encoders/modules.py:246-295).  Tokenisation is HF T5Tokenizer (sentencepiece, host);
the encoder stacks run on the MI355X kernels through landiff_amd.t5.T5EncoderRunner, fed from the HF checkpoint's state
dict.  The hot path proper starts at their outputs (PromptInputs)."""
from __future__ import annotations

import functools

import torch

from .t5 import T5Config, T5EncoderRunner


@functools.lru_cache(maxsize=2)
def _load(name_or_dir: str, device_str: str):
    from transformers import T5EncoderModel, T5Tokenizer
    tok = T5Tokenizer.from_pretrained(name_or_dir)
    hf = T5EncoderModel.from_pretrained(name_or_dir, torch_dtype=torch.bfloat16)     # weights only: no HF forward is run
    c = hf.config
    assert c.is_gated_act and c.dense_act_fn == "gelu_new", "T5 v1.1 / FLAN-T5 (gated-gelu) expected"
    cfg = T5Config(vocab=c.vocab_size, d_model=c.d_model, d_kv=c.d_kv, heads=c.num_heads, d_ff=c.d_ff, layers=c.num_layers,
                   num_buckets=c.relative_attention_num_buckets, max_distance=c.relative_attention_max_distance,
                   eps=c.layer_norm_epsilon)
    run = T5EncoderRunner(hf.state_dict(), cfg, device_str)
    del hf
    return tok, run


@torch.no_grad()
def encode_flan_t5(prompts: list[str], device, max_length: int = 512, model_path: str = "google/flan-t5-xxl") -> list[torch.Tensor]:
    """-> list of [n_i, 4096] (padding removed), as FlanT5XXL.encode_texts_padded + TextCond(padding=False).  Each prompt
    is encoded on its own, unpadded: the same states the padded, masked HF batch yields at the kept positions.
    model_path: FlanT5XXL's `model_path` (text_encoder.py:137-146) -- the hub name, or a local directory."""
    tok, run = _load(model_path, str(device))
    out = []
    for p in prompts:
        ids = tok(p, return_tensors="pt", truncation=True, max_length=max_length).input_ids[0]
        out.append(run.encode(ids))
    return out


@torch.no_grad()
def encode_t5_v11(prompts: list[str], model_dir: str, max_length: int, device) -> torch.Tensor:
    """-> [B, max_length, 4096] padded to max_length; like FrozenT5Embedder.forward the pad positions are NOT masked."""
    tok, run = _load(model_dir, str(device))
    batch = tok(prompts, truncation=True, max_length=max_length, return_length=True, return_overflowing_tokens=False,
                padding="max_length", return_tensors="pt")
    return torch.stack([run.encode(batch["input_ids"][i]) for i in range(len(prompts))], dim=0)

"""Text encoders at the boundary (OUTSIDE the hot path): FLAN-T5-XXL for the LLM condition
(landiff/llm/modules/text_encoder.py:16-146) and T5-v1.1-XXL padded to 226 tokens for the DiT
(landiff/diffusion/sgm/modules/encoders/modules.py:246-295).  They run through HF transformers on PyTorch-ROCm; the hot
path starts at their outputs (PromptInputs)."""
from __future__ import annotations

import functools

import torch


@functools.lru_cache(maxsize=2)
def _load(name_or_dir: str, device_str: str):
    from transformers import T5EncoderModel, T5Tokenizer
    tok = T5Tokenizer.from_pretrained(name_or_dir)
    enc = T5EncoderModel.from_pretrained(name_or_dir, torch_dtype=torch.bfloat16).to(device_str).eval()
    return tok, enc


@torch.no_grad()
def encode_flan_t5(prompts: list[str], device, max_length: int = 512) -> list[torch.Tensor]:
    """-> list of [n_i, 4096] (padding removed), as FlanT5XXL.encode_texts_padded + TextCond(padding=False)."""
    tok, enc = _load("google/flan-t5-xxl", str(device))
    batch = tok(prompts, return_tensors="pt", padding=True, truncation=True, max_length=max_length).to(device)
    out = enc(input_ids=batch.input_ids, attention_mask=batch.attention_mask).last_hidden_state
    return [out[i, batch.attention_mask[i].bool()] for i in range(len(prompts))]


@torch.no_grad()
def encode_t5_v11(prompts: list[str], model_dir: str, max_length: int, device) -> torch.Tensor:
    """-> [B, max_length, 4096] padded to max_length (FrozenT5Embedder.forward)."""
    tok, enc = _load(model_dir, str(device))
    batch = tok(prompts, truncation=True, max_length=max_length, return_length=True, return_overflowing_tokens=False,
                padding="max_length", return_tensors="pt")
    return enc(input_ids=batch["input_ids"].to(device)).last_hidden_state

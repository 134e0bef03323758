"""landiff/llm/llm_cfg.py:18-81: the shipped LLM configuration -- 24 LlamaTransformerBlocks (16 heads, 2048 wide, MLP 11008,
GELU-tanh gate), vocabulary 2048 + 7, Rope1DPosEmb(dim 128, theta 10000, max_len 32768), I frame 330 / P frame 74 tokens,
TextCond over FLAN-T5-XXL (<= 512 tokens), MicroConditioner(frames, motion_score; frequency embedding 256) -- as a plain
dataclass instead of a fiddle graph."""
from landiff_amd.config import LLMConfig


def build_llm() -> LLMConfig:
    return LLMConfig()

"""landiff/llm/llm_cfg.py:18-81: the shipped LLM configuration (a plain dataclass here instead of a fiddle graph)."""
from landiff_amd.config import LLMConfig


def build_llm() -> LLMConfig:
    return LLMConfig()

"""Task API of landiff/llm/llm_infer.py:13-105 on the MI355X path."""
from __future__ import annotations

from dataclasses import dataclass, field, fields
from pathlib import Path

import torch

from landiff_amd.config import LLMConfig
from landiff_amd.llm import LLMRunner
from landiff_amd.text import encode_flan_t5
from landiff_amd.weights import load_llm_state


@dataclass
class ARSampleCfg:
    top_k: int | None = None
    top_p: float | None = None
    temperature: float = 1.0
    teacher_forcing: bool = False
    use_gt_first_frame: bool = False
    cfg: float = 0.0
    motion_score: float | None = None
    num_frames: int = 13  # 49 RGB frames, 13 semantic frames

    def to_dict(self):
        return {f.name: getattr(self, f.name) for f in fields(self) if getattr(self, f.name) != f.default}

    def __str__(self):
        d = self.to_dict()
        return ",".join(f"{k}_{v}" for k, v in d.items()) or "default"


@dataclass
class CodeTask:
    save_file_name: str
    prompt: str
    seed: int
    result: None | torch.Tensor = None
    sample_cfg: ARSampleCfg = field(default_factory=ARSampleCfg)
    # use_gt_first_frame: semantic tokens of the conditioning video/first frame (LongTensor, at least the I frame), e.g. from
    # landiff_amd.tokenizer_encoder.TokenizerEncoder.encode_to_index on its Theia feature maps.  (The reference reaches the
    # same prefix through inputs["video"] of Semantic1DLM.tokenize, lm_model.py:315-352.)
    first_frame_tokens: None | torch.Tensor = None


class ArModelInferWrapper(torch.nn.Module):
    """ArModelInferWrapper(ckpt_path, model_cfg)(CodeTask) -> CodeTask with .result LongTensor[1218] on CPU."""

    def __init__(self, ckpt_path: str, model_cfg: LLMConfig, device="cuda"):
        super().__init__()
        assert Path(ckpt_path).exists(), f"ckpt_path: {ckpt_path} does not exist"
        assert Path(ckpt_path).suffix == ".safetensors", f"ckpt_path: {ckpt_path} is not a safetensors file"
        self.config = model_cfg
        self.device_ = torch.device(device if device != "cuda" else f"cuda:{torch.cuda.current_device()}")
        self.runner = LLMRunner(load_llm_state(ckpt_path), model_cfg, self.device_)

    @torch.no_grad()
    def forward(self, code_task: CodeTask) -> CodeTask:
        sc = code_task.sample_cfg
        if sc.teacher_forcing:
            raise NotImplementedError("teacher_forcing replays a ground-truth token stream (training-time check); not on the inference path")
        first = None
        if sc.use_gt_first_frame:
            if code_task.first_frame_tokens is None:
                raise ValueError("use_gt_first_frame needs CodeTask.first_frame_tokens (TokenizerEncoder.encode_to_index output)")
            first = code_task.first_frame_tokens.reshape(-1)[: self.config.iframe_len]
        text = encode_flan_t5([code_task.prompt], self.device_)[0]
        torch.manual_seed(code_task.seed)
        torch.cuda.manual_seed(code_task.seed)
        tokens = self.runner.sample(text, motion_score=sc.motion_score if sc.motion_score is not None else 0.0,
                                    num_frames=sc.num_frames, guidance_scale=sc.cfg, temperature=sc.temperature,
                                    seed=code_task.seed, top_k=sc.top_k, top_p=sc.top_p, first_frame_tokens=first)
        code_task.result = tokens.cpu().reshape(-1)
        return code_task

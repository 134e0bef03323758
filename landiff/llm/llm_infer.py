"""Task API of landiff/llm/llm_infer.py:13-105 on the MI355X path."""
from __future__ import annotations

from dataclasses import dataclass, field, fields
from pathlib import Path

import torch

from landiff.utils import set_seed_for_single_process
from landiff_amd.config import LLMConfig
from landiff_amd.llm import LLMRunner, forced_token_schedule
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

    def to_dict_str(self):
        return str(self.to_dict()).replace(" ", "")

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
    # Ground-truth semantic tokens of a conditioning / reference video (LongTensor [n_visual], e.g. from
    # landiff_amd.tokenizer_encoder.TokenizerEncoder.encode_to_index on Theia feature maps).  The reference reaches them through
    # inputs["video"] of Semantic1DLM.tokenize (lm_model.py:315-352), which its own wrapper never fills; here they are task fields:
    #   first_frame_tokens -- use_gt_first_frame: at least the I frame (lm_model.py:332-352);
    #   gt_tokens          -- teacher_forcing: the whole clip, fed back instead of the sampled ids (lm_model.py:506-507).
    first_frame_tokens: None | torch.Tensor = None
    gt_tokens: None | torch.Tensor = None


def teacher_sequence(cfg: LLMConfig, S: int, num_frames: int, gt_tokens: torch.Tensor, skip: int = 0) -> torch.Tensor:
    """The token fed at every decode position after the prefill when teacher forcing: position p of the reference's `token`
    sequence (lm_model.py:506-507 `last_code = token[i : i + 1]`) = the forced special id where the layout has one, else the next
    ground-truth visual token.  S: position of the first START_OF_IFrame; skip: leading positions already in the prefix
    (use_gt_first_frame: the I frame, END_OF_IFrame and the first START_OF_PFrame)."""
    full_len, forced, _, n_visual = forced_token_schedule(cfg, S, num_frames)
    gt = gt_tokens.reshape(-1).tolist()
    assert len(gt) == n_visual, f"gt_tokens has {len(gt)} ids, the layout of {num_frames} frames has {n_visual} visual slots"
    it = iter(gt)
    seq = [forced[i] if i in forced else next(it) for i in range(S + 1, full_len)]
    return torch.tensor(seq[skip:], dtype=torch.int64)


class ArModelInferWrapper(torch.nn.Module):
    """ArModelInferWrapper(ckpt_path, model_cfg)(CodeTask) -> CodeTask with .result LongTensor[n_visual] on CPU.
    model_cfg: the LLMConfig that stands for the reference's fiddle graph (landiff.llm.llm_cfg.build_llm()).
    text_encoder: optional callable prompts -> list of [n_i, text_dim] states replacing the FLAN-T5-XXL run (pre-computed
    embeddings, as `load_weights=False` does in the reference's text encoder, text_encoder.py:126-131)."""

    def __init__(self, ckpt_path: str, model_cfg: LLMConfig, device="cuda", text_encoder=None):
        super().__init__()
        assert Path(ckpt_path).exists(), f"ckpt_path: {ckpt_path} does not exist"
        assert Path(ckpt_path).suffix == ".safetensors", f"ckpt_path: {ckpt_path} is not a safetensors file"
        self.config = model_cfg
        self.device_ = torch.device(device if device != "cuda" else f"cuda:{torch.cuda.current_device()}")
        self.text_encoder = text_encoder
        sd = load_llm_state(ckpt_path)
        from landiff_amd.weights import llm_spec
        want = {n: tuple(sh) for n, sh, _ in llm_spec(model_cfg)}
        missing = sorted(set(want) - set(sd))
        assert not missing, f"load_state_dict(strict=True): missing keys {missing[:5]}{'...' if len(missing) > 5 else ''}"
        bad = [k for k in want if tuple(sd[k].shape) != want[k]]
        assert not bad, f"checkpoint / model_cfg shape mismatch at {bad[:3]}: {[tuple(sd[k].shape) for k in bad[:3]]} vs {[want[k] for k in bad[:3]]}"
        self.runner = LLMRunner(sd, model_cfg, self.device_, max_text=model_cfg.max_cond_tokens)

    @torch.no_grad()
    def forward(self, code_task: CodeTask) -> CodeTask:
        sc, c = code_task.sample_cfg, self.config
        first = None
        if sc.use_gt_first_frame:
            src = code_task.first_frame_tokens if code_task.first_frame_tokens is not None else code_task.gt_tokens
            if src is None:
                raise ValueError("use_gt_first_frame needs CodeTask.first_frame_tokens (TokenizerEncoder.encode_to_index output)")
            first = src.reshape(-1)[: c.iframe_len]
        if self.text_encoder is not None:
            text = self.text_encoder([code_task.prompt])[0]
        else:
            text = encode_flan_t5([code_task.prompt], self.device_, max_length=c.max_cond_tokens, model_path=c.text_encoder_path)[0]
        fed = None
        if sc.teacher_forcing:
            if code_task.gt_tokens is None:
                raise ValueError("teacher_forcing needs CodeTask.gt_tokens (the clip's ground-truth semantic tokens)")
            skip = c.iframe_len + 2 if first is not None else 0
            fed = teacher_sequence(c, text.shape[0] + 3, sc.num_frames, code_task.gt_tokens, skip).to(self.device_)
        if sc.motion_score is None:
            # the reference passes input["motion_score"] = None, which MicroConditioner rejects (conditioner.py:100-113: the key
            # is present, so the config default is not consulted, and no null embedding is configured)
            raise ValueError("Condition key motion_score not found in data, and a default is not given, and null default is not set.")
        set_seed_for_single_process(code_task.seed)
        tokens = self.runner.sample(text, motion_score=sc.motion_score,
                                    num_frames=sc.num_frames, guidance_scale=sc.cfg, temperature=sc.temperature,
                                    seed=code_task.seed, top_k=sc.top_k, top_p=sc.top_p, first_frame_tokens=first,
                                    teacher_fed=fed)
        code_task.result = tokens.cpu().reshape(-1)
        return code_task

"""Task API of landiff/diffusion/dif_infer.py:84-302 on the MI355X path."""
from __future__ import annotations

from dataclasses import dataclass

import torch

from landiff_amd.config import PipelineConfig
from landiff_amd.detokenizer import Detokenizer
from landiff_amd.dit import ControlDiTRunner
from landiff_amd.sampler import DiffusionSampler
from landiff_amd.text import encode_t5_v11
from landiff_amd.vae import VAEDecoder
from landiff_amd.weights import load_diffusion_states, resolve_ckpt_root


@dataclass
class CogOutput:
    video: torch.Tensor   # rgb video tensor, shape [B, C, T, H, W], range [0,1]
    latent: torch.Tensor  # latent tensor, shape [B, T, C, H, W]


@dataclass
class VideoTask:
    save_file_name: str
    prompt: str
    seed: int
    fps: int = 8
    mp4: None | torch.Tensor = None
    semantic_token: None | torch.Tensor = None
    result: None | torch.Tensor = None


class CogModelInferWrapper(torch.nn.Module):
    """CogModelInferWrapper(ckpt_path)(VideoTask) -> VideoTask with .result FloatTensor[3,49,480,720] in [0,1] on CPU.
    infer_cfg_path / model_cfg_path are accepted for signature compatibility; the shipped YAML values are restated in
    landiff_amd.config."""

    def __init__(self, ckpt_path: str, infer_cfg_path: str | None = None, model_cfg_path: str | None = None, device="cuda"):
        super().__init__()
        cfg = PipelineConfig.full().check()
        self.cfg = cfg
        self.device_ = torch.device(device if device != "cuda" else f"cuda:{torch.cuda.current_device()}")
        root = resolve_ckpt_root()
        st = load_diffusion_states(ckpt_path, root)
        self.detok = Detokenizer(st["tok"], st["ups"], cfg.tok, cfg.ups, self.device_)
        self.dit = ControlDiTRunner(st["dit_main"], st["dit_control"], cfg.dit, self.device_)
        self.sampler = DiffusionSampler(cfg.sampler)
        self.vae = VAEDecoder(st["vae"], cfg.vae, self.device_)
        self.t5_dir = f"{root}/CogVideoX-2b-sat/t5-v1_1-xxl"

    @torch.no_grad()
    def forward(self, x: VideoTask) -> VideoTask:
        if x.mp4 is not None:
            raise NotImplementedError("video-conditioned generation (mp4 input) needs the tokenizer encoder (SURVEY 8f rank 3)")
        d = self.cfg.dit
        ctx = encode_t5_v11([x.prompt], self.t5_dir, d.text_len, self.device_)
        torch.manual_seed(x.seed)
        torch.cuda.manual_seed(x.seed)
        sem = self.detok.semantic_condition(x.semantic_token.to(self.device_).reshape(-1))
        self.dit.set_condition(ctx, sem)
        noise = torch.randn(1, d.latent_frames, d.in_channels, d.latent_h, d.latent_w, device=self.device_, dtype=torch.float32)
        z = self.sampler.run(self.dit.step, noise)
        _, video = self.vae.decode(z.to(torch.bfloat16).float(), want_float=True)
        x.result = video.cpu()
        return x

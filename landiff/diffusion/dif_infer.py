"""Task API of landiff/diffusion/dif_infer.py:84-302 on the MI355X path: CogModelInferWrapper(ckpt_path, infer_cfg_path,
model_cfg_path)(VideoTask) and, one level down, CogWrapper.forward(inputs, seed, semantic_token,
semantic_feature_before_upsample, vae_feature_prefix) -> CogOutput.  Every shape comes from the two YAML files (PyYAML;
landiff_amd.config.load_diffusion_config), every weight from the reference's checkpoint tree."""
from __future__ import annotations

import os
from dataclasses import dataclass

import torch

from landiff.utils import set_seed_for_single_process, stable_hash
from landiff_amd.config import DiffusionInferConfig, load_diffusion_config
from landiff_amd.detokenizer import Detokenizer
from landiff_amd.dit import ControlDiTRunner
from landiff_amd.sampler import DiffusionSampler
from landiff_amd.text import encode_t5_v11
from landiff_amd.vae import VAEDecoder
from landiff_amd.weights import load_diffusion_states, load_tokenizer_encoder_state, resolve_ckpt_path

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_INFER_CFG = "landiff/diffusion/configs/infer_cfgs/2b.yaml"
DEFAULT_MODEL_CFG = "landiff/diffusion/configs/cogvideox_2b_control_theia_interpolate_video_vq.yaml"


def _cfg_path(path: str) -> str:
    """The reference opens its YAML paths relative to the working directory (the repository root); when run from elsewhere the
    copy shipped next to this file is used for the two default names."""
    if os.path.exists(path):
        return path
    cand = os.path.join(os.path.dirname(os.path.dirname(_HERE)), path)
    if os.path.exists(cand):
        return cand
    raise FileNotFoundError(f"config file {path!r} not found (cwd {os.getcwd()!r})")


def _pre_process_cog_video(video: torch.Tensor) -> torch.Tensor:
    """dif_infer.py:22-34: [0,1] -> [-1,1]."""
    return torch.clamp(video * 2.0 - 1.0, -1.0, 1.0)


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


class CogWrapper(torch.nn.Module):
    """dif_infer.py:101-271.  Built from the parsed configs instead of an argument string; `seed` is sat's --seed default (1234),
    used only when forward() gets no seed (then hashed with the prompt, :190-194)."""

    def __init__(self, cfg: DiffusionInferConfig, ckpt_path: str, device, seed: int = 1234, text_encoder=None,
                 feature_extractor=None):
        super().__init__()
        self.cfg, self.device_, self.seed = cfg, device, seed
        self.image_size = list(cfg.image_size)
        self.text_encoder, self.feature_extractor = text_encoder, feature_extractor
        st = load_diffusion_states(resolve_ckpt_path(ckpt_path), None, base_dit_ckpt=cfg.base_dit_ckpt, vae_ckpt=cfg.vae_ckpt,
                                   tokenizer_ckpt=cfg.tokenizer_ckpt or None)
        self.detok = Detokenizer(st["tok"], st["ups"], cfg.tok, cfg.ups, device)
        self.dit = ControlDiTRunner(st["dit_main"], st["dit_control"], cfg.dit, device)
        self.sampler = DiffusionSampler(cfg.sampler)
        self.vae = VAEDecoder(st["vae"], cfg.vae, device)
        self._encoder = None            # tokenizer encoder half: built on the first video-conditioned call

    def _context(self, text: str) -> torch.Tensor:
        """FrozenT5Embedder on the prompt, padded to text_length, pad positions not masked (encoders/modules.py:246-295); the
        unconditional branch is zeroed by force_uc_zero_embeddings=["txt"] (dif_infer.py:176,214-218), which
        ControlDiTRunner.set_condition does."""
        d = self.cfg.dit
        if self.text_encoder is not None:
            ctx = self.text_encoder([text])
            ctx = ctx[0] if isinstance(ctx, (list, tuple)) else ctx
            ctx = ctx.reshape(1, d.text_len, d.text_dim)
        else:
            ctx = encode_t5_v11([text], resolve_ckpt_path(self.cfg.t5_dir), d.text_len, self.device_)
        return ctx

    def _semantic_from_video(self, mp4_btchw: torch.Tensor) -> torch.Tensor:
        """Control signal from a conditioning video (ControlDiffusionTransformer.forward :960-975 -> SemanticCond.forward(visual)
        :112-137 -> VideoVQWrap.forward(images) vq_warp.py:88-118): t equally spaced frames, [-1,1] -> uint8, padded to a square
        with grey 127, Theia features -> tokenizer encoder -> nearest code -> tokenizer decoder -> upsampler.  The Theia backbone
        is a Hugging Face remote-code model that this package does not build: `feature_extractor` supplies it."""
        if self.feature_extractor is None:
            raise NotImplementedError(
                "VideoTask.mp4 (video-conditioned generation) needs the Theia feature extractor (landiff/tokenizer/models/"
                "feature_extractor/theia_extractor.py), a Hugging Face remote-code model that is not built here: pass "
                "CogModelInferWrapper(..., feature_extractor=callable uint8 [T,3,S,S] -> features [T,C,h,w]) or give semantic_token")
        d, tc = self.cfg.dit, self.cfg.tok
        v = mp4_btchw[0]
        idx = torch.linspace(0, v.shape[0] - 1, d.latent_frames).long().to(v.device)
        v = v[idx]
        v = ((v + 1.0) / 2.0).clamp(0, 1)
        # torchvision v2.functional.to_dtype(uint8, scale=True) on a float image: image.mul(255 + 1 - 1e-3).to(uint8), i.e.
        # truncation of x * 255.999 (torchvision is not in this image: restated from its float -> int conversion rule)
        v = v.float().mul(255.0 + 1.0 - 1e-3).to(torch.uint8)
        H, W = v.shape[-2:]
        S = max(H, W)
        sq = torch.full((v.shape[0], v.shape[1], S, S), 127, dtype=torch.uint8, device=v.device)
        sq[..., :H, :W] = v                                                # pad_to_square: right / bottom (condition.py:14-27)
        feats = self.feature_extractor(sq)
        assert feats.shape == (tc.temporal, tc.out_channels, tc.grid_h, tc.grid_w), f"feature_extractor returned {tuple(feats.shape)}"
        if self._encoder is None:
            from landiff_amd.tokenizer_encoder import TokenizerEncoder
            self._encoder = TokenizerEncoder(load_tokenizer_encoder_state(resolve_ckpt_path(self.cfg.tokenizer_ckpt)), tc, self.device_)
        tokens = self._encoder.encode_to_index(feats.to(self.device_))
        return self.detok.semantic_condition(tokens)

    @torch.no_grad()
    def forward(self, inputs: dict, seed: int | None = None, semantic_token: torch.Tensor | None = None,
                semantic_feature_before_upsample: torch.Tensor | None = None,
                vae_feature_prefix: torch.Tensor | None = None) -> CogOutput:
        d, dev = self.cfg.dit, self.device_
        text, mp4 = inputs["caption"], inputs.get("video")
        if mp4 is not None:
            mp4 = _pre_process_cog_video(mp4.permute(0, 2, 1, 3, 4))       # b c t h w -> b t c h w
        text_seed = seed if seed is not None else (stable_hash(str(text)) + self.seed) % 2 ** 32
        set_seed_for_single_process(text_seed)
        ctx = self._context(text)
        # the control DiT evaluates the semantic condition once, at the first denoiser call, and caches it
        # (InferValueRegistry, dit_video_concat.py:939-982); precedence token > feature-before-upsample > video as there
        if semantic_token is not None:
            sem = self.detok.semantic_condition(semantic_token.to(dev).reshape(-1))
        elif semantic_feature_before_upsample is not None:
            sem = self.detok.semantic_condition_from_features(semantic_feature_before_upsample.to(dev))
        elif mp4 is not None:
            sem = self._semantic_from_video(mp4.to(dev))
        else:
            raise KeyError("mp4")            # the reference's kwargs["mp4"] lookup with nothing registered
        self.dit.set_condition(ctx, sem)
        noise = torch.randn(1, d.latent_frames, d.in_channels, d.latent_h, d.latent_w, device=dev, dtype=torch.float32)
        prefix = vae_feature_prefix.to(dev, torch.float32) if vae_feature_prefix is not None else None
        z = self.sampler.run(self.dit.step, noise, prefix=prefix)
        z = z.to(torch.bfloat16) if self.cfg.bf16 else z                   # samples.to(self.dtype), diffusion_video.py:314
        _, video = self.vae.decode(z.float(), want_float=True)
        return CogOutput(video=video[None], latent=z)


class CogModelInferWrapper(torch.nn.Module):
    """CogModelInferWrapper(ckpt_path, infer_cfg_path, model_cfg_path)(VideoTask) -> VideoTask with .result FloatTensor
    [3, 4T-3, H, W] in [0,1] on CPU (dif_infer.py:274-302).
    text_encoder: optional callable prompts -> [1, text_length, 4096] T5 states (pre-computed embeddings) replacing the
    T5-v1.1-XXL run; feature_extractor: the Theia backbone for VideoTask.mp4 (see CogWrapper._semantic_from_video)."""

    def __init__(self, ckpt_path: str, infer_cfg_path: str = DEFAULT_INFER_CFG, model_cfg_path: str = DEFAULT_MODEL_CFG,
                 device="cuda", text_encoder=None, feature_extractor=None):
        super().__init__()
        self.infer_cfg_path, self.model_cfg_path, self.ckpt_path = infer_cfg_path, model_cfg_path, ckpt_path
        cfg = load_diffusion_config(_cfg_path(model_cfg_path), _cfg_path(infer_cfg_path))
        dev = torch.device(device if device != "cuda" else f"cuda:{torch.cuda.current_device()}")
        self.init_infer_model = CogWrapper(cfg, ckpt_path, dev, text_encoder=text_encoder, feature_extractor=feature_extractor)

    @torch.no_grad()
    def forward(self, x: VideoTask) -> VideoTask:
        inputs = dict(caption=x.prompt, video=x.mp4 if x.mp4 is not None else None)
        output = self.init_infer_model.forward(inputs, seed=x.seed, semantic_token=x.semantic_token)
        video = output.video
        assert video.shape[0] == 1, f"video.shape[0] != 1, {video.shape[0]}"
        x.result = video.cpu()[0]
        return x

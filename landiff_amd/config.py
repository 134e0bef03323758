"""Shape/config dataclasses of the LanDiff inference path.

Values restate the reference's two config systems (no fiddle / OmegaConf needed at run time):
  * LLM:        landiff/llm/llm_cfg.py:18-81
  * tokenizer:  landiff/tokenizer/tokenizer_cfg.py:29-112
  * DiT/VAE/sampler: landiff/diffusion/configs/cogvideox_2b_control_theia_interpolate_video_vq.yaml
  * inference:  landiff/diffusion/configs/infer_cfgs/2b.yaml
``tiny()`` variants are the random-init plumbing configs (BASELINE.json configs[0], tests).
"""
from __future__ import annotations

from dataclasses import dataclass, field, replace


@dataclass(frozen=True)
class LLMConfig:
    num_layers: int = 24
    hidden: int = 2048
    heads: int = 16
    mlp: int = 11008
    visual_vocab: int = 2048           # + 7 specials (lm_model.py:62-70)
    text_dim: int = 4096               # FLAN-T5-XXL width
    freq_dim: int = 256                # MicroConditioner frequency embedding
    micro_hidden: int = 512
    rope_theta: float = 10000.0
    rope_max_len: int = 32768
    rms_eps: float = 1e-5
    ln_eps: float = 1e-5
    iframe_len: int = 330
    pframe_len: int = 74
    segment_length: int = 13
    segment_stride: int = 13

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    @property
    def vocab(self) -> int:
        return self.visual_vocab + 7

    # special ids, in the order Semantic1DLM registers them (lm_model.py:62-70)
    @property
    def EOS(self): return self.visual_vocab
    @property
    def BOS(self): return self.visual_vocab + 1
    @property
    def START_I(self): return self.visual_vocab + 2
    @property
    def END_I(self): return self.visual_vocab + 3
    @property
    def START_P(self): return self.visual_vocab + 4
    @property
    def END_P(self): return self.visual_vocab + 5
    @property
    def PAD(self): return self.visual_vocab + 6

    @staticmethod
    def tiny() -> "LLMConfig":
        return LLMConfig(num_layers=2, hidden=256, heads=2, mlp=512, visual_vocab=64, text_dim=128,
                         freq_dim=64, micro_hidden=64, iframe_len=6, pframe_len=3, segment_length=3,
                         segment_stride=3)


@dataclass(frozen=True)
class TokenizerConfig:
    width: int = 768
    layers: int = 12
    heads: int = 12
    grid_h: int = 30
    grid_w: int = 45
    temporal: int = 13
    pframe_tokens: int = 74
    num_latent_tokens: int = 1218
    codebook_size: int = 2048
    codebook_dim: int = 16
    token_size: int = 768
    out_channels: int = 768
    rope_theta: float = 10000.0
    ln_eps: float = 1e-5
    # VideoVQ.norm_features / denorm_features act only when `mean_std_path is not None` (video_titok_vq.py:221-233); the
    # shipped tokenizer_cfg.py:107 sets only `mean_std_dim=768`, so the `mean`/`std` buffers exist in the checkpoint but both
    # functions are the identity.  True mirrors a config that names a mean_std_path.
    norm_features: bool = False

    @property
    def head_dim(self): return self.width // self.heads
    @property
    def tokens_per_frame(self): return self.grid_h * self.grid_w
    @property
    def iframe_tokens(self): return self.num_latent_tokens - (self.temporal - 1) * self.pframe_tokens
    @property
    def n_visual(self): return self.temporal * self.tokens_per_frame
    @property
    def seq_len(self): return self.n_visual + self.num_latent_tokens

    @staticmethod
    def tiny() -> "TokenizerConfig":
        # matches LLMConfig.tiny(): 3 frames, 6 I tokens + 2*3 P tokens
        return TokenizerConfig(width=128, layers=2, heads=2, grid_h=4, grid_w=6, temporal=3,
                               pframe_tokens=3, num_latent_tokens=12, codebook_size=64, codebook_dim=16,
                               token_size=128, out_channels=128)


@dataclass(frozen=True)
class UpsamplerConfig:
    """vq_gan_blocks.Decoder + SemanticCond.conv_out (yaml :58-75, condition.py:47-56)."""
    z_channels: int = 768
    ch: int = 512
    ch_mult: tuple = (0.25, 1)
    num_res_blocks: int = 4
    out_ch: int = 64
    target_dim: int = 16
    gn_groups: int = 32
    gn_eps: float = 1e-6

    @staticmethod
    def tiny() -> "UpsamplerConfig":
        # pixel-shuffled channels (ch/4) must stay a multiple of 64 for the conv kernel
        return UpsamplerConfig(z_channels=128, ch=256, ch_mult=(0.25, 1), num_res_blocks=1, out_ch=64,
                               target_dim=16)


@dataclass(frozen=True)
class DiTConfig:
    hidden: int = 1920
    heads: int = 30
    layers_main: int = 30
    layers_control: int = 15
    time_embed_dim: int = 512
    patch: int = 2
    in_channels: int = 16
    out_channels: int = 16
    latent_h: int = 60
    latent_w: int = 90
    latent_frames: int = 13
    text_len: int = 226
    text_dim: int = 4096
    block_ln_eps: float = 1e-5         # sat layernorm_epsilon default
    qk_ln_eps: float = 1e-6            # dit_video_concat.py:521-537
    final_ln_eps: float = 1e-6         # :428-430
    height_interpolation: float = 1.875
    width_interpolation: float = 1.875
    time_interpolation: float = 1.0

    @property
    def head_dim(self): return self.hidden // self.heads
    @property
    def grid_h(self): return self.latent_h // self.patch
    @property
    def grid_w(self): return self.latent_w // self.patch
    @property
    def n_img(self): return self.latent_frames * self.grid_h * self.grid_w
    @property
    def seq_len(self): return self.text_len + self.n_img

    @staticmethod
    def tiny() -> "DiTConfig":
        # goes with TokenizerConfig.tiny(): 3 latent frames, latent 8x12 (= 2x the 4x6 semantic grid)
        return DiTConfig(hidden=128, heads=2, layers_main=3, layers_control=2, time_embed_dim=64,
                         latent_h=8, latent_w=12, latent_frames=3, text_len=10, text_dim=64)

    @staticmethod
    def config0() -> "DiTConfig":
        """BASELINE.json configs[0]: tiny DiT, 8 latent frames, 64x64 latent."""
        return DiTConfig(hidden=128, heads=2, layers_main=2, layers_control=1, time_embed_dim=64,
                         latent_h=64, latent_w=64, latent_frames=8, text_len=8, text_dim=64)


@dataclass(frozen=True)
class VAEConfig:
    ch: int = 128
    ch_mult: tuple = (1, 2, 2, 4)
    num_res_blocks: int = 3
    z_channels: int = 16
    out_ch: int = 3
    temporal_compress_times: int = 4
    gn_groups: int = 32
    gn_eps: float = 1e-6
    scale_factor: float = 1.15258426

    @staticmethod
    def tiny() -> "VAEConfig":
        return VAEConfig(ch=64, ch_mult=(1, 2, 2, 4), num_res_blocks=1)


@dataclass(frozen=True)
class SamplerConfig:
    num_steps: int = 50
    cfg_scale: float = 6.0
    cfg_exp: float = 5.0
    shift_scale: float = 3.0
    num_idx: int = 1000
    linear_start: float = 0.00085
    linear_end: float = 0.0120
    sampler: str = "vpsde_dpmpp2m"     # or "ddim" (VideoDDIMSampler)


@dataclass(frozen=True)
class PipelineConfig:
    llm: LLMConfig = field(default_factory=LLMConfig)
    tok: TokenizerConfig = field(default_factory=TokenizerConfig)
    ups: UpsamplerConfig = field(default_factory=UpsamplerConfig)
    dit: DiTConfig = field(default_factory=DiTConfig)
    vae: VAEConfig = field(default_factory=VAEConfig)
    sampler: SamplerConfig = field(default_factory=SamplerConfig)

    @staticmethod
    def full() -> "PipelineConfig":
        return PipelineConfig()

    @staticmethod
    def tiny(num_steps: int = 3) -> "PipelineConfig":
        return PipelineConfig(LLMConfig.tiny(), TokenizerConfig.tiny(), UpsamplerConfig.tiny(), DiTConfig.tiny(),
                              VAEConfig.tiny(), SamplerConfig(num_steps=num_steps))

    def check(self):
        assert self.tok.temporal == self.dit.latent_frames
        assert self.llm.iframe_len == self.tok.iframe_tokens and self.llm.pframe_len == self.tok.pframe_tokens
        assert self.llm.visual_vocab == self.tok.codebook_size
        assert self.dit.latent_h == 2 * self.tok.grid_h and self.dit.latent_w == 2 * self.tok.grid_w
        assert self.ups.z_channels == self.tok.out_channels and self.ups.target_dim == self.dit.in_channels
        assert self.vae.z_channels == self.dit.in_channels
        return self

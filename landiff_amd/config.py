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
    # TextCond's encoder (llm_cfg.py:60-61 -> FlanT5XXL(model_path="google/flan-t5-xxl"), text_encoder.py:137-146): an HF hub
    # name or a local directory with the T5 encoder checkpoint + tokenizer; max_cond_tokens_num=512 (llm_cfg.py:62)
    text_encoder_path: str = "google/flan-t5-xxl"
    max_cond_tokens: int = 512

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

    @staticmethod
    def config0() -> "LLMConfig":
        """Goes with DiTConfig.config0() / TokenizerConfig.config0(): 8 semantic frames, 64 I tokens + 7 x 16 P tokens."""
        return LLMConfig(num_layers=2, hidden=256, heads=2, mlp=512, visual_vocab=64, text_dim=128, freq_dim=64,
                         micro_hidden=64, iframe_len=64, pframe_len=16, segment_length=8, segment_stride=8)


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

    @staticmethod
    def config0() -> "TokenizerConfig":
        """BASELINE.json configs[0]: 8 frames on a 32x32 semantic grid (= half the 64x64 latent), 64 + 7 x 16 latent tokens."""
        return TokenizerConfig(width=128, layers=2, heads=2, grid_h=32, grid_w=32, temporal=8, pframe_tokens=16,
                               num_latent_tokens=64 + 7 * 16, codebook_size=64, codebook_dim=16, token_size=128,
                               out_channels=128)


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
    # frames covered by the checkpoint's position table, (num_frames - 1) // time_compressed_rate + 1 (dit_video_concat.py:
    # 200-231: the forward slices the first text_len + seq_length rows); 0 = latent_frames
    pos_frames: int = 0
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
    @property
    def pos_rows(self): return self.text_len + (self.pos_frames or self.latent_frames) * self.grid_h * self.grid_w

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
    def config0(num_steps: int = 2) -> "PipelineConfig":
        """BASELINE.json configs[0]: random-init tiny DiT, 8 latent frames, 64x64 latent, 2 DDIM steps (plumbing run of the
        landiff.infer_video entry point)."""
        return PipelineConfig(LLMConfig.config0(), TokenizerConfig.config0(), UpsamplerConfig.tiny(), DiTConfig.config0(),
                              VAEConfig.tiny(), SamplerConfig(num_steps=num_steps, sampler="ddim"))

    @staticmethod
    def tiny(num_steps: int = 3) -> "PipelineConfig":
        return PipelineConfig(LLMConfig.tiny(), TokenizerConfig.tiny(), UpsamplerConfig.tiny(), DiTConfig.tiny(),
                              VAEConfig.tiny(), SamplerConfig(num_steps=num_steps))

    def check_diffusion(self):
        """Consistency of the diffusion-side parts (tokenizer decoder / upsampler / DiT / VAE)."""
        assert self.tok.temporal == self.dit.latent_frames, "sampling_num_frames must equal the tokenizer's temporal_size"
        assert self.dit.latent_h == 2 * self.tok.grid_h and self.dit.latent_w == 2 * self.tok.grid_w
        assert self.ups.z_channels == self.tok.out_channels and self.ups.target_dim == self.dit.in_channels
        assert self.vae.z_channels == self.dit.in_channels
        return self

    def check(self):
        self.check_diffusion()
        assert self.llm.iframe_len == self.tok.iframe_tokens and self.llm.pframe_len == self.tok.pframe_tokens
        assert self.llm.visual_vocab == self.tok.codebook_size
        return self


def build_tokenizer_config0() -> TokenizerConfig:
    """`config_str` target for a configs[0] model YAML (the shipped YAML names landiff.tokenizer.tokenizer_cfg.build_tokenizer)."""
    return TokenizerConfig.config0()


# ------------------------------------------------------------------------------------------------
# the reference's two YAML files (landiff/diffusion/configs/*.yaml) -> the dataclasses above
# ------------------------------------------------------------------------------------------------
_SAMPLERS = {"VPSDEDPMPP2MSampler": "vpsde_dpmpp2m", "VideoDDIMSampler": "ddim"}
_GUIDERS = ("DynamicCFG",)


@dataclass(frozen=True)
class DiffusionInferConfig:
    """Everything CogModelInferWrapper reads from `model_cfg_path` + `infer_cfg_path` (dif_infer.py:274-287: "--base
    {model_cfg_path} {infer_cfg_path} --load {ckpt_path}", merged by arguments.py:302-329)."""
    dit: DiTConfig
    tok: TokenizerConfig
    ups: UpsamplerConfig
    vae: VAEConfig
    sampler: SamplerConfig
    t5_dir: str                 # conditioner_config ... FrozenT5Embedder.params.model_dir
    tokenizer_ckpt: str         # VideoVQWrap.params.ckpt_path
    vae_ckpt: str               # first_stage_config.params.ckpt_path
    base_dit_ckpt: str          # model.pretrain_diffusion_model_ckpt_path
    image_size: tuple           # args.sampling_image_size
    fps: int                    # args.sampling_fps
    bf16: bool
    force_inference: bool       # non-strict checkpoint load (infer_cfgs/2b.yaml:13)

    def pipeline(self, llm: LLMConfig | None = None) -> "PipelineConfig":
        return PipelineConfig(llm or LLMConfig(), self.tok, self.ups, self.dit, self.vae, self.sampler)


def _cls(target: str) -> str:
    return target.rsplit(".", 1)[-1]


def load_diffusion_config(model_cfg_path: str, infer_cfg_path: str) -> DiffusionInferConfig:
    """Reads the reference's model YAML (cogvideox_2b_control_theia_interpolate_video_vq.yaml :1-243) and inference YAML
    (infer_cfgs/2b.yaml :1-13) with PyYAML.  Only the `target:` classes of the shipped configuration are accepted; every
    shape of the diffusion side of the path comes from these files, nothing is assumed."""
    import importlib
    import yaml
    with open(model_cfg_path) as f:
        m = yaml.safe_load(f)["model"]
    with open(infer_cfg_path) as f:
        a = yaml.safe_load(f)["args"]
    net, ctl = m["network_config"], m["control_network_config"]
    assert _cls(net["target"]) == "DiffusionTransformer" and _cls(ctl["target"]) == "ControlDiffusionTransformer", \
        "network_config / control_network_config must be the DiffusionTransformer / ControlDiffusionTransformer pair"
    n, c = net["params"], ctl["params"]
    for k in ("time_embed_dim", "hidden_size", "num_attention_heads", "patch_size", "in_channels", "out_channels",
              "latent_width", "latent_height", "num_frames", "time_compressed_rate"):
        assert n[k] == c[k], f"main and control DiT disagree on {k}: {n[k]} vs {c[k]}"
    assert n["transformer_args"].get("layernorm_order", "pre") == "pre" and not n["transformer_args"].get("is_decoder", False)
    nm, cm = n["modules"], c["modules"]
    assert _cls(nm["pos_embed_config"]["target"]) == "Basic3DPositionEmbeddingMixin", "only the additive 3D sin-cos position embedding is built"
    assert _cls(nm["adaln_layer_config"]["target"]) == "ControlAdaLNMixin" and _cls(cm["adaln_layer_config"]["target"]) == "ControlOutAdaLNMixin"
    assert nm["adaln_layer_config"]["params"].get("qk_ln", True) and cm["adaln_layer_config"]["params"].get("qk_ln", True), "qk_ln: True expected"
    assert cm["adaln_layer_config"]["params"].get("use_zero_linears", False), "use_zero_linears: true expected"
    assert not c.get("use_semantic_injection_adaln", False) and not nm["adaln_layer_config"]["params"].get("use_semantic_injection_adaln", False)
    control_layers = nm["adaln_layer_config"]["params"].get("control_layers", c["num_layers"])
    assert control_layers == c["num_layers"], "control_layers must equal the control DiT's num_layers"
    pos = nm["pos_embed_config"]["params"]
    assert pos == cm["pos_embed_config"]["params"], "main and control position embeddings differ"
    latent_frames = int(a["sampling_num_frames"])
    H, W = a["sampling_image_size"]
    assert H % 8 == 0 and W % 8 == 0 and H // 8 == n["latent_height"] and W // 8 == n["latent_width"], \
        "sampling_image_size must be 8x the DiT's latent_height / latent_width (the position table is built for that grid)"
    # the position table covers (num_frames - 1) // time_compressed_rate + 1 latent frames (dit_video_concat.py:200-226)
    table_frames = (n["num_frames"] - 1) // n["time_compressed_rate"] + 1
    assert latent_frames <= table_frames, f"sampling_num_frames {latent_frames} exceeds the position table ({table_frames} latent frames)"
    assert a["latent_channels"] == n["in_channels"]
    dit = DiTConfig(hidden=n["hidden_size"], heads=n["num_attention_heads"], layers_main=n["num_layers"], layers_control=c["num_layers"],
                    time_embed_dim=n["time_embed_dim"], patch=n["patch_size"], in_channels=n["in_channels"], out_channels=n["out_channels"],
                    latent_h=n["latent_height"], latent_w=n["latent_width"], latent_frames=latent_frames,
                    pos_frames=table_frames, text_len=pos["text_length"],
                    text_dim=nm["patch_embed_config"]["params"]["text_hidden_size"],
                    height_interpolation=float(pos.get("height_interpolation", 1.0)),
                    width_interpolation=float(pos.get("width_interpolation", 1.0)),
                    time_interpolation=float(pos.get("time_interpolation", 1.0)))
    # semantic conditioner: tokenizer (a config function named by config_str, as VQWarp.__init__ imports it) + conv upsampler
    sc = cm["semantic_condition_config"]["params"]
    assert sc.get("feature_type", "video_theia_interpolate") == "video_theia_interpolate"
    vq = sc["semantic_model_config"]["params"]
    mod, fn = vq["config_str"].rsplit(".", 1)
    tok = getattr(importlib.import_module(mod), fn)()
    assert isinstance(tok, TokenizerConfig), f"{vq['config_str']}() must return a landiff_amd.config.TokenizerConfig"
    up = sc["upsample_model_config"]["params"]
    assert up.get("upsample_type", "pixelshuffle") == "pixelshuffle" and not up.get("attn_resolutions") and not up.get("use_mid_attention", False)
    ups = UpsamplerConfig(z_channels=up["z_channels"], ch=up["ch"], ch_mult=tuple(up["ch_mult"]), num_res_blocks=up["num_res_blocks"],
                          out_ch=up["out_ch"], target_dim=sc["target_dim"])
    assert sc["out_dim"] == up["out_ch"]
    fs = m["first_stage_config"]["params"]
    dec = fs["decoder_config"]["params"]
    assert _cls(fs["decoder_config"]["target"]) == "ContextParallelDecoder3D" and not dec.get("gather_norm", False) and not dec.get("attn_resolutions")
    vae = VAEConfig(ch=dec["ch"], ch_mult=tuple(dec["ch_mult"]), num_res_blocks=dec["num_res_blocks"], z_channels=dec["z_channels"],
                    out_ch=dec["out_ch"], temporal_compress_times=n["time_compressed_rate"], scale_factor=float(m["scale_factor"]))
    # sampler stack
    smp = m["sampler_config"]
    kind = _cls(smp["target"])
    assert kind in _SAMPLERS, f"sampler {kind}: only {sorted(_SAMPLERS)} are built"
    sp = smp["params"]
    assert sp.get("fixed_frames", 0) in (0, None), "fixed_frames is the streaming primitive (generate_stream), not a config-file switch here"
    den = m["denoiser_config"]["params"]
    assert _cls(m["denoiser_config"]["target"]) == "DiscreteDenoiser" and _cls(den["scaling_config"]["target"]) == "VideoScaling"
    disc = sp["discretization_config"]
    assert _cls(disc["target"]) == "ZeroSNRDDPMDiscretization" and _cls(den["discretization_config"]["target"]) == "ZeroSNRDDPMDiscretization"
    shift = float(disc.get("params", {}).get("shift_scale", 1.0))
    assert shift == float(den["discretization_config"].get("params", {}).get("shift_scale", 1.0)), "sampler and denoiser discretizations differ"
    gd = sp["guider_config"]
    assert _cls(gd["target"]) in _GUIDERS, "guider: DynamicCFG expected"
    gp = gd["params"]
    assert gp.get("num_steps", sp["num_steps"]) == sp["num_steps"]
    sampler = SamplerConfig(num_steps=int(sp["num_steps"]), cfg_scale=float(gp["scale"]), cfg_exp=float(gp["exp"]), shift_scale=shift,
                            num_idx=int(den["num_idx"]), sampler=_SAMPLERS[kind])
    emb = m["conditioner_config"]["params"]["emb_models"]
    assert len(emb) == 1 and _cls(emb[0]["target"]) == "FrozenT5Embedder" and emb[0]["input_key"] == "txt"
    assert emb[0]["params"]["max_length"] == pos["text_length"], "T5 max_length must equal the DiT's text_length"
    cfg = DiffusionInferConfig(dit=dit, tok=tok, ups=ups, vae=vae, sampler=sampler, t5_dir=emb[0]["params"]["model_dir"],
                               tokenizer_ckpt=vq.get("ckpt_path") or "", vae_ckpt=fs["ckpt_path"],
                               base_dit_ckpt=m["pretrain_diffusion_model_ckpt_path"], image_size=(int(H), int(W)),
                               fps=int(a.get("sampling_fps", 8)), bf16=bool(a.get("bf16", True)),
                               force_inference=bool(a.get("force_inference", False)))
    cfg.pipeline().check_diffusion()
    return cfg


def reference_yaml_docs(cfg: "PipelineConfig", *, tokenizer_config_str: str = "landiff.tokenizer.tokenizer_cfg.build_tokenizer",
                        ckpt_prefix: str = "ckpts/LanDiff") -> tuple:
    """The two YAML documents of the reference's configuration system for a PipelineConfig -- the inverse of
    load_diffusion_config: (model document, inference document) with the reference's keys and `target:` class paths
    (cogvideox_2b_control_theia_interpolate_video_vq.yaml / infer_cfgs/2b.yaml).  The files shipped under
    landiff/diffusion/configs/ are `yaml.safe_dump` of these documents for PipelineConfig.full(); tests write the configs[0] pair
    the same way.  Only what the inference path reads is emitted (no loss / encoder / regularizer sections)."""
    import copy
    d, u, v, sm = cfg.dit, cfg.ups, cfg.vae, cfg.sampler
    dm = "landiff.diffusion.sgm.modules.diffusionmodules."
    dv = "landiff.diffusion.dit_video_concat."
    pos = {"target": dv + "Basic3DPositionEmbeddingMixin",
           "params": {"text_length": d.text_len, "height_interpolation": d.height_interpolation, "width_interpolation": d.width_interpolation}}
    patch = {"target": dv + "ImagePatchEmbeddingMixin", "params": {"text_hidden_size": d.text_dim}}
    targs = {"checkpoint_activations": False, "vocab_size": 1, "max_sequence_length": 64, "layernorm_order": "pre", "skip_init": False,
             "model_parallel_size": 1, "is_decoder": False}
    frames = d.pos_frames or d.latent_frames
    def net(layers):
        return {"time_embed_dim": d.time_embed_dim, "elementwise_affine": True, "num_frames": v.temporal_compress_times * (frames - 1) + 1,
                "time_compressed_rate": v.temporal_compress_times, "latent_width": d.latent_w, "latent_height": d.latent_h,
                "num_layers": layers, "patch_size": d.patch, "in_channels": d.in_channels, "out_channels": d.out_channels,
                "hidden_size": d.hidden, "adm_in_channels": 256, "num_attention_heads": d.heads, "transformer_args": dict(targs)}
    control = net(d.layers_control)
    control["use_semantic_injection_adaln"] = False
    control["modules"] = {
        "semantic_condition_config": {"target": "landiff.diffusion.semantic_models.condition.SemanticCond", "params": {
            "out_dim": u.out_ch, "target_dim": u.target_dim, "feature_type": "video_theia_interpolate", "zero_init_conv_out": True,
            "semantic_model_config": {"target": "landiff.diffusion.semantic_models.feature_extractor.vq_warp.VideoVQWrap", "params": {
                "config_str": tokenizer_config_str, "ckpt_path": f"{ckpt_prefix}/tokenizer/model.safetensors", "freeze_model": True,
                "freeze_encoder": False}},
            "upsample_model_config": {"target": "landiff.diffusion.semantic_models.modules.vq_gan_blocks.Decoder", "params": {
                "z_channels": u.z_channels, "resolution": 16, "in_channels": u.ch, "out_ch": u.out_ch, "ch": u.ch, "ch_mult": list(u.ch_mult),
                "num_res_blocks": u.num_res_blocks, "attn_resolutions": [], "dropout": 0.0, "use_mid_attention": False,
                "upsample_type": "pixelshuffle"}}}},
        "pos_embed_config": copy.deepcopy(pos), "patch_embed_config": copy.deepcopy(patch),
        "adaln_layer_config": {"target": dv + "ControlOutAdaLNMixin", "params": {"qk_ln": True, "use_zero_linears": True}},
        "final_layer_config": {"target": dv + "EmptyFinalLayerMixin"}}
    main = net(d.layers_main)
    main["modules"] = {
        "pos_embed_config": pos, "patch_embed_config": patch,
        "adaln_layer_config": {"target": dv + "ControlAdaLNMixin", "params": {"qk_ln": True, "use_semantic_injection_adaln": False,
                                                                                "control_layers": d.layers_control}},
        "final_layer_config": {"target": dv + "FinalLayerMixin"}}
    disc = {"target": dm + "discretizer.ZeroSNRDDPMDiscretization", "params": {"shift_scale": sm.shift_scale}}
    sampler_cls = {"vpsde_dpmpp2m": "VPSDEDPMPP2MSampler", "ddim": "VideoDDIMSampler"}[sm.sampler]
    model = {"model": {
        "scale_factor": v.scale_factor, "disable_first_stage_autocast": True,
        "pretrain_diffusion_model_ckpt_path": f"{ckpt_prefix}/CogVideoX-2b-sat/transformer/1000/mp_rank_00_model_states.pt", "freeze_dit": True,
        "denoiser_config": {"target": dm + "denoiser.DiscreteDenoiser", "params": {
            "num_idx": sm.num_idx, "quantize_c_noise": False, "scaling_config": {"target": dm + "denoiser_scaling.VideoScaling"},
            "discretization_config": disc}},
        "control_network_config": {"target": dv + "ControlDiffusionTransformer", "params": control},
        "network_config": {"target": dv + "DiffusionTransformer", "params": main},
        "conditioner_config": {"target": "landiff.diffusion.sgm.modules.GeneralConditioner", "params": {"emb_models": [{
            "is_trainable": False, "input_key": "txt", "ucg_rate": 0.1,
            "target": "landiff.diffusion.sgm.modules.encoders.modules.FrozenT5Embedder",
            "params": {"model_dir": f"{ckpt_prefix}/CogVideoX-2b-sat/t5-v1_1-xxl", "max_length": d.text_len}}]}},
        "first_stage_config": {"target": "landiff.diffusion.vae_modules.autoencoder.VideoAutoencoderInferenceWrapper", "params": {
            "cp_size": 1, "ckpt_path": f"{ckpt_prefix}/CogVideoX-2b-sat/vae/3d-vae.pt", "ignore_keys": ["loss"],
            "decoder_config": {"target": "landiff.diffusion.vae_modules.cp_enc_dec.ContextParallelDecoder3D", "params": {
                "double_z": True, "z_channels": v.z_channels, "resolution": 256, "in_channels": 3, "out_ch": v.out_ch, "ch": v.ch,
                "ch_mult": list(v.ch_mult), "attn_resolutions": [], "num_res_blocks": v.num_res_blocks, "dropout": 0.0, "gather_norm": False}}}},
        "sampler_config": {"target": dm + "sampling." + sampler_cls, "params": {
            "num_steps": sm.num_steps, "verbose": True, "discretization_config": copy.deepcopy(disc),
            "guider_config": {"target": dm + "guiders.DynamicCFG", "params": {
                "scale": sm.cfg_scale, "exp": sm.cfg_exp, "num_steps": sm.num_steps}}}}}}
    infer = {"args": {"image2video": False, "latent_channels": d.in_channels, "mode": "inference", "batch_size": 1,
                      "sampling_image_size": [8 * d.latent_h, 8 * d.latent_w], "sampling_num_frames": d.latent_frames, "sampling_fps": 8,
                      "bf16": True, "amp_exclude_key": ["quantizer"], "force_inference": True}}
    return model, infer


def write_reference_yaml(root: str, cfg: "PipelineConfig", **kw) -> tuple:
    """Writes the two documents at the paths CogModelInferWrapper defaults to, relative to `root`."""
    import os
    import yaml
    model, infer = reference_yaml_docs(cfg, **kw)
    mp = os.path.join(root, "landiff/diffusion/configs/cogvideox_2b_control_theia_interpolate_video_vq.yaml")
    ip = os.path.join(root, "landiff/diffusion/configs/infer_cfgs/2b.yaml")
    os.makedirs(os.path.dirname(ip), exist_ok=True)
    head = ("# Generated by landiff_amd.config.write_reference_yaml (keys / `target:` names / values of the reference's file of this name,\n"
            "# inference-relevant sections only).  Read back by landiff_amd.config.load_diffusion_config.\n")
    for path, doc in ((mp, model), (ip, infer)):
        with open(path, "w") as f:
            f.write(head)
            yaml.safe_dump(doc, f, default_flow_style=None, sort_keys=True, width=150)
    return mp, ip

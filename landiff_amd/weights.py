"""State-dict key maps, seeded random init, and checkpoint discovery/loading.

Key names are the reference's: ``llm/model.safetensors`` (Semantic1DLM), ``tokenizer/model.safetensors``
(VideoVQ), ``diffusion/<iter>/mp_rank_00_model_states.pt['module']`` (SATControlVideoDiffusionEngine) and
``CogVideoX-2b-sat/vae/3d-vae.pt['state_dict']``.  Component state dicts use keys *relative* to:
  llm  -> ''                                   tok -> '' (decoder.*, quantizer.*)
  ups  -> '<control>.semantic_conditioner.'    (upsample_model.*, conv_out.*)
  dit_main / dit_control -> 'model.{main,control}_model.diffusion_model.'
  vae  -> '' (decoder.*)
"""
from __future__ import annotations

import os

import torch

from .config import DiTConfig, LLMConfig, PipelineConfig, TokenizerConfig, UpsamplerConfig, VAEConfig

Spec = list  # of (name, shape, kind)


# ----------------------------------------------------------------------------------------------
# key specs
# ----------------------------------------------------------------------------------------------
def llm_spec(c: LLMConfig) -> Spec:
    s = []
    for i in range(c.num_layers):
        p = f"transformer.blocks.{i}."
        s += [(p + "norm0.weight", (c.hidden,), "g"), (p + "norm1.weight", (c.hidden,), "g"),
              (p + "wqkv.weight", (3 * c.hidden, c.hidden), "w"), (p + "wo.weight", (c.hidden, c.hidden), "w"),
              (p + "mlp.w1.weight", (c.mlp, c.hidden), "w"), (p + "mlp.w3.weight", (c.mlp, c.hidden), "w"),
              (p + "mlp.w2.weight", (c.hidden, c.mlp), "w")]
    s += [("transformer.layer_norm.weight", (c.hidden,), "g"), ("transformer.layer_norm.bias", (c.hidden,), "b"),
          ("transformer.head.weight", (c.vocab, c.hidden), "w"),
          ("visual_embedding_model.tok_emb_code.weight", (c.vocab, c.hidden), "e"),
          ("cond_model.embeddings.fc0.weight", (c.hidden, c.text_dim), "w"), ("cond_model.embeddings.fc0.bias", (c.hidden,), "b"),
          ("cond_model.embeddings.fc1.weight", (c.hidden, c.hidden), "w"), ("cond_model.embeddings.fc1.bias", (c.hidden,), "b"),
          ("cond_model.null_text_embedding", (c.hidden,), "e")]
    for k in ("frames", "motion_score"):
        p = f"micro_condition.mlps.{k}."
        s += [(p + "0.weight", (c.micro_hidden, c.freq_dim), "w"), (p + "0.bias", (c.micro_hidden,), "b"),
              (p + "2.weight", (c.hidden, c.micro_hidden), "w"), (p + "2.bias", (c.hidden,), "b")]
    return s


def tokenizer_spec(c: TokenizerConfig) -> Spec:
    w = c.width
    s = [("decoder.mask_token", (1, 1, w), "e"),
         ("decoder.decoder_embed.weight", (w, c.token_size), "w"), ("decoder.decoder_embed.bias", (w,), "b"),
         ("decoder.ln_pre.weight", (w,), "g"), ("decoder.ln_pre.bias", (w,), "b")]
    for i in range(c.layers):
        p = f"decoder.transformer.{i}."
        s += [(p + "ln_1.weight", (w,), "g"), (p + "ln_1.bias", (w,), "b"),
              (p + "attn.wq.weight", (w, w), "w"), (p + "attn.wk.weight", (w, w), "w"),
              (p + "attn.wv.weight", (w, w), "w"), (p + "attn.wo.weight", (w, w), "w"),
              (p + "ln_2.weight", (w,), "g"), (p + "ln_2.bias", (w,), "b"),
              (p + "mlp.c_fc.weight", (4 * w, w), "w"), (p + "mlp.c_fc.bias", (4 * w,), "b"),
              (p + "mlp.c_proj.weight", (w, 4 * w), "w"), (p + "mlp.c_proj.bias", (w,), "b")]
    s += [("decoder.ln_post.weight", (w,), "g"), ("decoder.ln_post.bias", (w,), "b"),
          ("decoder.ffn.0.weight", (2 * w, w), "w"), ("decoder.ffn.0.bias", (2 * w,), "b"),
          ("decoder.ffn.2.weight", (c.out_channels, 2 * w), "w"), ("decoder.ffn.2.bias", (c.out_channels,), "b"),
          # vector-quantize-pytorch 1.19.2: Euclidean codebook + project_out
          ("quantizer._codebook.embed", (1, c.codebook_size, c.codebook_dim), "e1"),
          ("quantizer.project_out.weight", (c.token_size, c.codebook_dim), "w1"),
          ("quantizer.project_out.bias", (c.token_size,), "b")]
    return s


def tokenizer_encoder_spec(c: TokenizerConfig) -> Spec:
    """Encoder half of VideoVQ: TiTokEncoder (landiff/tokenizer/modules/blocks.py:311-656, patch_size 1,
    inside_latent_tokens, bias=False attention) + the feature normalisation buffers (video_titok_vq.py:55-68) +
    VectorQuantize.project_in / codebook (vector-quantize-pytorch 1.19.2)."""
    w, cin = c.width, c.out_channels
    s = [("mean", (cin,), "b"), ("std", (cin,), "g"),
         ("encoder.patch_embed.weight", (w, cin, 1, 1), "w"), ("encoder.patch_embed.bias", (w,), "b"),
         ("encoder.IFrame_latent_tokens", (c.iframe_tokens, w), "e"), ("encoder.PFrame_latent_tokens", (c.pframe_tokens, w), "e"),
         ("encoder.ln_pre.weight", (w,), "g"), ("encoder.ln_pre.bias", (w,), "b")]
    for i in range(c.layers):
        p = f"encoder.transformer.{i}."
        s += [(p + "ln_1.weight", (w,), "g"), (p + "ln_1.bias", (w,), "b"),
              (p + "attn.wq.weight", (w, w), "w"), (p + "attn.wk.weight", (w, w), "w"),
              (p + "attn.wv.weight", (w, w), "w"), (p + "attn.wo.weight", (w, w), "w"),
              (p + "ln_2.weight", (w,), "g"), (p + "ln_2.bias", (w,), "b"),
              (p + "mlp.c_fc.weight", (4 * w, w), "w"), (p + "mlp.c_fc.bias", (4 * w,), "b"),
              (p + "mlp.c_proj.weight", (w, 4 * w), "w"), (p + "mlp.c_proj.bias", (w,), "b")]
    s += [("encoder.ln_post.weight", (w,), "g"), ("encoder.ln_post.bias", (w,), "b"),
          ("encoder.proj_out.weight", (c.token_size, w), "w"), ("encoder.proj_out.bias", (c.token_size,), "b"),
          ("quantizer.project_in.weight", (c.codebook_dim, c.token_size), "w"), ("quantizer.project_in.bias", (c.codebook_dim,), "b"),
          ("quantizer._codebook.embed", (1, c.codebook_size, c.codebook_dim), "e1")]
    return s


def _res2d(p: str, cin: int, cout: int) -> Spec:
    s = [(p + "norm1.weight", (cin,), "g"), (p + "norm1.bias", (cin,), "b"),
         (p + "conv1.weight", (cout, cin, 3, 3), "w"), (p + "conv1.bias", (cout,), "b"),
         (p + "norm2.weight", (cout,), "g"), (p + "norm2.bias", (cout,), "b"),
         (p + "conv2.weight", (cout, cout, 3, 3), "w"), (p + "conv2.bias", (cout,), "b")]
    if cin != cout:
        s += [(p + "nin_shortcut.weight", (cout, cin, 1, 1), "w"), (p + "nin_shortcut.bias", (cout,), "b")]
    return s


def upsampler_levels(c: UpsamplerConfig):
    """[(level, [(cin, cout) per block], has_upsample)] from the coarsest level down (vq_gan_blocks.py:516-551)."""
    nres = len(c.ch_mult)
    block_in = int(c.ch * c.ch_mult[nres - 1])
    out = []
    for lvl in reversed(range(nres)):
        block_out = int(c.ch * c.ch_mult[lvl])
        blocks = []
        for _ in range(c.num_res_blocks + 1):
            blocks.append((block_in, block_out))
            block_in = block_out
        out.append((lvl, blocks, lvl != 0))
    return out


def upsampler_spec(c: UpsamplerConfig) -> Spec:
    nres = len(c.ch_mult)
    top = int(c.ch * c.ch_mult[nres - 1])
    p = "upsample_model."
    s = [(p + "conv_in.weight", (top, c.z_channels, 3, 3), "w"), (p + "conv_in.bias", (top,), "b")]
    s += _res2d(p + "mid.block_1.", top, top) + _res2d(p + "mid.block_2.", top, top)
    last = top
    for lvl, blocks, up in upsampler_levels(c):
        for j, (cin, cout) in enumerate(blocks):
            s += _res2d(p + f"up.{lvl}.block.{j}.", cin, cout)
            last = cout
        if up:  # PixelShuffle(2) then Conv2d(C/4 -> C)
            s += [(p + f"up.{lvl}.upsample.conv.weight", (last, last // 4, 3, 3), "w"),
                  (p + f"up.{lvl}.upsample.conv.bias", (last,), "b")]
    s += [(p + "norm_out.weight", (last,), "g"), (p + "norm_out.bias", (last,), "b"),
          (p + "conv_out.weight", (c.out_ch, last, 3, 3), "w"), (p + "conv_out.bias", (c.out_ch,), "b"),
          ("conv_out.weight", (c.target_dim, c.out_ch, 3, 3), "w"), ("conv_out.bias", (c.target_dim,), "b")]
    return s


def dit_spec(c: DiTConfig, control: bool) -> Spec:
    d, L, te = c.hidden, (c.layers_control if control else c.layers_main), c.time_embed_dim
    pd = c.patch * c.patch
    s = [("time_embed.0.weight", (te, d), "w"), ("time_embed.0.bias", (te,), "b"),
         ("time_embed.2.weight", (te, te), "w"), ("time_embed.2.bias", (te,), "b"),
         ("mixins.pos_embed.pos_embedding", (1, c.pos_rows, d), "pos"),
         ("mixins.patch_embed.proj.weight", (d, c.in_channels, c.patch, c.patch), "w"),
         ("mixins.patch_embed.proj.bias", (d,), "b"),
         ("mixins.patch_embed.text_proj.weight", (d, c.text_dim), "w"), ("mixins.patch_embed.text_proj.bias", (d,), "b")]
    for i in range(L):
        p = f"transformer.layers.{i}."
        s += [(p + "input_layernorm.weight", (d,), "g"), (p + "input_layernorm.bias", (d,), "b"),
              (p + "attention.query_key_value.weight", (3 * d, d), "w"), (p + "attention.query_key_value.bias", (3 * d,), "b"),
              (p + "attention.dense.weight", (d, d), "w"), (p + "attention.dense.bias", (d,), "b"),
              (p + "post_attention_layernorm.weight", (d,), "g"), (p + "post_attention_layernorm.bias", (d,), "b"),
              (p + "mlp.dense_h_to_4h.weight", (4 * d, d), "w"), (p + "mlp.dense_h_to_4h.bias", (4 * d,), "b"),
              (p + "mlp.dense_4h_to_h.weight", (d, 4 * d), "w"), (p + "mlp.dense_4h_to_h.bias", (d,), "b"),
              (f"mixins.adaln_layer.adaLN_modulations.{i}.1.weight", (12 * d, te), "w"),
              (f"mixins.adaln_layer.adaLN_modulations.{i}.1.bias", (12 * d,), "b"),
              (f"mixins.adaln_layer.query_layernorm_list.{i}.weight", (c.head_dim,), "g"),
              (f"mixins.adaln_layer.query_layernorm_list.{i}.bias", (c.head_dim,), "b"),
              (f"mixins.adaln_layer.key_layernorm_list.{i}.weight", (c.head_dim,), "g"),
              (f"mixins.adaln_layer.key_layernorm_list.{i}.bias", (c.head_dim,), "b")]
        if control:
            s += [(f"mixins.adaln_layer.zero_linears.{i}.weight", (d, d), "w")]
    s += [("transformer.final_layernorm.weight", (d,), "g"), ("transformer.final_layernorm.bias", (d,), "b")]
    if not control:
        s += [("mixins.final_layer.norm_final.weight", (d,), "g"), ("mixins.final_layer.norm_final.bias", (d,), "b"),
              ("mixins.final_layer.linear.weight", (pd * c.out_channels, d), "w"),
              ("mixins.final_layer.linear.bias", (pd * c.out_channels,), "b"),
              ("mixins.final_layer.adaLN_modulation.1.weight", (2 * d, te), "w"),
              ("mixins.final_layer.adaLN_modulation.1.bias", (2 * d,), "b")]
    return s


def _res3d(p: str, cin: int, cout: int, zq: int) -> Spec:
    s = []
    for n, ch in (("norm1", cin), ("norm2", cout)):
        s += [(p + f"{n}.norm_layer.weight", (ch,), "g"), (p + f"{n}.norm_layer.bias", (ch,), "b"),
              (p + f"{n}.conv_y.conv.weight", (ch, zq, 1, 1, 1), "w1"), (p + f"{n}.conv_y.conv.bias", (ch,), "g"),
              (p + f"{n}.conv_b.conv.weight", (ch, zq, 1, 1, 1), "w1"), (p + f"{n}.conv_b.conv.bias", (ch,), "b")]
    s += [(p + "conv1.conv.weight", (cout, cin, 3, 3, 3), "w"), (p + "conv1.conv.bias", (cout,), "b"),
          (p + "conv2.conv.weight", (cout, cout, 3, 3, 3), "w"), (p + "conv2.conv.bias", (cout,), "b")]
    if cin != cout:
        s += [(p + "nin_shortcut.weight", (cout, cin, 1, 1, 1), "w"), (p + "nin_shortcut.bias", (cout,), "b")]
    return s


def vae_levels(c: VAEConfig):
    """[(level, [(cin,cout)...], upsample: None|'space'|'space_time')] coarsest first (cp_enc_dec.py:984-1016)."""
    nres = len(c.ch_mult)
    tcl = {1: 0, 2: 1, 4: 2, 8: 3}[c.temporal_compress_times]
    block_in = c.ch * c.ch_mult[nres - 1]
    out = []
    for lvl in reversed(range(nres)):
        block_out = c.ch * c.ch_mult[lvl]
        blocks = []
        for _ in range(c.num_res_blocks + 1):
            blocks.append((block_in, block_out))
            block_in = block_out
        up = None
        if lvl != 0:
            up = "space" if lvl < nres - tcl else "space_time"
        out.append((lvl, blocks, up))
    return out


def vae_spec(c: VAEConfig) -> Spec:
    zq = c.z_channels
    top = c.ch * c.ch_mult[-1]
    p = "decoder."
    s = [(p + "conv_in.conv.weight", (top, zq, 3, 3, 3), "w"), (p + "conv_in.conv.bias", (top,), "b")]
    s += _res3d(p + "mid.block_1.", top, top, zq) + _res3d(p + "mid.block_2.", top, top, zq)
    last = top
    for lvl, blocks, up in vae_levels(c):
        for j, (cin, cout) in enumerate(blocks):
            s += _res3d(p + f"up.{lvl}.block.{j}.", cin, cout, zq)
            last = cout
        if up:
            s += [(p + f"up.{lvl}.upsample.conv.weight", (last, last, 3, 3), "w"),
                  (p + f"up.{lvl}.upsample.conv.bias", (last,), "b")]
    s += [(p + "norm_out.norm_layer.weight", (last,), "g"), (p + "norm_out.norm_layer.bias", (last,), "b"),
          (p + "norm_out.conv_y.conv.weight", (last, zq, 1, 1, 1), "w1"), (p + "norm_out.conv_y.conv.bias", (last,), "g"),
          (p + "norm_out.conv_b.conv.weight", (last, zq, 1, 1, 1), "w1"), (p + "norm_out.conv_b.conv.bias", (last,), "b"),
          (p + "conv_out.conv.weight", (c.out_ch, last, 3, 3, 3), "w"), (p + "conv_out.conv.bias", (c.out_ch,), "b")]
    return s


# ----------------------------------------------------------------------------------------------
# seeded random init (synthetic weights for bench / parity tests)
# ----------------------------------------------------------------------------------------------
def init_state(spec: Spec, seed: int, dtype=torch.float32, device="cpu") -> dict:
    """Deterministic synthetic weights: matrices ~ N(0, 1/fan_in) (variance preserving so that
    activations stay O(1) through deep stacks), gains ~ 1 + 0.1 N, biases ~ 0.05 N.
    device="cuda" draws on the GPU (full-size bench weights; a different stream than the CPU one)."""
    g = torch.Generator(device=device).manual_seed(seed)
    rn = lambda shape: torch.randn(shape, generator=g, device=device)
    out = {}
    for name, shape, kind in spec:
        if kind in ("w", "w1"):
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            t = rn(shape) * (fan_in ** -0.5)
        elif kind == "g":
            t = 1.0 + 0.1 * rn(shape)
        elif kind == "b":
            t = 0.05 * rn(shape)
        elif kind == "pos":
            t = 0.1 * rn(shape)
        elif kind in ("e", "e1"):
            t = rn(shape) * (1.0 if kind == "e1" else 0.5)
        else:
            raise ValueError(kind)
        out[name] = t.to(dtype)
    return out


def init_pipeline_state(cfg: PipelineConfig, seed: int = 1234, dtype=torch.float32, parts=None, device="cpu") -> dict:
    """{'llm','tok','ups','dit_main','dit_control','vae'} -> state dicts of synthetic weights."""
    specs = {
        "llm": lambda: llm_spec(cfg.llm), "tok": lambda: tokenizer_spec(cfg.tok),
        "ups": lambda: upsampler_spec(cfg.ups), "dit_main": lambda: dit_spec(cfg.dit, False),
        "dit_control": lambda: dit_spec(cfg.dit, True), "vae": lambda: vae_spec(cfg.vae),
    }
    out = {}
    for i, (k, fn) in enumerate(specs.items()):
        if parts is None or k in parts:
            out[k] = init_state(fn(), seed + 101 * i, dtype, device)
    return out


# ----------------------------------------------------------------------------------------------
# checkpoint layout (ckpts/README.md:27-45, landiff/utils.py:129-179)
# ----------------------------------------------------------------------------------------------
CKPT_FILES = {
    "llm": "llm/model.safetensors",
    "tokenizer": "tokenizer/model.safetensors",
    "diffusion_latest": "diffusion/latest",
    "base_latest": "CogVideoX-2b-sat/transformer/latest",
    "vae": "CogVideoX-2b-sat/vae/3d-vae.pt",
    "t5": "CogVideoX-2b-sat/t5-v1_1-xxl",
}


def resolve_ckpt_root(repo_root: str | None = None) -> str:
    """$LANDIFF_HOME, else <repo>/ckpts/LanDiff (the reference then falls back to an HF download,
    which needs network and is not attempted here)."""
    home = os.environ.get("LANDIFF_HOME")
    if home and os.path.isdir(home):
        return home
    root = os.path.join(repo_root or os.getcwd(), "ckpts", "LanDiff")
    if os.path.isdir(root):
        return root
    raise FileNotFoundError("LanDiff checkpoints not found: set LANDIFF_HOME or populate ckpts/LanDiff "
                            "(layout: ckpts/README.md of the reference)")


def _sat_module(path_dir: str) -> dict:
    with open(os.path.join(path_dir, "latest")) as f:
        it = f.read().strip()
    return _load_pickle(os.path.join(path_dir, it, "mp_rank_00_model_states.pt"))["module"]


def _sub(sd: dict, prefix: str) -> dict:
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


def load_llm_state(path: str) -> dict:
    from safetensors.torch import load_file
    return load_file(path)


def load_tokenizer_encoder_state(path: str) -> dict:
    """tokenizer/model.safetensors (VideoVQ state dict, tokenizer_cfg.py:104-118) -> the keys of tokenizer_encoder_spec:
    encoder.*, quantizer.project_in.*, quantizer._codebook.embed and the feature mean/std buffers
    (video_titok_vq.py:55-68; identity normalisation when the checkpoint carries no statistics)."""
    from safetensors.torch import load_file
    sd = load_file(path)
    out = {k: v for k, v in sd.items()
           if k.startswith("encoder.") or k.startswith("quantizer.project_in.") or k == "quantizer._codebook.embed"}
    cin = out["encoder.patch_embed.weight"].shape[1]
    out["mean"] = sd["mean"].reshape(-1).float() if "mean" in sd else torch.zeros(cin)
    out["std"] = sd["std"].reshape(-1).float() if "std" in sd else torch.ones(cin)
    return out


def resolve_ckpt_path(path: str, root: str | None = None) -> str:
    """A checkpoint path as the CLI / the YAML files give it: used as is when it exists (absolute, or relative to the working
    directory -- the reference resolves "ckpts/LanDiff/..." against the cwd); otherwise its part below "ckpts/LanDiff/" is
    looked up in the checkpoint root ($LANDIFF_HOME), so the shipped relative paths work from any directory."""
    if os.path.exists(path):
        return path
    marker = "ckpts/LanDiff/"
    norm = path.replace(os.sep, "/")
    if marker in norm:
        cand = os.path.join(root or resolve_ckpt_root(), norm.split(marker, 1)[1])
        if os.path.exists(cand):
            return cand
    raise FileNotFoundError(f"checkpoint path {path!r} not found (cwd {os.getcwd()!r}; LANDIFF_HOME={os.environ.get('LANDIFF_HOME')!r})")


def _load_pickle(path: str) -> dict:
    # a full pickle (argparse namespaces next to the tensors): torch >= 2.6 defaults to weights_only=True and would refuse it;
    # the reference loads these trusted local files with the old default (dit_video_concat.py:1176, sat load_checkpoint)
    return torch.load(path, map_location="cpu", weights_only=False)


def load_diffusion_states(diffusion_dir: str, root: str | None = None, *, base_dit_ckpt: str | None = None,
                          vae_ckpt: str | None = None, tokenizer_ckpt: str | None = None) -> dict:
    """Component state dicts of the diffusion stage, assembled in the reference's load order:
      1. VideoVQWrap loads `tokenizer_ckpt` (tokenizer/model.safetensors, vq_warp.py:38-48), the first stage loads
         `vae_ckpt`['state_dict'] (autoencoder.py:603-614), ControlDiffWarp loads `base_dit_ckpt`['module'] with the 'model.'
         prefix stripped into BOTH the main and the control DiT (dit_video_concat.py:1176-1189);
      2. sat load_checkpoint(diffusion_dir: `latest` -> <iter>/mp_rank_00_model_states.pt['module'], non-strict under
         force_inference) then overrides whatever it carries (dif_infer.py:144).
    base_dit_ckpt / vae_ckpt default to the layout of ckpts/README.md:27-45 under `root` (base: transformer/latest)."""
    mod = _sat_module(diffusion_dir)
    if base_dit_ckpt is not None:
        base = _load_pickle(resolve_ckpt_path(base_dit_ckpt, root))["module"]
    else:
        base = _sat_module(os.path.join(root, "CogVideoX-2b-sat", "transformer"))
    base = _sub(base, "model.diffusion_model.")
    main = dict(base)
    main.update(_sub(mod, "model.main_model.diffusion_model."))
    ctrl_all = _sub(mod, "model.control_model.diffusion_model.")
    ctrl = {k: v for k, v in base.items()}
    ctrl.update({k: v for k, v in ctrl_all.items() if not k.startswith("semantic_conditioner.")})
    sem = _sub(ctrl_all, "semantic_conditioner.")
    tok = {}
    if tokenizer_ckpt:
        from safetensors.torch import load_file
        tok = {k: v for k, v in load_file(resolve_ckpt_path(tokenizer_ckpt, root)).items()
               if k.startswith("decoder.") or k.startswith("quantizer.") or k in ("mean", "std")}
    tok.update(_sub(sem, "semantic_model.model."))
    ups = {k: v for k, v in sem.items() if k.startswith("upsample_model.") or k.startswith("conv_out.")}
    vae_path = resolve_ckpt_path(vae_ckpt, root) if vae_ckpt is not None else os.path.join(root, CKPT_FILES["vae"])
    vae = {k: v for k, v in _load_pickle(vae_path)["state_dict"].items() if k.startswith("decoder.")}    # lightning pickle
    vae.update({k: v for k, v in _sub(mod, "first_stage_model.").items() if k.startswith("decoder.")})
    return {"dit_main": main, "dit_control": ctrl, "tok": tok, "ups": ups, "vae": vae}


def save_checkpoint_tree(root: str, states: dict, *, iteration: str = "1", base_iteration: str = "1000",
                         split_base: bool = True) -> str:
    """Writes component state dicts ({'llm','tok','ups','dit_main','dit_control','vae'}, e.g. init_pipeline_state) as a
    checkpoint tree in the reference's layout (ckpts/README.md:27-45) -- the inverse of load_llm_state / load_diffusion_states:

        <root>/llm/model.safetensors                                     Semantic1DLM state dict
        <root>/tokenizer/model.safetensors                               VideoVQ state dict (decoder + quantizer halves given)
        <root>/diffusion/{latest, <iteration>/mp_rank_00_model_states.pt}   ['module']: control DiT + semantic conditioner,
                                                                         and the main DiT's own (non-base) keys
        <root>/CogVideoX-2b-sat/transformer/{latest, <base_iteration>/mp_rank_00_model_states.pt}   ['module']: 'model.diffusion_model.*'
        <root>/CogVideoX-2b-sat/vae/3d-vae.pt                            ['state_dict']: 'decoder.*'

    split_base: the main DiT's weights go to the CogVideoX base checkpoint (as released: the main DiT is frozen and the control
    checkpoint holds only what was trained); False writes them into the diffusion checkpoint as well (both orders must load
    to the same result).  Synthetic-weight runs of the drop-in entry point (BASELINE configs[0]) and tests use this."""
    import argparse
    from safetensors.torch import save_file
    c = lambda sd: {k: v.detach().cpu().contiguous() for k, v in sd.items()}
    os.makedirs(os.path.join(root, "llm"), exist_ok=True)
    os.makedirs(os.path.join(root, "tokenizer"), exist_ok=True)
    if "llm" in states:
        save_file(c(states["llm"]), os.path.join(root, CKPT_FILES["llm"]))
    save_file(c(states["tok"]), os.path.join(root, CKPT_FILES["tokenizer"]))
    ctl = "model.control_model.diffusion_model."
    mod = {ctl + k: v for k, v in c(states["dit_control"]).items()}
    mod.update({ctl + "semantic_conditioner.semantic_model.model." + k: v for k, v in c(states["tok"]).items()})
    mod.update({ctl + "semantic_conditioner." + k: v for k, v in c(states["ups"]).items()})
    base = {"model.diffusion_model." + k: v for k, v in c(states["dit_main"]).items()}
    if not split_base:
        mod.update({"model.main_model.diffusion_model." + k: v for k, v in c(states["dit_main"]).items()})
    for d, it, sd in ((os.path.join(root, "diffusion"), iteration, mod),
                      (os.path.join(root, "CogVideoX-2b-sat", "transformer"), base_iteration, base)):
        os.makedirs(os.path.join(d, it), exist_ok=True)
        with open(os.path.join(d, "latest"), "w") as f:
            f.write(it)
        torch.save({"module": sd, "args": argparse.Namespace(mode="inference")}, os.path.join(d, it, "mp_rank_00_model_states.pt"))
    os.makedirs(os.path.join(root, "CogVideoX-2b-sat", "vae"), exist_ok=True)
    torch.save({"state_dict": c(states["vae"])}, os.path.join(root, CKPT_FILES["vae"]))
    return root

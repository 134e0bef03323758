"""Generates tests/golden/*.npz by IMPORTING THE REFERENCE in this container (TEST INFRASTRUCTURE).

Run:  python oracle/gen_golden.py            (needs /root/reference; never runs on the GPU box)

The reference's absent third-party imports (sat, fiddle, omegaconf, vector_quantize_pytorch,
pytorch_lightning, torchvision, imageio, transformers ...) are satisfied by permissive in-memory
stub modules created below -- nothing of the reference is copied; only inputs/outputs of its own
code are stored.  Harness adaptations that do not touch arithmetic: CPU devices,
world-size-1 gloo group for the VAE's fake context parallelism, dense-mask attention path.
Each fixture also stores the seeded random weights it was produced with, so the tests feed the
same numbers to the oracle.
"""
from __future__ import annotations

import hashlib
import importlib.machinery
import os
import sys
import types
import typing

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


# ------------------------------------------------------------------------------------------
def install_stubs():
    os.environ["LANDIFF_SKIP_INIT"] = "1"
    sys.path.insert(0, REF)

    class _Any:
        def __init__(self, *a, **k): pass
        def __call__(self, *a, **k): return _Any()
        def __getattr__(self, n): return _Any()

    class StubModule(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            full = self.__name__ + "." + name
            if full in sys.modules:
                return sys.modules[full]
            return type(name, (_Any,), {})

    def stub(name):
        parts = name.split(".")
        for i in range(1, len(parts) + 1):
            n = ".".join(parts[:i])
            if n not in sys.modules:
                m = StubModule(n)
                m.__path__ = []
                m.__spec__ = importlib.machinery.ModuleSpec(n, None)
                sys.modules[n] = m

    for s in ["sat", "sat.helpers", "sat.model", "sat.model.base_model", "sat.model.mixins", "sat.mpu",
              "sat.mpu.layers", "sat.ops", "sat.ops.layernorm", "sat.transformer_defaults", "sat.arguments",
              "sat.training", "sat.training.model_io", "deepspeed", "omegaconf", "fiddle",
              "vector_quantize_pytorch", "pytorch_lightning", "kornia", "beartype", "beartype.typing", "imageio",
              "torchvision", "torchvision.transforms", "torchvision.transforms.v2", "flash_attn",
              "flash_attn.flash_attn_interface", "wandb", "decord", "webdataset", "transformers", "open_clip"]:
        stub(s)
    sys.modules["beartype.typing"].__dict__.update({k: getattr(typing, k) for k in dir(typing) if not k.startswith("_")})
    sys.modules["beartype"].beartype = lambda f: f
    sys.modules["pytorch_lightning"].LightningModule = nn.Module
    sys.modules["sat.model.mixins"].BaseMixin = nn.Module
    sys.modules["sat.model.base_model"].BaseModel = nn.Module
    sys.modules["sat.model.base_model"].non_conflict = lambda f: f
    sys.modules["sat.ops.layernorm"].LayerNorm = nn.LayerNorm


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    conv = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu()
            v = v.float().numpy() if v.dtype == torch.bfloat16 else v.numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def sd_arrays(prefix, sd):
    return {f"{prefix}/{k}": v for k, v in sd.items()}


SGM = "landiff.diffusion.sgm.modules.diffusionmodules."


# ------------------------------------------------------------------------------------------
def gen_schedule_and_sampler():
    from landiff.diffusion.sgm.modules.diffusionmodules.denoiser import DiscreteDenoiser
    from landiff.diffusion.sgm.modules.diffusionmodules.sampling import VideoDDIMSampler, VPSDEDPMPP2MSampler
    from landiff.diffusion.sgm.modules.diffusionmodules.util import timestep_embedding

    disc = {"target": SGM + "discretizer.ZeroSNRDDPMDiscretization", "params": {"shift_scale": 3.0}}

    def make(cls, n):
        return cls(num_steps=n, discretization_config=disc, device="cpu", verbose=False,
                   guider_config={"target": SGM + "guiders.DynamicCFG", "params": {"scale": 6, "exp": 5, "num_steps": n}})

    den = DiscreteDenoiser(weighting_config={"target": SGM + "denoiser_weighting.EpsWeighting"},
                           scaling_config={"target": SGM + "denoiser_scaling.VideoScaling"}, num_idx=1000,
                           discretization_config=disc, quantize_c_noise=False)
    s50 = make(VPSDEDPMPP2MSampler, 50)
    x0 = torch.zeros(1, 2, 2, 2, 2)
    _, _, a50, _, _, _, ts50 = s50.prepare_sampling_loop(x0, {}, {})
    q = den.possibly_quantize_sigma(a50[:-1])
    qidx = den.sigma_to_idx(a50[:-1])
    scales = [s50.guider.scale_schedule(None, int(50 - t)) for t in ts50.tolist()[::-1][:50]]
    temb = timestep_embedding(torch.tensor([999.0, 19.0, 500.0]), 64)
    save("schedule", alpha_cumprod_sqrt=a50, timesteps=ts50, denoiser_sigmas=den.sigmas, quantized=q, quantized_idx=qidx,
         cfg_scales=np.array(scales, dtype=np.float64), timestep_embedding=temb)

    # trajectories with an analytic "network" (pins multipliers, sigma quantisation, CFG order, RNG call order)
    def network(x, t, c, **kw):
        ctx = c["crossattn"].mean(dim=(1, 2)).view(-1, 1, 1, 1, 1)
        return 0.6 * x * torch.cos(t * 0.01).view(-1, 1, 1, 1, 1) + 0.1 * ctx + 0.05 * torch.sin(x * 3.0)

    out = {}
    for name, cls, n in (("vpsde", VPSDEDPMPP2MSampler, 50), ("vpsde7", VPSDEDPMPP2MSampler, 7), ("ddim", VideoDDIMSampler, 10),
                         ("vpsde_fixed2", VPSDEDPMPP2MSampler, 6)):
        smp = make(cls, n)
        if name == "vpsde_fixed2":          # streaming primitive: the first 2 latent frames are pinned (sampling.py:800-835)
            smp.fixed_frames = 2
        torch.manual_seed(1234)
        x = torch.randn(1, 3, 4, 4, 6)
        cond = {"crossattn": torch.randn(1, 5, 8)}
        uc = {"crossattn": torch.zeros(1, 5, 8)}
        out[name + "_x0"] = x.clone()
        out[name + "_cond"] = cond["crossattn"].clone()
        denoiser = lambda inp, sigma, c, **kw: den(network, inp, sigma, c, **kw)
        res = smp(denoiser, x.clone(), cond, uc=uc)
        out[name + "_out"] = res
    save("sampler", **out)


# ------------------------------------------------------------------------------------------
def gen_rope_and_mask():
    from landiff.modules.pos_emb import Rope1DPosEmb, Rope3DPosEmb, apply_rope
    from landiff.tokenizer.modules import flex_attention_mask as fam

    r1 = Rope1DPosEmb(dim=128, max_len=64, device="cpu")
    f1 = r1.get_freqs_cis_by_seqlens([40])
    torch.manual_seed(0)
    q = torch.randn(1, 40, 2, 128)
    k = torch.randn(1, 40, 2, 128)
    qo, ko = apply_rope(q, k, f1[None])
    # 3D table exactly as TiTokDecoder.freqs_cis builds it (blocks.py:862-904), tiny grid
    T, H, W, nI, nP = 3, 4, 6, 6, 3
    r3 = Rope3DPosEmb(dim=64, max_time=100, max_height=30, max_width=45, one_dim_max_time=1000, multiple=16, device="cpu")
    vis = Rope3DPosEmb.shape_to_index(T, H, W, device="cpu")
    vis, _ = Rope3DPosEmb.shift_rope_index(vis, 0)
    lat = Rope3DPosEmb.len_to_rope_index(nI + (T - 1) * nP, device="cpu")
    lat, _ = Rope3DPosEmb.shift_rope_index(lat, 0, shift_all=True)
    idx = torch.cat([vis, lat], 0)
    f3 = r3.get_freqs_cis_by_idx(idx, torch.ones_like(idx[..., 0], dtype=torch.bool))
    save("rope", f1_real=f1.real, f1_imag=f1.imag, q=q, k=k, q_out=qo, k_out=ko, f3_real=f3.real, f3_imag=f3.imag,
         grid=np.array([T, H, W, nI, nP]))

    # decoder mask: tiny dense (scalar _mask_fn, the reference's own oracle) and full size via its vectorised fn
    def make_mask_obj(T, tpf, nI, nP):
        m = object.__new__(fam.VideoDecoderMask)
        m.num_frames, m.tokens_per_frame, m.IFrame_tokens, m.PFrame_tokens = T, tpf, nI, nP
        m.seq_len = T * tpf + nI + nP * (T - 1)
        m.block_size, m.device = 128, torch.device("cpu")
        return m

    m = make_mask_obj(4, 6, 5, 3)
    n = m.seq_len
    dense = np.zeros((n, n), dtype=bool)
    for qi in range(n):
        for ki in range(n):
            dense[qi, ki] = bool(m._mask_fn(1, 1, qi, ki))
    mf = make_mask_obj(13, 1350, 330, 74)
    L = mf.seq_len
    Lp = (L + 127) // 128 * 128
    qi = torch.arange(Lp)[:, None]
    ki = torch.arange(Lp)[None, :]
    h = hashlib.sha256()
    occ = np.zeros((Lp // 128, Lp // 128), dtype=np.int32)
    for r0 in range(0, Lp, 128):                     # 128 query rows at a time (keeps memory small)
        blk = mf.vmap_fn(None, None, qi[r0:r0 + 128], ki).numpy().astype(np.uint8)
        h.update(np.packbits(blk, axis=None).tobytes())
        occ[r0 // 128] = blk.reshape(128, Lp // 128, 128).sum(axis=(0, 2))
    save("decoder_mask", tiny_dense=dense, tiny_cfg=np.array([4, 6, 5, 3]), full_cfg=np.array([13, 1350, 330, 74]),
         full_sha256=np.frombuffer(h.digest(), dtype=np.uint8), full_block_occupancy=occ)


# ------------------------------------------------------------------------------------------
def gen_llm():
    import torch.nn.functional as F
    from landiff.llm.models.lm_model import Semantic1DLM
    from landiff.llm.models.transformer import GPT
    from landiff.llm.modules.conditioner import MicroConditioner, TextCond
    from landiff.llm.modules.transformer_blocks import LlamaTransformerBlock
    from landiff.modules.pos_emb import Rope1DPosEmb
    from landiff_amd.config import LLMConfig
    from landiff_amd.weights import init_state, llm_spec

    cfg = LLMConfig.tiny()

    class FakeTok(nn.Module):
        segment_length, segment_stride = cfg.segment_length, cfg.segment_stride
        def vocab_size(self): return cfg.visual_vocab
        def encode_codes(self, visual):           # use_gt_first_frame run: inputs["video"] carries the token ids themselves
            return [visual.long()]

    class FakeT5(nn.Module):
        dimension, max_length = cfg.text_dim, 512
        def __init__(self, dt):
            super().__init__()
            self.fwd_dtype = dt
            self.n = 0
        def tokenize_padded(self, x):
            return types.SimpleNamespace(input_ids=torch.zeros(len(x), self.n, dtype=torch.long),
                                         attention_mask=torch.ones(len(x), self.n, dtype=torch.long))

    for tag, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        torch.manual_seed(7)
        blocks = [LlamaTransformerBlock(num_heads=cfg.heads, hidden_dim=cfg.hidden, mlp_dim=cfg.mlp,
                                        activation=nn.GELU(approximate="tanh"), drop_path=0.0) for _ in range(cfg.num_layers)]
        gpt = GPT(cfg.vocab, hidden_dim=cfg.hidden, causal=True, fwd_dtype=dt, blocks=blocks,
                  rope=Rope1DPosEmb(dim=cfg.head_dim, theta_base=10000, max_len=512, device="cpu"))
        t5 = FakeT5(dt)
        cond = TextCond(text_encoder=t5, max_cond_tokens_num=512, embed_dim=cfg.hidden, padding=False,
                        freeze_text_encoder=True, cfg_drop_prob=0.1, use_mlp_embeddings=True)
        micro = MicroConditioner(out_dim=cfg.hidden, hidden_dim=cfg.micro_hidden, frequency_embedding_size=cfg.freq_dim,
                                 crossattn_condition_keys=("frames", "motion_score"), fwd_dtype=dt,
                                 defaults={"frames": 1, "motion_score": 0})
        model = Semantic1DLM(tokenizer=FakeTok(), cond_model=cond, transformer=gpt, micro_condition=micro,
                             Iframe_len=cfg.iframe_len, Pframe_len=cfg.pframe_len, fwd_dtype=dt)
        # weights: the build's own seeded synthetic state dict, loaded STRICTLY into the reference module --
        # this also pins the state-dict key map against the reference's classes
        sd = init_state(llm_spec(cfg), seed=5)
        model.load_state_dict(sd, strict=True)
        model.eval()
        g = torch.Generator().manual_seed(11)
        n_text = 9
        t5.n = n_text
        text = torch.randn(n_text, cfg.text_dim, generator=g)
        logits_log = []
        orig = gpt.sample
        def rec(x, freqs_cis=None, _o=orig):
            out = _o(x, freqs_cis=freqs_cis)
            logits_log.append(out.detach().float().clone())
            return out
        gpt.sample = rec
        fed = []
        emb_fwd = model.visual_embedding_model.forward
        def rec_emb(x, _o=emb_fwd):
            fed.append(x.detach().clone().reshape(-1))
            return _o(x)
        model.visual_embedding_model.forward = rec_emb
        inputs = {"caption": ["x"], "caption_embedding": [text], "motion_score": torch.tensor([0.1]),
                  "frames": torch.tensor([float(cfg.segment_length)])}
        torch.manual_seed(42)
        codes = model.sample(inputs, guidance_scale=7.5, temperature=1.0, seed=None, num_frames=cfg.segment_length)
        logits = torch.cat(logits_log, 0).clone()       # [steps*2, V]: rows (cond, uncond) per step
        fed_tokens = torch.cat(fed).clone()             # token fed back after each step (unclamped, forced incl.)
        extra = {}
        if tag == "fp32":   # a 2-segment run pins the multi-segment forced-token schedule
            logits_log.clear()
            inputs["frames"] = torch.tensor([float(2 * cfg.segment_length)])
            torch.manual_seed(43)
            extra["codes_2seg"] = model.sample(inputs, guidance_scale=7.5, temperature=1.0, seed=None,
                                               num_frames=2 * cfg.segment_length)
        save(f"llm_{tag}", text=text, codes=codes, logits=logits, fed_tokens=fed_tokens, seed=np.array(5), **extra)
        if tag == "fp32":   # use_gt_first_frame (lm_model.py:332-352): the I frame of a given token stream joins the prefix
            logits_log.clear()
            n_codes = cfg.iframe_len + (cfg.segment_length - 1) * cfg.pframe_len
            gt = torch.randint(0, cfg.visual_vocab, (n_codes,), generator=g)
            inputs["frames"] = torch.tensor([float(cfg.segment_length)])
            inputs["video"] = [gt.float()]
            torch.manual_seed(44)
            codes_gt = model.sample(inputs, guidance_scale=7.5, temperature=1.0, seed=None, num_frames=cfg.segment_length,
                                    use_gt_first_frame=True)
            save("llm_fp32_gt_first_frame", text=text, gt=gt, codes=codes_gt, logits=torch.cat(logits_log, 0).clone(), seed=np.array(5))
            # top-k / top-p filtering of the unrestricted positions (lm_model.py:441-447, top_p_probability)
            del inputs["video"]
            torch.manual_seed(45)
            codes_k = model.sample(inputs, guidance_scale=7.5, temperature=1.0, seed=None, num_frames=cfg.segment_length, top_k=5)
            torch.manual_seed(46)
            codes_p = model.sample(inputs, guidance_scale=7.5, temperature=0.7, seed=None, num_frames=cfg.segment_length, top_p=0.8)
            save("llm_fp32_topk_topp", text=text, codes_top_k5=codes_k, codes_top_p08_t07=codes_p, seed=np.array(5))


# ------------------------------------------------------------------------------------------
def gen_titok():
    import torch.nn.attention.flex_attention as fa
    from landiff.modules.pos_emb import Rope3DPosEmb
    from landiff.tokenizer.modules.blocks import AttentionImp, AttentionMaskType, PositionalEmbedingType, TiTokDecoder
    from landiff.tokenizer.modules import flex_attention_mask as fam
    from landiff_amd.config import TokenizerConfig
    from landiff_amd.weights import init_state, tokenizer_spec

    # harness adaptations (SURVEY 8c i,ii): create_mask defaults to device="cuda" and mis-classifies bound methods
    orig_create_mask = fa.create_mask
    def create_mask_cpu(fn, B, H, Q_LEN, KV_LEN, device="cpu", **kw):
        return orig_create_mask(lambda b, h, q, k: fn(b, h, q, k), B, H, Q_LEN, KV_LEN, device="cpu")
    fa.create_mask = create_mask_cpu
    fam.flex_attention_mod.create_mask = create_mask_cpu

    cfg = TokenizerConfig.tiny()
    rope = Rope3DPosEmb(dim=cfg.head_dim, max_time=100, max_height=30, max_width=45, one_dim_max_time=100000,
                        multiple=16, device="cpu")
    dec = TiTokDecoder(image_size=(cfg.grid_h, cfg.grid_w), image_channels=cfg.out_channels, patch_size=1,
                       model_size="base", width=cfg.width, num_layers=cfg.layers, num_heads=cfg.heads,
                       num_latent_tokens=cfg.num_latent_tokens, token_size=cfg.token_size,
                       output_channels=cfg.out_channels, use_checkpoint=False, qk_norm=False, bias=False,
                       causal=False, code_drop=False, positional_embedding_type=PositionalEmbedingType.ROPE_3D,
                       rope_layer=rope, attention_imp=AttentionImp.TORCH,
                       attention_mask_type=AttentionMaskType.VIDEO_DECODER_MASK, use_cls_token=False,
                       temporal_size=cfg.temporal, PFrame_tokens=cfg.pframe_tokens)
    sd = init_state(tokenizer_spec(cfg), seed=6)
    dec_sd = {k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}
    dec.load_state_dict(dec_sd, strict=True)
    dec.eval()
    torch.manual_seed(3)
    z = torch.randn(1, cfg.token_size, 1, cfg.num_latent_tokens)
    with torch.no_grad():
        out = dec(z)
    save("titok_fp32", z=z, out=out, seed=np.array(6))


def gen_feature_norm():
    """VideoVQ.norm_features / denorm_features (video_titok_vq.py:221-233) called as the reference's own unbound methods,
    once with mean_std_path=None (the shipped tokenizer_cfg.py: buffers present, functions the identity) and once with
    a path set: pins the gate, not just the formula."""
    from landiff.tokenizer.models.video_titok_vq import VideoVQ as TowDVQ
    g = torch.Generator().manual_seed(21)
    C = 8
    x = torch.randn(1, 3, C, 4, 6, generator=g)
    mean, std = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    off = types.SimpleNamespace(mean_std_path=None, mean=mean, std=std)
    on = types.SimpleNamespace(mean_std_path="stats.pt", mean=mean, std=std)
    save("feature_norm", x=x, mean=mean, std=std,
         norm_off=TowDVQ.norm_features(off, x), norm_on=TowDVQ.norm_features(on, x),
         denorm_off=TowDVQ.denorm_features(off, x), denorm_on=TowDVQ.denorm_features(on, x),
         denorm_on_bf16=TowDVQ.denorm_features(on, x.to(torch.bfloat16)).to(torch.bfloat16))


def gen_titok_encoder():
    """Encoder half: TiTokEncoder.forward on a tiny config + VideoEncoderMask (tiny dense through the
    reference's scalar _mask_fn; full size through its vectorised vmap_fn, hashed)."""
    import torch.nn.attention.flex_attention as fa
    from landiff.modules.pos_emb import Rope3DPosEmb
    from landiff.tokenizer.modules.blocks import AttentionImp, AttentionMaskType, PositionalEmbedingType, TiTokEncoder
    from landiff.tokenizer.modules import flex_attention_mask as fam
    from landiff_amd.config import TokenizerConfig
    from landiff_amd.weights import init_state, tokenizer_encoder_spec

    orig_create_mask = fa.create_mask
    def create_mask_cpu(fn, B, H, Q_LEN, KV_LEN, device="cpu", **kw):
        return orig_create_mask(lambda b, h, q, k: fn(b, h, q, k), B, H, Q_LEN, KV_LEN, device="cpu")
    fa.create_mask = create_mask_cpu
    fam.flex_attention_mod.create_mask = create_mask_cpu

    cfg = TokenizerConfig.tiny()
    rope = Rope3DPosEmb(dim=cfg.head_dim, max_time=100, max_height=30, max_width=45, one_dim_max_time=100000,
                        multiple=16, device="cpu")
    enc = TiTokEncoder(image_size=(cfg.grid_h, cfg.grid_w), image_channels=cfg.out_channels, patch_size=1,
                       model_size="base", num_latent_tokens=cfg.num_latent_tokens, token_size=cfg.token_size,
                       width=cfg.width, num_layers=cfg.layers, num_heads=cfg.heads, use_checkpoint=False, qk_norm=False,
                       causal=False, bias=False, use_cls_token=False, rope_layer=rope,
                       positional_embedding_type=PositionalEmbedingType.ROPE_3D, attention_imp=AttentionImp.TORCH,
                       attention_mask_type=AttentionMaskType.VIDEO_ENCODER_MASK, temporal_size=cfg.temporal,
                       PFrame_tokens=cfg.pframe_tokens, inside_latent_tokens=True)
    sd = init_state(tokenizer_encoder_spec(cfg), seed=8)
    enc_sd = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    enc.load_state_dict(enc_sd, strict=True)
    enc.eval()
    torch.manual_seed(4)
    x = torch.randn(1, cfg.temporal, cfg.out_channels, cfg.grid_h, cfg.grid_w)
    x = x + 2.0 * torch.randn(1, cfg.temporal, cfg.out_channels, 1, 1)           # frame-dependent content
    with torch.no_grad():
        out = enc(x, forward_T=cfg.temporal)                                       # [1, token_size, 1, L]
    save("titok_enc_fp32", x=x, out=out, seed=np.array(8))

    gen_feature_norm()

    def make_mask_obj(T, tpf, nI, nP):
        m = object.__new__(fam.VideoEncoderMask)
        m.num_frames, m.tokens_per_frame, m.IFrame_tokens, m.PFrame_tokens = T, tpf, nI, nP
        m.seq_len = T * tpf + nI + nP * (T - 1)
        m.block_size, m.device = 128, torch.device("cpu")
        return m

    m = make_mask_obj(4, 6, 5, 3)
    n = m.seq_len
    dense = np.zeros((n, n), dtype=bool)
    for qi in range(n):
        for ki in range(n):
            dense[qi, ki] = bool(m._mask_fn(1, 1, qi, ki))
    mf = make_mask_obj(13, 1350, 330, 74)
    L = mf.seq_len
    qi = torch.arange(L)[:, None]
    ki = torch.arange(L)[None, :]
    h = hashlib.sha256()
    rows = np.zeros(L, dtype=np.int64)
    for r0 in range(0, L, 128):
        blk = mf.vmap_fn(None, None, qi[r0:r0 + 128], ki).numpy().astype(np.uint8)
        h.update(np.packbits(blk, axis=None).tobytes())
        rows[r0:r0 + blk.shape[0]] = blk.sum(1)
    save("encoder_mask", tiny_dense=dense, tiny_cfg=np.array([4, 6, 5, 3]), full_cfg=np.array([13, 1350, 330, 74]),
         full_sha256=np.frombuffer(h.digest(), dtype=np.uint8), full_row_counts=rows)


def gen_upsampler():
    from landiff.diffusion.semantic_models.modules.vq_gan_blocks import Decoder
    from landiff_amd.config import UpsamplerConfig
    from landiff_amd.weights import init_state, upsampler_spec

    cfg = UpsamplerConfig.tiny()
    dec = Decoder(z_channels=cfg.z_channels, resolution=16, in_channels=512, out_ch=cfg.out_ch, ch=cfg.ch,
                  ch_mult=list(cfg.ch_mult), num_res_blocks=cfg.num_res_blocks, attn_resolutions=[], dropout=0.0,
                  use_mid_attention=False, upsample_type="pixelshuffle")
    sd = init_state(upsampler_spec(cfg), seed=7)
    dec.load_state_dict({k[len("upsample_model."):]: v for k, v in sd.items() if k.startswith("upsample_model.")}, strict=True)
    conv_out = nn.Conv2d(cfg.out_ch, cfg.target_dim, 3, 1, 1)
    conv_out.load_state_dict({"weight": sd["conv_out.weight"], "bias": sd["conv_out.bias"]})
    torch.manual_seed(4)
    x = torch.randn(3, cfg.z_channels, 4, 6)
    with torch.no_grad():
        up = dec(x)
        out = conv_out(up)
    save("upsampler_fp32", x=x, up=up, out=out, seed=np.array(7))


def gen_vae():
    import torch.distributed as dist
    from landiff.diffusion.vae_modules.cp_enc_dec import ContextParallelDecoder3D
    from landiff.diffusion.sgm.util import initialize_context_parallel
    from landiff_amd.config import VAEConfig
    from landiff_amd.weights import init_state, vae_spec

    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=0, world_size=1)
    initialize_context_parallel(1)
    cfg = VAEConfig.tiny()
    dec = ContextParallelDecoder3D(double_z=True, z_channels=cfg.z_channels, resolution=256, in_channels=3,
                                   out_ch=cfg.out_ch, ch=cfg.ch, ch_mult=list(cfg.ch_mult), attn_resolutions=[],
                                   num_res_blocks=cfg.num_res_blocks, dropout=0.0, gather_norm=False)
    sd = init_state(vae_spec(cfg), seed=8)
    dec.load_state_dict({k[len("decoder."):]: v for k, v in sd.items()}, strict=True)
    dec.eval()
    torch.manual_seed(5)
    T = 7
    latent = torch.randn(1, cfg.z_channels, T, 4, 6)
    z = latent / cfg.scale_factor
    outs = []
    with torch.no_grad():
        loop = (T - 1) // 2                                      # CogWrapper.decode_latent schedule
        for i in range(loop):
            a, b = (0, 3) if i == 0 else (i * 2 + 1, i * 2 + 3)
            outs.append(dec(z[:, :, a:b].contiguous(), clear_fake_cp_cache=(i == loop - 1)))
        chunked = torch.cat(outs, dim=2)
        full = dec(z.contiguous(), clear_fake_cp_cache=True)     # witness: chunked != full (per-chunk GroupNorm)
    save("vae_fp32", latent=latent, chunked=chunked, full_first9=full[:, :, :9], seed=np.array(8))


def gen_dit():
    """AdaLN layer logic from the reference's own mixin code, driven through a sat shim:
    the shim supplies sat's SelfAttention/MLP/LayerNorm as the build restates them."""
    import torch.nn.functional as F
    from landiff.diffusion import dit_video_concat as D
    from landiff.diffusion.sgm.modules.diffusionmodules.util import timestep_embedding
    from landiff_amd.config import DiTConfig
    from landiff_amd.weights import dit_spec, init_state

    cfg = DiTConfig.tiny()
    d, te, H, hd = cfg.hidden, cfg.time_embed_dim, cfg.heads, cfg.head_dim
    out = {}
    for control in (False, True):
        L = cfg.layers_control if control else cfg.layers_main
        sd = init_state(dit_spec(cfg, control), seed=9 + int(control))
        Mixin = D.ControlOutAdaLNMixin if control else D.ControlAdaLNMixin
        kw = dict(width=cfg.grid_w, height=cfg.grid_h, hidden_size=d, num_layers=L, time_embed_dim=te,
                  compressed_num_frames=cfg.latent_frames, qk_ln=True, hidden_size_head=hd, elementwise_affine=True)
        if control:
            kw["use_zero_linears"] = True
        else:
            kw["control_layers"] = cfg.layers_control
        mix = Mixin(**kw)
        mix.load_state_dict({k[len("mixins.adaln_layer."):]: v for k, v in sd.items() if k.startswith("mixins.adaln_layer.")}, strict=True)

        class Layer(nn.Module):                      # sat BaseTransformerLayer restated (shim)
            def __init__(self, i):
                super().__init__()
                p = f"transformer.layers.{i}."
                self.i = i
                self.input_layernorm = nn.LayerNorm(d, eps=cfg.block_ln_eps)
                self.post_attention_layernorm = nn.LayerNorm(d, eps=cfg.block_ln_eps)
                self.qkv, self.dense = nn.Linear(d, 3 * d), nn.Linear(d, d)
                self.h4, self.h1 = nn.Linear(d, 4 * d), nn.Linear(4 * d, d)
                for mod, nm in ((self.input_layernorm, "input_layernorm"), (self.post_attention_layernorm, "post_attention_layernorm"),
                                (self.qkv, "attention.query_key_value"), (self.dense, "attention.dense"),
                                (self.h4, "mlp.dense_h_to_4h"), (self.h1, "mlp.dense_4h_to_h")):
                    mod.load_state_dict({"weight": sd[p + nm + ".weight"], "bias": sd[p + nm + ".bias"]})
            def attention(self, x, mask, **kw):
                B, N, _ = x.shape
                q, k, v = self.qkv(x).chunk(3, dim=-1)
                sh = lambda t: t.view(B, N, H, hd).permute(0, 2, 1, 3)
                sdpa = lambda q, k, v, m, **_: F.scaled_dot_product_attention(q, k, v)
                o = mix.attention_fn(sh(q), sh(k), sh(v), mask, old_impl=sdpa, **kw)
                return self.dense(o.permute(0, 2, 1, 3).reshape(B, N, d))
            def mlp(self, x, **kw):
                return self.h1(F.gelu(self.h4(x), approximate="tanh"))

        tr = types.SimpleNamespace(layers=[Layer(i) for i in range(L)], layernorm_order="pre")
        object.__setattr__(mix, "transformer", tr)
        torch.manual_seed(20 + int(control))
        B, N = 2, cfg.text_len + 24
        h = torch.randn(B, N, d)
        emb = torch.randn(B, te)
        ctrl = [torch.randn(B, N, d) * 0.3 for _ in range(cfg.layers_control)]
        hs = [h]
        with torch.no_grad():
            for i in range(L):
                kwargs = dict(layer_id=i, emb=emb, text_length=cfg.text_len)
                if not control:
                    kwargs["control_layers_output"] = [{"hidden_states": c} for c in ctrl]
                hs.append(mix.layer_forward(hs[-1], None, **kwargs))
        tag = "control" if control else "main"
        out[f"{tag}_h"], out[f"{tag}_emb"], out[f"{tag}_out"] = h, emb, torch.stack(hs[1:])
        if not control:
            out["main_ctrl"] = torch.stack(ctrl)
            # final layer (reference FinalLayerMixin code) + patch embed + pos table + unpatchify
            fin = D.FinalLayerMixin(hidden_size=d, time_embed_dim=te, patch_size=cfg.patch, out_channels=cfg.out_channels,
                                    latent_width=cfg.latent_w, latent_height=cfg.latent_h, elementwise_affine=True)
            fin.load_state_dict({k[len("mixins.final_layer."):]: v for k, v in sd.items() if k.startswith("mixins.final_layer.")}, strict=True)
            n_img = cfg.latent_frames * cfg.grid_h * cfg.grid_w
            hx = torch.randn(B, cfg.text_len + n_img, d)
            with torch.no_grad():
                out["final_in"], out["final_out"] = hx, fin.final_forward(hx, text_length=cfg.text_len, emb=emb)
            pe = D.ImagePatchEmbeddingMixin(cfg.in_channels, d, cfg.patch, text_hidden_size=cfg.text_dim)
            pe.load_state_dict({k[len("mixins.patch_embed."):]: v for k, v in sd.items() if k.startswith("mixins.patch_embed.")}, strict=True)
            img = torch.randn(B, cfg.latent_frames, cfg.in_channels, cfg.latent_h, cfg.latent_w)
            ctx = torch.randn(B, cfg.text_len, cfg.text_dim)
            with torch.no_grad():
                out["embed_img"], out["embed_ctx"] = img, ctx
                out["embed_out"] = pe.word_embedding_forward(None, images=img, encoder_outputs=ctx)
            pos = D.get_3d_sincos_pos_embed(d, cfg.grid_h, cfg.grid_w, cfg.latent_frames, height_interpolation=1.875,
                                            width_interpolation=1.875)
            out["pos_embed"] = pos.astype(np.float32)
    save("dit_fp32", **out)


# ------------------------------------------------------------------------------------------
def main():
    install_stubs()
    which = sys.argv[1:] or ["schedule", "rope", "llm", "titok", "titok_enc", "ups", "vae", "dit"]
    if "schedule" in which:
        gen_schedule_and_sampler()
    if "rope" in which:
        gen_rope_and_mask()
    if "llm" in which:
        gen_llm()
    if "titok" in which:
        gen_titok()
    if "titok_enc" in which:
        gen_titok_encoder()
    if "ups" in which:
        gen_upsampler()
    if "vae" in which:
        gen_vae()
    if "dit" in which:
        gen_dit()


if __name__ == "__main__":
    main()

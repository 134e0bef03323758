"""GPU parity of the tokenizer encoder half (VideoVQ.encode_to_index(features=...)) against the oracle,
which is itself pinned to the reference's TiTokEncoder / VideoEncoderMask by tests/test_oracle_golden.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfgs():
    from landiff_amd.config import TokenizerConfig
    tiny = TokenizerConfig.tiny()
    # several q-blocks and key tiles per group, frames that straddle tile boundaries
    mid = TokenizerConfig(width=128, layers=2, heads=2, grid_h=6, grid_w=11, temporal=4, pframe_tokens=21,
                          num_latent_tokens=45 + 3 * 21, codebook_size=256, codebook_dim=16, token_size=128, out_channels=128)
    return {"tiny": tiny, "mid": mid}


def _features(cfg, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(cfg.temporal, cfg.out_channels, cfg.grid_h, cfg.grid_w, generator=g)
    return x + 2.0 * torch.randn(cfg.temporal, cfg.out_channels, 1, 1, generator=g)


@pytest.mark.parametrize("name", ["tiny", "mid"])
def test_encoder_matches_oracle(cuda, name):
    from landiff_amd.tokenizer_encoder import TokenizerEncoder
    from landiff_amd.weights import init_state, tokenizer_encoder_spec
    from oracle.tokenizer import TokenizerEncoderOracle
    cfg = _cfgs()[name]
    sd = init_state(tokenizer_encoder_spec(cfg), 17)
    x = _features(cfg, 5)
    orc = TokenizerEncoderOracle(sd, cfg, torch.bfloat16)
    ref = orc.encode(orc.norm_features(x[None]))[0].float()                     # [L, token_size]
    enc = TokenizerEncoder(sd, cfg, cuda)
    out = enc.encode(x.to(cuda)).float().cpu()
    err = (out - ref).abs().max().item() / (ref.abs().max().item() + 1e-6)
    assert err < 3e-2, err                                                       # bf16 flow, different accumulation order
    # the fp32 oracle is the tighter yardstick: the HIP path must not be further from it than the bf16 oracle is (x2)
    o32 = TokenizerEncoderOracle(sd, cfg, torch.float32)
    ref32 = o32.encode(o32.norm_features(x[None]))[0]
    d_hip = (out - ref32).abs().max().item()
    d_bf = (ref - ref32).abs().max().item()
    assert d_hip < 2.0 * d_bf + 1e-3, (d_hip, d_bf)
    # indices: equal wherever the oracle's best code wins by a clear margin
    idx = enc.nearest_code(enc.encode(x.to(cuda))).cpu()
    z = orc.encode(orc.norm_features(x[None]))[0]
    zc = torch.nn.functional.linear(z.float(), sd["quantizer.project_in.weight"].to(torch.bfloat16).float(),
                                    sd["quantizer.project_in.bias"].to(torch.bfloat16).float()).to(torch.bfloat16).float()
    e = sd["quantizer._codebook.embed"][0].float()
    d2 = torch.cdist(zc, e) ** 2
    top2 = d2.topk(2, dim=1, largest=False)
    clear = (top2.values[:, 1] - top2.values[:, 0]) > 0.05 * top2.values[:, 1]
    assert clear.float().mean().item() >= 0.25, clear.float().mean().item()     # (tiny: 12 tokens -- enough clear winners to mean something)
    assert torch.equal(idx[clear], top2.indices[:, 0][clear])


def test_norm_features_gate(cuda):
    """The checkpoint's mean/std buffers are applied only when the config names a mean_std_path (video_titok_vq.py:221-233;
    the shipped tokenizer_cfg.py does not): with the flag off, non-trivial statistics in the state dict must not change a
    single token id; with it on, encoder and decoder follow the reference's formulas (golden: its own two methods)."""
    import dataclasses, os
    from landiff_amd import ops
    from landiff_amd.tokenizer_encoder import TokenizerEncoder
    from landiff_amd.weights import init_state, tokenizer_encoder_spec
    from oracle.tokenizer import TokenizerEncoderOracle
    cfg = _cfgs()["mid"]
    sd = init_state(tokenizer_encoder_spec(cfg), 17)
    g = torch.Generator().manual_seed(3)
    sd_stats = dict(sd, mean=torch.randn(cfg.out_channels, generator=g), std=torch.rand(cfg.out_channels, generator=g) + 0.5)
    sd_plain = dict(sd, mean=torch.zeros(cfg.out_channels), std=torch.ones(cfg.out_channels))
    x = _features(cfg, 5).to(cuda)
    z_off = TokenizerEncoder(sd_stats, cfg, cuda).encode(x)
    z_plain = TokenizerEncoder(sd_plain, cfg, cuda).encode(x)
    assert torch.equal(z_off, z_plain)                                           # buffers ignored: bit-identical latents
    assert torch.equal(TokenizerEncoder(sd_stats, cfg, cuda).encode_to_index(x), TokenizerEncoder(sd_plain, cfg, cuda).encode_to_index(x))
    on = dataclasses.replace(cfg, norm_features=True)
    z_on = TokenizerEncoder(sd_stats, on, cuda).encode(x).float().cpu()
    assert (z_on - z_off.float().cpu()).abs().max().item() > 1e-2                # the flag does change the input
    orc = TokenizerEncoderOracle(sd_stats, on, torch.bfloat16)
    ref = orc.encode(orc.norm_features(x.cpu()[None]))[0].float()
    assert (z_on - ref).abs().max().item() / (ref.abs().max().item() + 1e-6) < 3e-2
    # decoder side: ld_feature_denorm against the reference's denorm_features (bf16 features, fp32 buffers, cast back to bf16)
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "feature_norm.npz"))
    xg = torch.from_numpy(gold["x"]).to(torch.bfloat16).permute(0, 1, 3, 4, 2).contiguous()          # channels-last
    C = xg.shape[-1]
    out = ops.feature_denorm(xg.reshape(-1, C).to(cuda), torch.from_numpy(gold["mean"]).to(cuda), torch.from_numpy(gold["std"]).to(cuda),
                             out=torch.empty(xg.numel() // C, C, device=cuda, dtype=torch.bfloat16))
    want = torch.from_numpy(gold["denorm_on_bf16"]).permute(0, 1, 3, 4, 2).reshape(-1, C)
    assert torch.equal(out.float().cpu(), want)


def test_feature_norm_and_vq_kernels(cuda):
    from landiff_amd import ops
    g = torch.Generator().manual_seed(1)
    T, C, H, W = 3, 200, 7, 13
    x = torch.randn(T, C, H, W, generator=g) * 3
    mean, std = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    ref = ((x.permute(0, 2, 3, 1) - mean) / (std + 1e-8)).reshape(-1, C).to(torch.bfloat16)
    for dt in (torch.float32, torch.bfloat16):
        out = torch.empty(T * H * W, C, device=cuda, dtype=torch.bfloat16)
        ops.feature_norm_cl(x.to(cuda, dt), mean.to(cuda), std.to(cuda), out, T, C, H * W)
        r = ref if dt == torch.float32 else ((x.to(dt).float().permute(0, 2, 3, 1) - mean) / (std + 1e-8)).reshape(-1, C).to(torch.bfloat16)
        assert torch.equal(out.cpu(), r)
    # nearest code: exact on well separated data, lowest index on exact ties
    V, dim, L = 300, 16, 1000
    e = torch.randn(V, dim, generator=g)
    e[17] = e[5]                                                       # duplicate code: index 5 must win
    pick = torch.randint(0, V, (L,), generator=g)
    xq = (e[pick] + 0.01 * torch.randn(L, dim, generator=g)).to(torch.bfloat16)
    xpad = torch.zeros(L, 64, dtype=torch.bfloat16); xpad[:, :dim] = xq
    idx = torch.empty(L, device=cuda, dtype=torch.int64)
    ops.vq_nearest(xpad.to(cuda), e.to(cuda), idx, dim)
    want = pick.clone(); want[want == 17] = 5
    d2 = torch.cdist(xq.float(), e) ** 2
    top2 = d2.topk(2, dim=1, largest=False)
    clear = (top2.values[:, 1] - top2.values[:, 0] > 1e-3) | (pick == 5) | (pick == 17)
    assert torch.equal(idx.cpu()[clear], want[clear])
    assert clear.float().mean().item() > 0.9


def test_encoder_round_trip_through_decoder_shapes(cuda):
    """encode_to_index yields ids the detokenizer accepts: [L] int64 in [0, codebook_size)."""
    from landiff_amd.tokenizer_encoder import TokenizerEncoder
    from landiff_amd.weights import init_state, tokenizer_encoder_spec
    cfg = _cfgs()["tiny"]
    enc = TokenizerEncoder(init_state(tokenizer_encoder_spec(cfg), 2), cfg, cuda)
    ids = enc.encode_to_index(_features(cfg, 9).to(cuda))
    assert ids.shape == (cfg.num_latent_tokens,) and ids.dtype == torch.int64
    assert int(ids.min()) >= 0 and int(ids.max()) < cfg.codebook_size

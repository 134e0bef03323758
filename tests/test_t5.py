"""T5 text encoders.  The checker is the third-party implementation the reference itself calls
(HF transformers T5EncoderModel, landiff/llm/modules/text_encoder.py:36-42), instantiated with seeded random weights."""
import pytest
import torch


def test_relative_buckets_match_transformers():
    from transformers.models.t5.modeling_t5 import T5Attention
    from landiff_amd.t5 import relative_buckets
    for n in (1, 2, 17, 226, 512):
        for nb, md in ((32, 128), (16, 64)):
            ctx = torch.arange(n)[:, None]
            mem = torch.arange(n)[None, :]
            ref = T5Attention._relative_position_bucket(mem - ctx, bidirectional=True, num_buckets=nb, max_distance=md)
            got = relative_buckets(n, nb, md)
            idx = (mem - ctx) + n - 1
            assert torch.equal(got[idx].long(), ref), (n, nb, md)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [5, 64, 226, 300])
def test_t5_encoder_matches_transformers(cuda, n):
    from transformers import T5EncoderModel
    from landiff_amd.t5 import T5Config, T5EncoderRunner
    cfg = T5Config.tiny()
    torch.manual_seed(7)
    hf32 = T5EncoderModel(cfg.hf()).eval()
    with torch.no_grad():     # HF initialises the layer norms to 1 and the tables small: randomise so every path matters
        for name, prm in hf32.named_parameters():
            if "layer_norm" in name:
                prm.copy_(1.0 + 0.2 * torch.randn_like(prm))
            elif "relative_attention_bias" in name:
                prm.copy_(torch.randn_like(prm))
    sd = hf32.state_dict()
    ids = torch.randint(0, cfg.vocab, (n,), generator=torch.Generator().manual_seed(n))
    with torch.no_grad():
        ref32 = hf32(input_ids=ids[None]).last_hidden_state[0]
        hf16 = T5EncoderModel(cfg.hf()).eval().to(torch.bfloat16)
        hf16.load_state_dict({k: v.to(torch.bfloat16) for k, v in sd.items()})
        ref16 = hf16(input_ids=ids[None]).last_hidden_state[0].float()
    run = T5EncoderRunner(sd, cfg, cuda)
    out = run.encode(ids.to(cuda)).float().cpu()
    assert out.shape == (n, cfg.d_model)
    scale = ref32.abs().max().item()
    floor = (ref16 - ref32).abs().max().item() / scale          # the bf16 HF module's own distance from fp32
    err = (out - ref32).abs().max().item() / scale
    assert err < max(2 * floor, 2e-2), (err, floor)
    # and close to the bf16 module itself (same rounding points, different summation order)
    assert (out - ref16).abs().max().item() / scale < max(2 * floor, 2e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [226, 512])
def test_t5_encoder_layer_at_xxl_width_matches_transformers(cuda, n):
    """ONE encoder block at the real width of T5-v1.1-XXL / FLAN-T5-XXL (d_model 4096, 64 heads x 64, d_ff 10 240, 32 buckets)
    plus the final norm, at the two sequence lengths the pipeline feeds it: 226 (the DiT context, padded, unmasked:
    landiff/diffusion/sgm/modules/encoders/modules.py:246-295) and 512 (the LLM condition's cap, landiff/llm/modules/
    text_encoder.py:16-146).  Checker = transformers.T5EncoderModel on seeded random weights in fp32; 2x-floor rule against the
    bf16 HF module's own distance.  What only appears at this width: ld_t5_attn over 64 heads with a 4096-wide row stride, the
    fused q|k|v GEMM at N = 12 288, the gated-gelu pair at N = 10 240 (256 x 256 tiles instead of the tiny config's 128 x 128)."""
    from transformers import T5EncoderModel
    from landiff_amd.t5 import T5Config, T5EncoderRunner
    cfg = T5Config(vocab=512, layers=1)
    torch.manual_seed(11)
    hf32 = T5EncoderModel(cfg.hf()).eval()
    with torch.no_grad():
        for name, prm in hf32.named_parameters():
            if "layer_norm" in name:
                prm.copy_(1.0 + 0.2 * torch.randn_like(prm))
            elif "relative_attention_bias" in name:
                prm.copy_(torch.randn_like(prm))
    sd = hf32.state_dict()
    ids = torch.randint(0, cfg.vocab, (n,), generator=torch.Generator().manual_seed(n))
    with torch.no_grad():
        ref32 = hf32(input_ids=ids[None]).last_hidden_state[0]
        hf16 = T5EncoderModel(cfg.hf()).eval().to(torch.bfloat16)
        hf16.load_state_dict({k: v.to(torch.bfloat16) for k, v in sd.items()})
        ref16 = hf16(input_ids=ids[None]).last_hidden_state[0].float()
    run = T5EncoderRunner(sd, cfg, cuda)
    out = run.encode(ids.to(cuda)).float().cpu()
    assert out.shape == (n, cfg.d_model) and torch.isfinite(out).all()
    scale = ref32.abs().max().item()
    floor = (ref16 - ref32).abs().max().item() / scale
    err = (out - ref32).abs().max().item() / scale
    print(f"T5 block at XXL width, n = {n}: err {err:.4f}, bf16-HF floor {floor:.4f}")
    assert err < max(2 * floor, 2e-2), (err, floor)
    assert (out - ref16).abs().max().item() / scale < max(2 * floor, 2e-2)

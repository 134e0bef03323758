"""Tests of the measured alternatives that only the VARIANTS build of the library carries (landiff_amd/csrc/build.sh with
LD_BUILD_VARIANTS=1 -> landiff_amd/variants/liblandiff_hip_variants.so): the 128-query-row one-wave-per-SIMD attention tile, the
chained / persistent forms of the decode step and the 512 x 128 tile of the 8-phase GEMM loop.  Not collected by the normal run (the file name does not match test_*.py):
tests/test_gpu_variants.py::test_variants_build_suite runs this file in a child pytest process with LANDIFF_HIP_LIB pointing
at the variants library, so that the processes of the normal suite only ever map the shipped library."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from test_gpu_attn import _last_kernel, _run  # noqa: E402

pytestmark = pytest.mark.gpu


def test_this_process_runs_on_the_variants_library():
    from landiff_amd import _lib
    assert os.path.samefile(_lib.LIB_PATH, _lib.VARIANTS_LIB_PATH) and _lib.has_variants()


@pytest.fixture(scope="module")
def setup(cuda):
    from landiff_amd.config import PipelineConfig
    from landiff_amd.weights import init_pipeline_state
    cfg = PipelineConfig.tiny(num_steps=3).check()
    states = init_pipeline_state(cfg, seed=1234)
    return cfg, states


# ---- the 128-query-row / one-wave-per-SIMD tile (ld_attn_q128.hip; LD_ATTN_Q128 is read on every call) ----
@pytest.fixture
def q128(monkeypatch):
    monkeypatch.setenv("LD_ATTN_Q128", "2")        # 2: every unmasked problem of >= 6 key tiles, whatever its size


@pytest.mark.parametrize("B,H,N", [(1, 2, 1122), (2, 1, 1400), (1, 1, 2175), (1, 1, 1152), (1, 1, 384), (1, 1, 385), (1, 2, 448),
                                   (1, 1, 500), (1, 1, 512), (2, 1, 575), (1, 1, 640), (1, 1, 700), (1, 1, 768), (1, 1, 830),
                                   (1, 3, 2600)])       # 2600: two full 512-row workgroups per head + ragged rows in every wave of the last
def test_attn_q128_tile(cuda, q128, B, H, N):
    assert _run(cuda, B, H, N, seed=N) < 2e-2
    assert _last_kernel() == "ld_attn_q128_kernel"


def test_attn_q128_spike_and_overflow_fallback(cuda, q128):
    assert _run(cuda, 1, 2, 1400, spike=True, relative=True) < 1e-2
    err = _run(cuda, 1, 2, 1122, spike=True, q_scale=6.0, relative=True)      # the fast pass must notice and redo with the running max
    assert err == err and err < 0.2
    assert _last_kernel() == "ld_attn_q128_kernel"


def test_attn_q128_equals_q64_bit_for_bit_in_process(cuda, monkeypatch):
    """Same inputs through both wave tiles in one process (the knob is read per call): identical bits, ragged tail included."""
    from landiff_amd import ops
    g = torch.Generator(device=cuda).manual_seed(7)
    B, H, N = 2, 3, 3001
    Npad = (N + 127) // 128 * 128
    q = torch.zeros(B, H, Npad, 64, device=cuda, dtype=torch.bfloat16); k = torch.zeros_like(q)
    vt = torch.zeros(B, H, 64, Npad, device=cuda, dtype=torch.bfloat16)
    q[:, :, :N] = torch.randn(B, H, N, 64, device=cuda, generator=g).to(torch.bfloat16)
    k[:, :, :N] = torch.randn(B, H, N, 64, device=cuda, generator=g).to(torch.bfloat16)
    vt[:, :, :, :N] = torch.randn(B, H, 64, N, device=cuda, generator=g).to(torch.bfloat16)
    outs, names = [], []
    for knob in ("0", "2"):
        monkeypatch.setenv("LD_ATTN_Q128", knob)
        out = torch.zeros(B, N, H * 64, device=cuda, dtype=torch.bfloat16)
        ops.attn_fwd(q, k, vt, out, N, N, 0.125)
        outs.append(out); names.append(_last_kernel())
    assert names == ["ld_attn_q64_kernel", "ld_attn_q128_kernel"], names
    assert torch.equal(outs[0], outs[1])


def test_llm_chained_and_fused_blocks_equal_per_operation_chain_tiny(cuda, setup):
    """The two other forms of a decode step's blocks -- dependent launches on two streams (ld_llm_decode_blocks_chained) and one
    persistent launch with grid barriers (ld_llm_decode_forward_fused) -- against the per-operation chain on the tiny config: ids,
    every step's CFG logits and the KV cache bit for bit."""
    from landiff_amd.llm import LLMRunner
    cfg, st = setup
    c = cfg.llm
    text = torch.randn(5, c.text_dim, generator=torch.Generator().manual_seed(19))
    run = LLMRunner(st["llm"], c, cuda, max_text=32, max_frames=c.segment_length)
    assert run.fused_supported and run.chained_supported
    res = {}
    for mode in ("chain", "chained", "fused"):
        log = []
        ids = run.sample(text, num_frames=c.segment_length, guidance_scale=7.5, seed=31, logits_log=log, mode=mode).clone()
        torch.cuda.synchronize()
        res[mode] = (ids, torch.cat(log, 0), run.kc[0].clone(), run.vc[-1].clone())
        assert run._mode == mode
    assert int(run.fused_ctl[0]) > 0 and int(run.fused_ctl[1]) == 0          # steps ran through the persistent launch, no time-out
    assert run._chain_epoch > 0 and int(run.chain_ctl[0]) == 0               # ... and through the chained launches
    for mode in ("chained", "fused"):
        for u, v in zip(res["chain"], res[mode]):
            assert torch.equal(u, v), mode


def test_llm_full_size_chained_and_fused_blocks_equal_chain(cuda):
    """24 x 2048: the dependent-launch form (two streams, device-side waits, 120 launches per step) and the persistent one-launch
    form (256 workgroups, 143 grid barriers per step) of a decode step against the per-operation chain: one frame's decode
    (~330 steps), ids, final logits and the KV cache bit for bit."""
    from landiff_amd.config import LLMConfig
    from landiff_amd.llm import LLMRunner
    from landiff_amd.weights import init_state, llm_spec
    cfg = LLMConfig()
    run = LLMRunner(init_state(llm_spec(cfg), 5, dtype=torch.bfloat16, device=cuda), cfg, cuda)
    assert run.fused_supported and run.chained_supported
    text = torch.randn(48, cfg.text_dim, device=cuda, generator=torch.Generator(device=cuda).manual_seed(6))
    res = {}
    for mode in ("chain", "chained", "fused"):
        ids = run.sample(text, guidance_scale=7.5, seed=42, num_frames=1, mode=mode).clone()
        torch.cuda.synchronize()
        res[mode] = (ids, run.logits.clone(), run.kc[0].clone(), run.vc[-1].clone())
    assert int(run.fused_ctl[0]) > 300 and int(run.fused_ctl[1]) == 0
    assert run._chain_epoch > 300 and int(run.chain_ctl[0]) == 0
    for mode in ("chained", "fused"):
        for u, v in zip(res["chain"], res[mode]):
            assert torch.equal(u, v), mode


# ---- the 512 x 128 tile of the 8-phase GEMM loop for 128-column convolutions (ld_gemm8p_m512_kernel; LD_GEMM_M512 per call) ----
@pytest.mark.parametrize("T,H,W,Cin,Cout,resid", [(2, 256, 264, 128, 128, False), (3, 212, 210, 128, 128, True), (2, 256, 260, 128, 96, True),
                                                   (9, 120, 128, 128, 128, False)])
def test_conv_512x128_tile_route(cuda, monkeypatch, T, H, W, Cin, Cout, resid):
    """LD_GEMM_M512=1 (variants build): convolutions with one 128-wide column of output and 2048 <= K <= 4096 (the VAE's 480 x 720
    level) on ld_gemm8p_m512_kernel, the 8-phase loop on a 512 x 128 tile, 4 wave rows x 2 wave columns with the wave tile of the
    256 x 256 kernel.  Bit-identical to the 128 x 128 two-stage route of the shipped library (the knob is re-read per call under
    LD_TUNING=1) and within bf16 rounding of torch fp32 on sampled rows; row counts that are not multiples of 512, a column count
    below 128, both epilogues, GroupNorm partial sums from its epilogue, repeatable."""
    from landiff_amd import _lib, ops
    lib = _lib.load()
    assert lib.ld_conv_route(T, H, W, Cin, Cout, 3, 3, 3) == 0
    monkeypatch.setenv("LD_GEMM_M512", "1")
    assert lib.ld_conv_route(T, H, W, Cin, Cout, 3, 3, 3) == 4
    M = T * H * W
    g = torch.Generator(device="cpu").manual_seed(13)
    xp = torch.zeros(T + 2, H + 2, W + 2, Cin, device=cuda, dtype=torch.bfloat16)
    xp[:, 1:1 + H, 1:1 + W] = torch.randn(T + 2, H, W, Cin, generator=g).to(cuda, torch.bfloat16)
    wcl = (torch.randn(Cout, 3, 3, 3, Cin, generator=g) * 0.03).to(cuda, torch.bfloat16)
    bias = torch.randn(Cout, generator=g).to(cuda, torch.bfloat16)
    epi = dict(bias=bias)
    if resid:
        epi["resid"] = torch.randn(M, Cout, generator=g).to(cuda, torch.bfloat16)
    out = ops.conv_cl(xp, wcl, T, H, W, **epi)
    again = ops.conv_cl(xp, wcl, T, H, W, **epi)
    assert torch.equal(out, again)
    if Cout % 8 == 0:
        out_g, part = ops.conv_cl(xp, wcl, T, H, W, gn_partials=True, **epi)
        assert torch.equal(out_g, out)
    monkeypatch.setenv("LD_GEMM_M512", "0")
    assert lib.ld_conv_route(T, H, W, Cin, Cout, 3, 3, 3) == 0
    old = ops.conv_cl(xp, wcl, T, H, W, **epi)
    assert torch.equal(out, old)
    if Cout % 8 == 0:
        _, part_old = ops.conv_cl(xp, wcl, T, H, W, gn_partials=True, **epi)
        assert torch.equal(part, part_old)                 # the same epilogue on the same 64-row units: the same partial sums
    # torch fp32 on 4096 output rows (first / last rows of the problem and of tiles included): patches gathered from the padded input
    rows = torch.cat([torch.arange(0, 1024), torch.arange(M - 1024, M), torch.randint(0, M, (2048,), generator=g)]).to(cuda)
    t, h, w = rows // (H * W), (rows // W) % H, rows % W
    patch = torch.stack([xp[t + dt, h + dh, w + dw] for dt in range(3) for dh in range(3) for dw in range(3)], dim=1)      # [n][27][Cin]
    ref = patch.reshape(len(rows), -1).float() @ wcl.reshape(Cout, -1).float().T + bias.float()
    if resid:
        ref = ref.to(torch.bfloat16).float() + epi["resid"][rows].float()
    assert (out[rows].float() - ref).abs().max().item() / ref.abs().max().item() < 1e-2

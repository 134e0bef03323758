"""The AR decode at its real context length: 1 240 of a video's 1 244 decode steps run with 100-170 keys
in each split of the decode attention and a non-trivial merge of the partial results -- a regime the prefill + 3-step test of
tests/test_gpu_fullsize.py never enters.

  * ld_llm_kv_attn alone (fused RoPE + KV append + split-K attention + in-launch merge) against torch at B 2, H 16, D 128 over KV
    lengths 1 ... 1313, for every split count the launcher accepts;
  * the real LLMConfig cut to 2 blocks, prefill of 64 text tokens, 1 243 teacher-forced decode steps: CFG logits against the fp32
    oracle at KV lengths {128, 255, 256, 257, 700, 1024, 1300, last}, 2x-floor rule.

Reference: landiff/llm/modules/transformer_blocks.py:128-187 (local_kvcache_inference), landiff/modules/pos_emb.py:16-46."""
import dataclasses
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

B, H, D = 2, 16, 128
LENGTHS = [1, 2, 17, 128, 255, 256, 257, 511, 700, 1024, 1300, 1313]


def _rope_fp32(x, cos, sin):
    """apply_rope of the reference (pos_emb.py:16-46) on [..., H, D] with one position's [D/2] factors: fp32 products and
    differences, one rounding to bf16 at the end."""
    xf = x.float().reshape(*x.shape[:-1], D // 2, 2)
    a, b = xf[..., 0], xf[..., 1]
    return torch.stack([a * cos - b * sin, a * sin + b * cos], dim=-1).flatten(-2)


@pytest.mark.parametrize("nsplit", [8, 6])
def test_kv_attn_split_vs_torch_over_context_lengths(cuda, nsplit):
    """One decode-step attention launch per KV length L (the new token sits at position L - 1): the output against (i) torch fp32
    and (ii) a torch restatement of the kernel's dtype flow (scores and scaled scores rounded to bf16 as the reference's bf16
    einsum / division do, fp32 softmax); the appended cache row must be the rotated key / the value, bit for bit; rows beyond L
    are never read (they hold NaN here) and rows before it never written."""
    from landiff_amd import ops
    from oracle.llm import rope_table
    Lmax = 1344
    assert -(-Lmax // nsplit) <= 256
    g = torch.Generator().manual_seed(11)
    cos, sin = rope_table(D, Lmax, 10000.0)
    cos_d, sin_d = cos.to(cuda).contiguous(), sin.to(cuda).contiguous()
    kc0 = torch.randn(B, Lmax, H, D, generator=g).to(torch.bfloat16)
    vc0 = torch.randn(B, Lmax, H, D, generator=g).to(torch.bfloat16)
    ws = torch.zeros(B * H * (nsplit * 130 + 1), device=cuda, dtype=torch.float32)
    out = torch.empty(B, H * D, device=cuda, dtype=torch.bfloat16)
    pos = torch.zeros(1, device=cuda, dtype=torch.int32)
    worst = 0.0
    for L in LENGTHS:
        p = L - 1
        qkv = (torch.randn(B, 3, H, D, generator=g) * 1.5).to(torch.bfloat16)
        kc, vc = kc0.clone(), vc0.clone()
        kc[:, p:] = float("nan"); vc[:, p:] = float("nan")          # position p is written by the launch, everything behind it unused
        kc_d, vc_d = kc.to(cuda), vc.to(cuda)
        pos.fill_(p)
        out.fill_(float("nan"))
        ops.llm_kv_attn(None, kc_d, vc_d, pos, out, B, 1, H, Lmax, workspace=ws, nsplit=nsplit, qkv_fused=qkv.to(cuda).reshape(B, -1),
                        cos_t=cos_d, sin_t=sin_d)
        torch.cuda.synchronize()
        # ---- the append ----
        q_r = _rope_fp32(qkv[:, 0], cos[p], sin[p])
        k_r = _rope_fp32(qkv[:, 1], cos[p], sin[p]).to(torch.bfloat16)
        assert torch.equal(kc_d[:, p].cpu(), k_r), L
        assert torch.equal(vc_d[:, p].cpu(), qkv[:, 2]), L
        assert torch.equal(kc_d[:, :p].cpu(), kc0[:, :p]) and torch.equal(vc_d[:, :p].cpu(), vc0[:, :p]), L
        assert torch.isnan(kc_d[:, p + 1:].float()).all() and torch.isnan(vc_d[:, p + 1:].float()).all(), L
        assert int(ws[B * H * nsplit * 130:].abs().sum().item()) == 0, L          # the arrival counters are back at zero
        # ---- the attention ----
        K = torch.cat([kc0[:, :p], k_r[:, None]], 1).float()          # [B, L, H, D]
        V = torch.cat([vc0[:, :p], qkv[:, 2][:, None]], 1).float()
        qb = q_r.to(torch.bfloat16).float()
        s32 = torch.einsum("bhd,blhd->bhl", qb, K) / D ** 0.5
        ref32 = torch.einsum("bhl,blhd->bhd", torch.softmax(s32, -1), V).reshape(B, H * D)
        s16 = (torch.einsum("bhd,blhd->bhl", qb, K).to(torch.bfloat16).float() * (1.0 / D ** 0.5)).to(torch.bfloat16).float()
        ref16 = torch.einsum("bhl,blhd->bhd", torch.softmax(s16, -1), V).reshape(B, H * D)
        got = out.float().cpu()
        assert torch.isfinite(got).all(), L
        scale = ref32.abs().max().item()
        e32 = (got - ref32).abs().max().item() / scale
        e16 = (got - ref16).abs().max().item() / scale
        worst = max(worst, e16)
        assert e16 < 6e-3, (L, e16)           # same dtype flow: what is left is the bf16 output rounding (2^-9 of the value)
        assert e32 < 3e-2, (L, e32)           # against fp32: + the reference's own bf16 score rounding
    print(f"ld_llm_kv_attn nsplit={nsplit}: worst error vs the same-dtype-flow restatement {worst:.5f} of the output range")


def test_llm_two_blocks_full_width_decode_at_real_context_lengths_vs_oracle(cuda, oracle_bg):
    """The AR decoder at its real width (2048, 16 heads x 128, MLP 11008, vocabulary 2055, CFG pair) cut to 2 blocks so that the
    fp32 oracle runs 1 243 cached decode steps in about a minute: CFG logits of the HIP decode chain (teacher-forced) against
    the oracle at the KV lengths where split boundaries, the 256-key limit per split and the longest context lie.  The fp32 and
    the bf16 oracle decode run in child processes from the start of the session (tests/oracle_jobs.py: job_llm_two_blocks_*)."""
    from landiff_amd.llm import LLMRunner
    from oracle_jobs import llm_two_blocks_inputs
    cfg, sd, text, fed, S, full_len, steps, check_it = llm_two_blocks_inputs()
    run = LLMRunner({k: v.to(cuda) for k, v in sd.items()}, cfg, cuda)
    log = []
    run.sample(text.to(cuda), guidance_scale=7.5, motion_score=0.1, seed=42, logits_log=log, teacher_fed=fed.to(cuda))
    dev = torch.cat(log, 0).cpu()                                      # [1 + steps, vocab]: prefill, then one row per decode step
    assert dev.shape[0] == steps + 1 and steps >= 1240, (dev.shape, steps)
    assert check_it[0] >= 0 and check_it[-1] == steps - 1
    ref32, _ = oracle_bg.result("llm_two_blocks_fp32")
    ref16, _ = oracle_bg.result("llm_two_blocks_bf16")
    rel = lambda a, b: ((a.float() - b.float()).abs().max() / b.float().abs().max()).item()
    rows = []
    for it in check_it:
        err, floor = rel(dev[it + 1:it + 2], ref32[it]), rel(ref16[it], ref32[it])
        rows.append((S + 2 + it, err, floor))
        assert err < max(2 * floor, 2e-2), (S + 2 + it, err, floor)
    print("2-block full-width decode, CFG logits vs fp32 oracle (KV length: err / bf16-oracle floor): "
          + ", ".join(f"{L}: {e:.4f} / {f:.4f}" for L, e, f in rows))


def test_decode_is_unaffected_by_attention_launches_sharing_the_gpu(cuda):
    """Regression test of a round-5 finding (the full-size streaming determinism test tripped over it): the AR decode on a second
    stream UNDER DiT-sized attention launches (generate_many, the streaming loop) sampled other tokens than the same decode on a
    quiet GPU -- logits off by up to 2 of 30 from the prefill on.  tools/llm_race_probe2.py traced it to ld_llm_rope_append: the
    compiler's SLP vectoriser had packed RoPE's two dot products into v_pk_mul_f32 / v_pk_add_f32 with op_sel / neg modifiers, and
    those came out one fp32 rounding different while the 64-row attention kernel's MFMA waves shared the SIMD (never alone, never
    beside the GEMM or the plain attention kernel).  ld_llm.hip is built with -fno-slp-vectorize since (csrc/build.sh).  Here: the
    full-width decoder cut to 4 blocks, one frame's decode on a side stream from a helper thread, quiet and under back-to-back
    attention launches of the DiT shape -- the CFG logits of every step and the ids must be identical, and so must the attention
    output."""
    import threading
    from landiff_amd import ops
    from landiff_amd.config import LLMConfig
    from landiff_amd.llm import LLMRunner
    from landiff_amd.weights import init_state, llm_spec
    cfg = dataclasses.replace(LLMConfig(), num_layers=4)
    run = LLMRunner(init_state(llm_spec(cfg), 9, dtype=torch.bfloat16, device=cuda), cfg, cuda)
    g = torch.Generator(device=cuda).manual_seed(12)
    text = torch.randn(64, cfg.text_dim, device=cuda, generator=g)
    Bq, Hq, N = 2, 30, 17776
    Npad = (N + 127) // 128 * 128
    q = torch.zeros(Bq, Hq, Npad, 64, device=cuda, dtype=torch.bfloat16); k = torch.zeros_like(q)
    vt = torch.zeros(Bq, Hq, 64, Npad, device=cuda, dtype=torch.bfloat16)
    q[:, :, :N] = torch.randn(Bq, Hq, N, 64, device=cuda, generator=g).to(torch.bfloat16)
    k[:, :, :N] = torch.randn(Bq, Hq, N, 64, device=cuda, generator=g).to(torch.bfloat16)
    vt[:, :, :, :N] = torch.randn(Bq, Hq, 64, N, device=cuda, generator=g).to(torch.bfloat16)
    ao = torch.zeros(Bq, N, Hq * 64, device=cuda, dtype=torch.bfloat16)
    ops.attn_fwd(q, k, vt, ao, N, N, 0.125)
    torch.cuda.synchronize()
    ao_ref = ao.clone()
    side = torch.cuda.Stream(device=cuda, priority=-1)

    def decode(loaded):
        log, res, done = [], {}, threading.Event()

        def work():
            try:
                torch.cuda.set_device(cuda)
                with torch.cuda.stream(side):
                    res["ids"] = run.sample(text, guidance_scale=7.5, seed=42, num_frames=1, logits_log=log, mode="chain").clone()
                    side.synchronize()
            except BaseException as e:                       # noqa: BLE001 -- reported by the main thread
                res["error"] = e
            done.set()
        side.wait_stream(torch.cuda.current_stream(cuda))
        th = threading.Thread(target=work)
        th.start()
        n = 0
        while loaded and not done.is_set():
            ops.attn_fwd(q, k, vt, ao, N, N, 0.125)
            n += 1
            if n % 4 == 0:
                torch.cuda.current_stream(cuda).synchronize()
        th.join()
        torch.cuda.synchronize()
        if "error" in res:
            raise res["error"]
        return res["ids"], torch.cat(log, 0), n

    ids0, log0, _ = decode(False)
    for rep in range(2):
        ids, log, n = decode(True)
        assert n >= 8, n                                      # the decode really ran under attention launches
        assert torch.equal(log, log0), (rep, int((log != log0).any(dim=1).float().argmax()), (log - log0).abs().max().item())
        assert torch.equal(ids, ids0), rep
        assert torch.equal(ao, ao_ref), rep

"""GPU parity of the fused attention kernel against a plain torch fp32 reference."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pack(x, Npad):  # [B,H,N,64] -> zero padded [B,H,Npad,64]
    B, H, N, D = x.shape
    out = torch.zeros(B, H, Npad, D, device=x.device, dtype=x.dtype)
    out[:, :, :N] = x
    return out


def _run(cuda, B, H, N, fid=None, seed=0, spike=False, q_scale=1.0, relative=False):
    from landiff_amd import ops
    g = torch.Generator(device="cpu").manual_seed(seed)
    q = (torch.randn(B, H, N, 64, generator=g) * q_scale).to(cuda, torch.bfloat16)
    k = torch.randn(B, H, N, 64, generator=g).to(cuda, torch.bfloat16)
    v = torch.randn(B, H, N, 64, generator=g).to(cuda, torch.bfloat16)
    if spike:  # force a late running-max jump (exercises the online-softmax rescale)
        k[:, :, N - 3] = q[:, :, 5] * 4
    Npad = (N + 127) // 128 * 128
    qp, kp = _pack(q, Npad), _pack(k, Npad)
    vt = _pack(v, Npad).transpose(2, 3).contiguous()
    out = torch.zeros(B, N, H * 64, device=cuda, dtype=torch.bfloat16)
    kw = {}
    mask = None
    if fid is not None:
        fq = torch.full((Npad,), 0, dtype=torch.int32)
        fq[:N] = torch.from_numpy(fid)
        fk = torch.full((Npad,), np.iinfo(np.int32).max, dtype=torch.int32)
        fk[:N] = torch.from_numpy(fid)
        kt = fk.view(-1, 64)
        kw = dict(fid_q=fq.to(cuda), fid_k=fk.to(cuda), kt_min=kt.min(1).values.to(cuda).contiguous(),
                  kt_max=kt.max(1).values.to(cuda).contiguous())
        f = torch.from_numpy(fid).to(cuda)
        mask = f[None, :] <= f[:, None]  # [q, kv]
    ops.attn_fwd(qp, kp, vt, out, N, N, 0.125, **kw)
    s = (q.float() @ k.float().transpose(-1, -2)) * 0.125
    if mask is not None:
        s = s.masked_fill(~mask, float("-inf"))
    ref = (torch.softmax(s, -1) @ v.float()).permute(0, 2, 1, 3).reshape(B, N, H * 64)
    err = (out.float() - ref).abs().max().item()
    if relative:
        err /= ref.abs().max().item()
    return err


@pytest.mark.parametrize("B,H,N", [(1, 1, 128), (2, 3, 200), (1, 2, 1000), (1, 1, 64 * 5 + 17)])
def test_attn_full(cuda, B, H, N):
    assert _run(cuda, B, H, N) < 2e-2


def test_attn_rescale_branch(cuda):
    assert _run(cuda, 1, 2, 700, spike=True) < 2e-2


# key counts for which ld_attn_fwd_bf16 takes the software-pipelined kernel: n = ceil(N/64) >= 6 tiles (ld_attn_p16.hip, any n;
# the round-1 kernel of ld_attn_pipe.hip, LD_ATTN_VARIANT=8, needs (n - 2) % 4 == 0); 2175 has a ragged last tile, 1152 is an
# exact multiple of the query block
@pytest.mark.parametrize("B,H,N", [(1, 2, 1122), (2, 1, 1400), (1, 1, 2175), (1, 1, 1152),
                                   # every tile count modulo 4, ragged and exact last tiles (6 .. 13 tiles of 64 keys)
                                   (1, 1, 384), (1, 1, 385), (1, 2, 448), (1, 1, 500), (1, 1, 512), (2, 1, 575), (1, 1, 640), (1, 1, 700),
                                   (1, 1, 768), (1, 1, 830)])
def test_attn_pipelined(cuda, B, H, N):
    assert _run(cuda, B, H, N, seed=N) < 2e-2


def test_attn_pipelined_spike(cuda):
    # a late key that dominates its row (scores still inside the fast pass's exp2 window); the spiked rows are ~one-hot
    # with |out| ~ 4, so the bound is relative to the output range (bf16 output rounding alone is 4e-3 of it)
    assert _run(cuda, 1, 2, 1400, spike=True, relative=True) < 1e-2


def test_attn_pipelined_overflow_falls_back(cuda):
    # |q.k| * scale * log2(e) far beyond 128: the max-free fast pass overflows its denominators, the workgroup must
    # notice and redo the rows with the running-max pass (without the redo the rows come out as 0 or NaN: relative
    # error 1).  With scores of several hundred the bf16 rounding of q alone moves exp2 arguments by ~0.5, so the bound is
    # loose; the plain running-max kernel has the same 0.12 on this input.
    err = _run(cuda, 1, 2, 1122, spike=True, q_scale=6.0, relative=True)
    assert err == err and err < 0.2


def test_attn_frame_mask(cuda):
    # TiTok decoder layout: T frames x tpf visual tokens, then I tokens (fid 0), then P tokens per frame
    T, tpf, nI, nP = 5, 90, 22, 7
    fid = np.concatenate([np.repeat(np.arange(T), tpf), np.zeros(nI, np.int64), np.repeat(np.arange(1, T), nP)]).astype(np.int32)
    assert _run(cuda, 1, 2, len(fid), fid=fid, seed=3) < 2e-2


def _last_kernel():
    from landiff_amd import _lib
    return (_lib.load().ld_attn_last_kernel() or b"").decode()


# ---- the dynamic form of the 64-row kernel (ld_attn_q64_dyn_kernel: one workgroup per slot pulls query blocks, XCD by XCD) ----
def _qkv(cuda, B, H, N, seed):
    g = torch.Generator(device=cuda).manual_seed(seed)
    Npad = (N + 127) // 128 * 128
    q = torch.zeros(B, H, Npad, 64, device=cuda, dtype=torch.bfloat16); k = torch.zeros_like(q)
    vt = torch.zeros(B, H, 64, Npad, device=cuda, dtype=torch.bfloat16)
    q[:, :, :N] = torch.randn(B, H, N, 64, device=cuda, generator=g).to(torch.bfloat16)
    k[:, :, :N] = torch.randn(B, H, N, 64, device=cuda, generator=g).to(torch.bfloat16)
    vt[:, :, :, :N] = torch.randn(B, H, 64, N, device=cuda, generator=g).to(torch.bfloat16)
    return q, k, vt


@pytest.mark.parametrize("B,H,N", [(2, 30, 17776), (1, 40, 13100)])
def test_attn_dynamic_queue_equals_round_robin_dispatch(cuda, monkeypatch, B, H, N):
    """Grids of >= four rounds take the dynamic form by default; LD_ATTN_DYN=0 (re-read per call under LD_TUNING=1, conftest) is the
    hardware's own dispatch of one workgroup per query block.  Same per-block code: identical bits.  The queue counters are static
    device memory in 64 sets, one per (device, stream), zeroed on the stream before each launch: 70 launches in a row, and launches
    on two streams at once, stay identical."""
    from landiff_amd import ops
    q, k, vt = _qkv(cuda, B, H, N, seed=N)
    monkeypatch.setenv("LD_ATTN_DYN", "0")
    ref = torch.zeros(B, N, H * 64, device=cuda, dtype=torch.bfloat16)
    ops.attn_fwd(q, k, vt, ref, N, N, 0.125)
    assert _last_kernel() == "ld_attn_q64_kernel"
    monkeypatch.delenv("LD_ATTN_DYN")
    out = torch.zeros_like(ref)
    for i in range(70):
        out.zero_()
        ops.attn_fwd(q, k, vt, out, N, N, 0.125)
        if i in (0, 1, 63, 64, 69):
            assert torch.equal(out, ref), i
    assert _last_kernel() == "ld_attn_q64_dyn_kernel"
    q2, k2, vt2 = _qkv(cuda, B, H, N, seed=N + 1)
    monkeypatch.setenv("LD_ATTN_DYN", "0")
    ref2 = torch.zeros_like(ref)
    ops.attn_fwd(q2, k2, vt2, ref2, N, N, 0.125)
    monkeypatch.delenv("LD_ATTN_DYN")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    o1 = [torch.zeros_like(ref) for _ in range(4)]; o2 = [torch.zeros_like(ref) for _ in range(4)]
    torch.cuda.synchronize()
    for i in range(4):                       # two launches in flight together, each pulling from its own counter set
        with torch.cuda.stream(s1):
            ops.attn_fwd(q, k, vt, o1[i], N, N, 0.125)
        with torch.cuda.stream(s2):
            ops.attn_fwd(q2, k2, vt2, o2[i], N, N, 0.125)
    torch.cuda.synchronize()
    assert all(torch.equal(o, ref) for o in o1) and all(torch.equal(o, ref2) for o in o2)


def test_attn_dynamic_queue_survives_poisoned_counters_and_stream_exhaustion(cuda):
    """The one piece of library state (include/landiff_hip.h): (i) counters left dirty by a launch that died mid-flight -- simulated
    with ld_attn_queue_poke -- do not reach the next launch (the set is zeroed on the launch stream first); (ii) ld_reset puts the
    sets and the stream assignments back; (iii) the 65th stream of a device that launches attention gets no set and takes the
    static launch of the same body; (iv) so does a launch captured into a graph.  Every output bit-identical."""
    from landiff_amd import ops
    B, H, N = 2, 30, 17776
    q, k, vt = _qkv(cuda, B, H, N, seed=5)
    torch.cuda.synchronize()
    ops.reset()
    ref = torch.zeros(B, N, H * 64, device=cuda, dtype=torch.bfloat16)
    ops.attn_fwd(q, k, vt, ref, N, N, 0.125)
    assert _last_kernel() == "ld_attn_q64_dyn_kernel"
    out = torch.zeros_like(ref)
    for value in (0xFFFFFFFF, 7, 1 << 20):                       # exhausted / partially consumed / far past the end
        ops.attn_queue_poke(value)
        out.zero_()
        ops.attn_fwd(q, k, vt, out, N, N, 0.125)
        assert _last_kernel() == "ld_attn_q64_dyn_kernel"
        assert torch.equal(out, ref), hex(value)
    ops.attn_queue_poke(0xFFFFFFFF)
    torch.cuda.synchronize()
    ops.reset()
    out.zero_()
    ops.attn_fwd(q, k, vt, out, N, N, 0.125)
    assert torch.equal(out, ref)
    # 70 distinct streams (raw hipStreamCreate: torch hands out streams from a pool of 32): the first 63 new ones run the dynamic
    # form (the current stream holds a set since the reset), the rest the static one
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    raw = []
    for _ in range(70):
        h = ctypes.c_void_p()
        assert hip.hipStreamCreate(ctypes.byref(h)) == 0
        raw.append(h)
    assert len({h.value for h in raw}) == 70
    streams = [torch.cuda.ExternalStream(h.value) for h in raw]
    torch.cuda.synchronize()
    outs = [torch.zeros_like(ref) for _ in range(4)]
    names = []
    for i, s in enumerate(streams):
        with torch.cuda.stream(s):
            ops.attn_fwd(q, k, vt, outs[i % 4], N, N, 0.125)
            names.append(_last_kernel())
        if i % 4 == 3:
            torch.cuda.synchronize()
            assert all(torch.equal(o, ref) for o in outs), i
            for o in outs:
                o.zero_()
            torch.cuda.synchronize()                             # (the new streams do not order themselves behind the zero fill)
    torch.cuda.synchronize()
    assert names.count("ld_attn_q64_dyn_kernel") == 63 and names[63:] == ["ld_attn_q64_kernel"] * 7, names
    del streams
    for h in raw:
        assert hip.hipStreamDestroy(h) == 0
    ops.reset()
    # captured launch: static form, replay equals eager
    g = torch.cuda.CUDAGraph()
    out.zero_()
    s = torch.cuda.Stream()
    cnt = torch.full((1,), 7, device=cuda, dtype=torch.int32)
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            ops.attn_fwd(q, k, vt, out, N, N, 0.125)
            assert _last_kernel() == "ld_attn_q64_kernel"
            ops.attn_last_fallbacks(cnt)                         # (a captured node too: the static form keeps no count)
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref) and int(cnt.item()) == -1


# ---- round 6: the fallback at the headline shape, priced by tools/attn_logit_sweep.py (profiles/r06_attn_logit_sweep.txt) ----
def _sink_problem(cuda, T, frac, sink_key, seed=21):
    """The DiT shape with ONE sink key per head whose logit q.k/8 is T for the queries of a fraction of the 256-row blocks
    (tools/attn_logit_sweep.py; T < 0: every key's logit is ~T for those queries -- the underflow side of the window)."""
    import math
    B, H, N = 2, 30, 17776
    q, k, vt = _qkv(cuda, B, H, N, seed=seed)
    Npad = q.shape[2]
    nblk = (Npad + 255) // 256
    g = torch.Generator(device=cuda).manual_seed(seed + 1)
    sel = torch.rand(B, H, nblk, device=cuda, generator=g) < frac
    rows = sel[:, :, :, None].expand(B, H, nblk, 256).reshape(B, H, nblk * 256)[:, :, :Npad]
    beta = math.sqrt(8.0 * abs(T))
    if T > 0:
        k[:, :, sink_key, :] = 0
        k[:, :, sink_key, 0] = beta
    else:
        k[:, :, :N, 0] = beta
    q[:, :, :, 0] = torch.where(rows, torch.full_like(q[:, :, :, 0], beta if T > 0 else -beta), q[:, :, :, 0])
    q[:, :, N:] = 0
    return q, k, vt, sel, B, H, N


def _err_vs_fp32(q, k, vt, out, N, heads):
    worst = 0.0
    for b, h in heads:
        s = (q[b, h, :N].float() @ k[b, h, :N].float().T) * 0.125
        ref = torch.softmax(s, dim=1) @ vt[b, h, :, :N].float().T
        got = out[b, :, h * 64:(h + 1) * 64].float()
        assert torch.isfinite(got).all(), (b, h)
        worst = max(worst, float((got - ref).abs().max() / ref.abs().max()))
    return worst


@pytest.mark.parametrize("T,frac,sink_key", [(90, 0.1, 0), (120, 1.0, 0), (90, 0.3, 17775), (-70, 0.2, 0), (70, 1.0, 0)])
def test_attn_fallback_at_the_dit_shape_with_sink_keys(cuda, monkeypatch, T, frac, sink_key):
    """The pulled launch at the headline shape on data that drives the max-free fast pass out of its window ([2^-80, 2^110] for a
    row's denominator: a row maximum of q.k/8 beyond ~76, or every logit below ~-55): the affected 256-row blocks re-run through
    the running-max pass inside the launch.  Sink at key 0 / at the last key / the underflow side / a sink INSIDE the window (70:
    no block re-runs).  Against torch fp32 on heads with and without affected blocks, run to run identical, and the same bits as
    the one-workgroup-per-block dispatch (LD_ATTN_DYN=0: same per-block code)."""
    from landiff_amd import ops
    q, k, vt, sel, B, H, N = _sink_problem(cuda, T, frac, sink_key)
    out = torch.full((B, N, H * 64), float("nan"), device=cuda, dtype=torch.bfloat16)
    ops.attn_fwd(q, k, vt, out, N, N, 0.125)
    assert _last_kernel() == "ld_attn_q64_dyn_kernel"
    # ld_attn_last_fallbacks: exactly the blocks that hold sink queries re-ran (none for a sink inside the window)
    cnt = torch.full((1,), -7, device=cuda, dtype=torch.int32)
    ops.attn_last_fallbacks(cnt)
    assert int(cnt.item()) == (int(sel.sum().item()) if (T > 76 or T < -55) else 0), (int(cnt.item()), int(sel.sum().item()))
    # a logit of ~100 in bf16 operands carries an absolute error of ~0.4: the sink rows' probabilities move by tens of per cent of
    # tiny tails; the output (dominated by the sink's V row) stays within a few bf16 ulps of the range
    err = _err_vs_fp32(q, k, vt, out, N, [(0, 0), (1, 17), (1, 29)])
    assert err < 3e-2, err
    out2 = torch.full_like(out, float("nan"))
    ops.attn_fwd(q, k, vt, out2, N, N, 0.125)
    assert torch.equal(out, out2)
    monkeypatch.setenv("LD_ATTN_DYN", "0")
    ref = torch.full_like(out, float("nan"))
    ops.attn_fwd(q, k, vt, ref, N, N, 0.125)
    assert _last_kernel() == "ld_attn_q64_kernel"
    assert torch.equal(out, ref)
    ops.attn_last_fallbacks(cnt)
    assert int(cnt.item()) == -1          # the static dispatch has the window but keeps no count
    ops.attn_fwd(q, k, vt, ref, N, N, 0.125, exact=True)
    ops.attn_last_fallbacks(cnt)
    assert int(cnt.item()) == 0           # the exact form has no window to leave


@pytest.mark.parametrize("T,frac,sink_key", [(None, 0.0, 0), (90, 0.1, 0), (120, 1.0, 0), (90, 0.3, 17775), (-70, 0.2, 0)])
def test_attn_exact_form_any_logit_range(cuda, T, frac, sink_key):
    """ld_attn_fwd_bf16_exact at the DiT shape: row maxima first, then the pipelined loop with -max as the score accumulators' initial
    value -- benign data, sinks far outside the fast pass's window (at the first and at the last key), the underflow side.  Against
    torch fp32 on three heads, run to run identical; on benign data within bf16 rounding of the default launch (the same bf16 scores,
    2^(s - max) instead of 2^s); on the out-of-window problems within bf16 rounding of the default launch's in-kernel fallback."""
    from landiff_amd import ops
    if T is None:
        B, H, N = 2, 30, 17776
        q, k, vt = _qkv(cuda, B, H, N, seed=77)
    else:
        q, k, vt, _, B, H, N = _sink_problem(cuda, T, frac, sink_key)
    out = torch.full((B, N, H * 64), float("nan"), device=cuda, dtype=torch.bfloat16)
    ops.attn_fwd(q, k, vt, out, N, N, 0.125, exact=True)
    assert _last_kernel() == "ld_attn_q64_exact_kernel"
    assert _err_vs_fp32(q, k, vt, out, N, [(0, 0), (1, 17), (1, 29)]) < 3e-2
    out2 = torch.full_like(out, float("nan"))
    ops.attn_fwd(q, k, vt, out2, N, N, 0.125, exact=True)
    assert torch.equal(out, out2)
    ref = torch.full_like(out, float("nan"))
    ops.attn_fwd(q, k, vt, ref, N, N, 0.125)
    d = (out.float() - ref.float()).abs().max().item() / ref.float().abs().max().item()
    assert d < 2e-2, d


def test_attn_exact_form_small_and_masked_shapes_take_the_plain_kernel(cuda):
    """Shapes outside the pipelined tile (fewer than 6 key tiles; frame masks) run the plain online-softmax kernel in both entry
    points: the same bits."""
    from landiff_amd import ops
    B, H, N = 1, 2, 300
    q, k, vt = _qkv(cuda, B, H, N, seed=3)
    a = torch.zeros(B, N, H * 64, device=cuda, dtype=torch.bfloat16); b = torch.zeros_like(a)
    ops.attn_fwd(q, k, vt, a, N, N, 0.125)
    ops.attn_fwd(q, k, vt, b, N, N, 0.125, exact=True)
    assert torch.equal(a, b) and _last_kernel().startswith("ld_attn_kernel")
    cnt = torch.full((1,), -7, device=cuda, dtype=torch.int32)
    ops.attn_last_fallbacks(cnt)
    assert int(cnt.item()) == 0           # running-max kernel: nothing to fall back from

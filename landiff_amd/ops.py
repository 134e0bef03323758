"""Thin tensor-level wrappers over the C ABI (include/landiff_hip.h).

torch is used only for device memory and the current HIP stream; all arithmetic happens in
liblandiff_hip.so.  Every wrapper raises if its tensors are not on a GPU.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from ._lib import Epilogue, check

ACT = {None: 0, "none": 0, "gelu_tanh": 1, "gelu_erf": 2, "silu": 3, "tanh": 4}


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: torch.Tensor | None):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.LandiffHipError("landiff_amd ops need GPU tensors (no CPU fallback)")
    return ctypes.c_void_p(t.data_ptr())


def _bf16(t: torch.Tensor, name: str):
    if t.dtype != torch.bfloat16:
        raise TypeError(f"{name} must be bfloat16, got {t.dtype}")


def make_epilogue(
    *,
    bias=None,
    act=None,
    mul=None,
    resid=None,
    gate=None,
    gate_bstride=0,
    gate_off_img=0,
    gate_off_txt=0,
    rows_per_batch=0,
    text_len=0,
    add2=None,
    out_f32=False,
) -> Epilogue:
    e = Epilogue()
    e.bias = _ptr(bias)
    e.act = ACT[act]
    e.mul = _ptr(mul)
    e.ldmul = mul.stride(-2) if mul is not None else 0
    e.resid = _ptr(resid)
    e.ldr = resid.stride(-2) if resid is not None else 0
    e.resid_f32 = int(resid is not None and resid.dtype == torch.float32)
    e.gate = _ptr(gate)
    e.gate_bstride = gate_bstride
    e.gate_off_img = gate_off_img
    e.gate_off_txt = gate_off_txt
    e.rows_per_batch = rows_per_batch
    e.text_len = text_len
    e.add2 = _ptr(add2)
    e.ldadd = add2.stride(-2) if add2 is not None else 0
    e.out_f32 = int(out_f32)
    return e


def gemm(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor | None = None, **epi) -> torch.Tensor:
    """out[M,N] = epilogue(a[M,K] @ w[N,K]^T).  a may have a row stride (last dim contiguous)."""
    _bf16(a, "a"); _bf16(w, "w")
    assert a.dim() == 2 and w.dim() == 2 and a.stride(1) == 1 and w.is_contiguous()
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K
    out_f32 = bool(epi.get("out_f32", False))
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32 if out_f32 else torch.bfloat16)
    assert out.stride(1) == 1 and out.shape == (M, N)
    assert (out.dtype == torch.float32) == out_f32
    e = make_epilogue(**epi)
    lib = _lib.load()
    check(lib.ld_gemm_bf16(_ptr(a), a.stride(0), _ptr(w), _ptr(out), out.stride(0), M, N, K,
                           ctypes.byref(e), _stream()), "ld_gemm_bf16")
    return out


def quantize_fp8(x: torch.Tensor, q: torch.Tensor | None = None, scale: torch.Tensor | None = None):
    """Row-wise dynamic e4m3 quantisation: x bf16 [M,K] -> (q uint8 [M,K], scale fp32 [M]); x ~= q * scale[:, None]."""
    _bf16(x, "x")
    M, K = x.shape
    assert x.stride(1) == 1
    if q is None:
        q = torch.empty(M, K, device=x.device, dtype=torch.uint8)
    if scale is None:
        scale = torch.empty(M, device=x.device, dtype=torch.float32)
    assert q.dtype == torch.uint8 and q.shape == (M, K) and q.stride(1) == 1 and scale.dtype == torch.float32 and scale.is_contiguous()
    check(_lib.load().ld_quantize_fp8(_ptr(x), x.stride(0), _ptr(q), q.stride(0), _ptr(scale), M, K, _stream()), "ld_quantize_fp8")
    return q, scale


def gemm_fp8(a8: torch.Tensor, scale_a: torch.Tensor, w8: torch.Tensor, scale_w: torch.Tensor,
             out: torch.Tensor | None = None, **epi) -> torch.Tensor:
    """out[M,N] = epilogue((a8 * scale_a[:,None]) @ (w8 * scale_w[:,None])^T), e4m3 operands as uint8 tensors."""
    assert a8.dtype == torch.uint8 and w8.dtype == torch.uint8 and a8.stride(1) == 1 and w8.is_contiguous()
    M, K = a8.shape
    N = w8.shape[0]
    assert w8.shape[1] == K and scale_a.shape == (M,) and scale_w.shape == (N,)
    assert scale_a.dtype == torch.float32 and scale_w.dtype == torch.float32 and scale_a.is_contiguous() and scale_w.is_contiguous()
    out_f32 = bool(epi.get("out_f32", False))
    if out is None:
        out = torch.empty((M, N), device=a8.device, dtype=torch.float32 if out_f32 else torch.bfloat16)
    assert out.stride(1) == 1 and out.shape == (M, N) and (out.dtype == torch.float32) == out_f32
    e = make_epilogue(**epi)
    check(_lib.load().ld_gemm_fp8(_ptr(a8), a8.stride(0), _ptr(scale_a), _ptr(w8), _ptr(scale_w), _ptr(out), out.stride(0),
                                  M, N, K, ctypes.byref(e), _stream()), "ld_gemm_fp8")
    return out


def quantize_mxfp8(x: torch.Tensor, q: torch.Tensor | None = None, scales: torch.Tensor | None = None):
    """MXFP8 quantisation: x bf16 [M,K] -> (q uint8 [M,K] e4m3 codes, scales uint8 [K/128, M, 4] E8M0 bytes, K-tile-major)."""
    _bf16(x, "x")
    M, K = x.shape
    assert x.stride(1) == 1 and K % 128 == 0
    if q is None:
        q = torch.empty(M, K, device=x.device, dtype=torch.uint8)
    if scales is None:
        scales = torch.empty(K // 128, M, 4, device=x.device, dtype=torch.uint8)
    assert q.dtype == torch.uint8 and q.shape == (M, K) and q.stride(1) == 1
    assert scales.dtype == torch.uint8 and tuple(scales.shape) == (K // 128, M, 4) and scales.is_contiguous()
    check(_lib.load().ld_quantize_mxfp8(_ptr(x), x.stride(0), _ptr(q), q.stride(0), _ptr(scales), M, M, K,
                                        _stream()), "ld_quantize_mxfp8")
    return q, scales


def gemm_mxfp8(a8: torch.Tensor, sa: torch.Tensor, w8: torch.Tensor, sw: torch.Tensor, out: torch.Tensor | None = None,
               out_scales: torch.Tensor | None = None, **epi) -> torch.Tensor:
    """out[M,N] = epilogue(dequant(a8, sa) @ dequant(w8, sw)^T) with the MX block scales applied inside the MFMA.
    out_scales (uint8 [N/128, M, 4]): the output is MXFP8 too (out uint8 [M,N]); bias + act="gelu_tanh" only."""
    assert a8.dtype == torch.uint8 and w8.dtype == torch.uint8 and a8.stride(1) == 1 and w8.is_contiguous()
    M, K = a8.shape
    N = w8.shape[0]
    assert w8.shape[1] == K and tuple(sa.shape) == (K // 128, M, 4) and tuple(sw.shape) == (K // 128, N, 4)
    assert sa.dtype == torch.uint8 and sw.dtype == torch.uint8 and sa.is_contiguous() and sw.is_contiguous()
    out_f32 = bool(epi.get("out_f32", False))
    if out_scales is not None:
        assert out is not None and out.dtype == torch.uint8 and out_scales.dtype == torch.uint8
        assert tuple(out_scales.shape) == (N // 128, M, 4) and out_scales.is_contiguous()
    elif out is None:
        out = torch.empty((M, N), device=a8.device, dtype=torch.float32 if out_f32 else torch.bfloat16)
    assert out.stride(1) == 1 and out.shape == (M, N)
    assert out_scales is not None or (out.dtype == torch.float32) == out_f32
    e = make_epilogue(**epi)
    check(_lib.load().ld_gemm_mxfp8(_ptr(a8), a8.stride(0), _ptr(sa), _ptr(w8), _ptr(sw), _ptr(out), out.stride(0),
                                    _ptr(out_scales), M if out_scales is not None else 0,
                                    M, N, K, ctypes.byref(e), _stream()), "ld_gemm_mxfp8")
    return out


def conv_cl(x_padded: torch.Tensor, w: torch.Tensor, T: int, H: int, W: int,
            out: torch.Tensor | None = None, gn_partials: bool = False, **epi):
    """Channels-last conv.  x_padded [T+kT-1, H+kH-1, W+kW-1, Cin]; w [Cout, kT, kH, kW, Cin].
    gn_partials=True: returns (out, partials) -- the epilogue also leaves the GroupNorm partial sums of the bf16 output
    (fp32 [ceil(M / 64)][Cout / 4][2]) for groupnorm_stats_from_conv, instead of a second read of the activation."""
    _bf16(x_padded, "x_padded"); _bf16(w, "w")
    assert x_padded.is_contiguous() and w.is_contiguous() and w.dim() == 5
    Cout, kT, kH, kW, Cin = w.shape
    assert tuple(x_padded.shape) == (T + kT - 1, H + kH - 1, W + kW - 1, Cin), (x_padded.shape, w.shape, T, H, W)
    out_f32 = bool(epi.get("out_f32", False))
    if out is None:
        out = torch.empty((T * H * W, Cout), device=x_padded.device,
                          dtype=torch.float32 if out_f32 else torch.bfloat16)
    assert out.stride(-1) == 1
    e = make_epilogue(**epi)
    lib = _lib.load()
    if gn_partials:
        part = torch.empty(int(lib.ld_conv_gn_partials_size(T * H * W, Cout)), device=x_padded.device, dtype=torch.float32)
        check(lib.ld_conv_cl_bf16_gn(_ptr(x_padded), _ptr(w), _ptr(out), out.stride(-2), T, H, W, Cin, Cout,
                                     kT, kH, kW, ctypes.byref(e), _ptr(part), _stream()), "ld_conv_cl_bf16_gn")
        return out, part
    check(lib.ld_conv_cl_bf16(_ptr(x_padded), _ptr(w), _ptr(out), out.stride(-2), T, H, W, Cin, Cout,
                              kT, kH, kW, ctypes.byref(e), _stream()), "ld_conv_cl_bf16")
    return out


def attn_fwd(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, out: torch.Tensor, Nq: int, Nk: int,
             scale: float, fid_q=None, fid_k=None, kt_min=None, kt_max=None, q_row0: int = 0, exact: bool = False) -> torch.Tensor:
    """q,k [B,H,Npad,64]; vt [B,H,64,Npad]; out [B,N,H*64] (bf16).  Optional frame mask arrays (int32).
    q_row0: the Nq query rows start at row q_row0 of q / fid_q / out (pointer offsets only; Npad must cover
    q_row0 + Nq rounded up to 128).  exact: ld_attn_fwd_bf16_exact -- the two-pass safe softmax at a fixed ~1.55 x, for logit
    ranges beyond the window of the default launch's fast pass."""
    _bf16(q, "q"); _bf16(k, "k"); _bf16(vt, "vt"); _bf16(out, "out")
    B, H, Npad, D = q.shape
    assert D == 64 and k.shape == q.shape and tuple(vt.shape) == (B, H, 64, Npad)
    assert q.is_contiguous() and k.is_contiguous() and vt.is_contiguous() and out.stride(-1) == 1
    assert out.shape[0] == B and out.shape[2] == H * 64
    lib = _lib.load()
    qp, op, fqp = _ptr(q), _ptr(out), _ptr(fid_q)
    if q_row0:
        assert B == 1 and q_row0 % 2 == 0 and q_row0 + (Nq + 127) // 128 * 128 <= Npad
        qp = ctypes.c_void_p(qp.value + q_row0 * 64 * 2)
        op = ctypes.c_void_p(op.value + q_row0 * out.stride(1) * 2)
        fqp = ctypes.c_void_p(fqp.value + q_row0 * 4) if fqp is not None else None
    fn = lib.ld_attn_fwd_bf16_exact if exact else lib.ld_attn_fwd_bf16
    check(fn(qp, _ptr(k), _ptr(vt), op, B, H, Nq, Nk, Npad, out.stride(0), out.stride(1), float(scale),
             fqp, _ptr(fid_k), _ptr(kt_min), _ptr(kt_max), _stream()),
          "ld_attn_fwd_bf16_exact" if exact else "ld_attn_fwd_bf16")
    return out


def attn_last_fallbacks(out: torch.Tensor) -> torch.Tensor:
    """ld_attn_last_fallbacks: out int32 [1] on the device <- the number of 256-row query blocks of this thread's last attn_fwd launch
    (on the current stream) that left the fast pass's window and were recomputed; 0: kernel without a window, -1: no count kept."""
    assert out.dtype == torch.int32 and out.is_cuda and out.numel() >= 1
    check(_lib.load().ld_attn_last_fallbacks(_ptr(out), _stream()), "ld_attn_last_fallbacks")
    return out


def reset() -> None:
    """ld_reset on the current stream: forget the stream -> counter-set assignments of the dynamic attention launch on the current
    device and zero the sets.  Only when no attention launch is in flight on another stream (after a device synchronise)."""
    check(_lib.load().ld_reset(_stream()), "ld_reset")


def attn_queue_poke(value: int, set_index: int = -1) -> None:
    """Test hook (ld_attn_queue_poke): leave `value` in the attention work-queue counters, as an aborted launch would."""
    check(_lib.load().ld_attn_queue_poke(set_index, value, _stream()), "ld_attn_queue_poke")


# ------------------------------------------------------------------------------------------------
# small-batch / LLM kernels
# ------------------------------------------------------------------------------------------------
def gemv(x, w, out, *, w2=None, bias=None, resid=None, in_act=None, act=None, norm_w=None, norm_eps=0.0):
    """out[B,N] = epi(x[B,K] @ w[N,K]^T) for B <= 4 (weight streaming)."""
    B, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K and out.shape == (B, N) and x.stride(1) == 1 and out.stride(1) == 1 and w.is_contiguous()
    w_f32 = w.dtype == torch.float32
    lib = _lib.load()
    check(lib.ld_gemv(_ptr(x), x.stride(0), int(x.dtype == torch.float32), _ptr(w), _ptr(w2), int(w_f32), _ptr(bias),
                      _ptr(resid), resid.stride(0) if resid is not None else 0, _ptr(out), out.stride(0),
                      int(out.dtype == torch.float32), B, N, K, ACT[in_act], ACT[act], _ptr(norm_w), float(norm_eps),
                      _stream()), "ld_gemv")
    return out


def rmsnorm(x, w, out, eps):
    rows, D = x.shape
    _bf16(x, "x")
    assert w.dtype == torch.float32 and x.is_contiguous() and out.is_contiguous()
    check(_lib.load().ld_rmsnorm_bf16(_ptr(x), _ptr(w), _ptr(out), rows, D, float(eps), _stream()), "ld_rmsnorm_bf16")
    return out


def layernorm_bf16_to_f32(x, w, b, out, eps):
    rows, D = x.shape
    check(_lib.load().ld_layernorm_bf16_to_f32(_ptr(x), x.stride(0), _ptr(w), _ptr(b), _ptr(out), rows, D, float(eps),
                                               _stream()), "ld_layernorm_bf16_to_f32")
    return out


def llm_rope_append(qkv, cos_t, sin_t, pos, q_out, k_cache, v_cache, B, m, H, Lmax):
    check(_lib.load().ld_llm_rope_append(_ptr(qkv), _ptr(cos_t), _ptr(sin_t), _ptr(pos), _ptr(q_out), _ptr(k_cache),
                                         _ptr(v_cache), B, m, H, Lmax, _stream()), "ld_llm_rope_append")


def llm_kv_attn(q, k_cache, v_cache, pos, out, B, m, H, Lmax, workspace=None, nsplit=1, qkv_fused=None, cos_t=None,
                sin_t=None):
    if workspace is not None and nsplit > 1 and m == 1:
        # split decode attention: B*H*nsplit partial results of 130 words + one arrival counter per (batch row, head) at the tail,
        # which must start at zero (the last arriver re-zeroes it) -- include/landiff_hip.h, ld_llm_kv_attn
        need = B * H * (nsplit * 130 + 1)
        assert workspace.dtype == torch.float32 and workspace.numel() >= need, \
            f"ld_llm_kv_attn workspace: {workspace.numel()} fp32 words, needs B*H*(nsplit*130+1) = {need}"
    check(_lib.load().ld_llm_kv_attn(_ptr(q), _ptr(k_cache), _ptr(v_cache), _ptr(pos), _ptr(out), B, m, H, Lmax,
                                     _ptr(workspace), nsplit, _ptr(qkv_fused), _ptr(cos_t), _ptr(sin_t), _stream()),
          "ld_llm_kv_attn")


def llm_embed(table, token, out):
    B, D = out.shape
    check(_lib.load().ld_llm_embed(_ptr(table), _ptr(token), _ptr(out), B, D, _stream()), "ld_llm_embed")


def llm_layer_table(blocks, k_caches, v_caches):
    """Host-side array of ld_llm_layer for llm_decode_forward (keeps no references: the caller owns the tensors)."""
    arr = (_lib.LlmLayer * len(blocks))()
    for i, w in enumerate(blocks):
        for name in ("wqkv", "wo", "w1", "w3", "w2"):
            assert w[name].dtype == torch.bfloat16 and w[name].is_contiguous()
            setattr(arr[i], name, w[name].data_ptr())
        for name in ("n0", "n1"):
            assert w[name].dtype == torch.float32 and w[name].is_contiguous()
            setattr(arr[i], name, w[name].data_ptr())
        arr[i].k_cache, arr[i].v_cache = k_caches[i].data_ptr(), v_caches[i].data_ptr()
    return arr


def llm_decode_forward(table, emb, token, pos, x, qkv, att, gate, attn_ws, cos_t, sin_t, lnf_w, lnf_b, lnf_out, head, logits,
                       heads, Lmax, nsplit, rms_eps, ln_eps, pos_value=-1):
    """One decode step (embedding -> all blocks -> final LN -> fp32 head), every launch queued from native code."""
    B, hidden = x.shape
    for t in (x, qkv, att, gate, lnf_out, logits, head):
        assert t.is_contiguous()
    check(_lib.load().ld_llm_decode_forward(ctypes.addressof(table), len(table), _ptr(emb), _ptr(token), _ptr(pos), int(pos_value), _ptr(x),
                                            _ptr(qkv), _ptr(att), _ptr(gate), _ptr(attn_ws), _ptr(cos_t), _ptr(sin_t),
                                            _ptr(lnf_w), _ptr(lnf_b), _ptr(lnf_out), _ptr(head), _ptr(logits), B, hidden,
                                            heads, gate.shape[1], logits.shape[1], Lmax, nsplit, float(rms_eps),
                                            float(ln_eps), _stream()), "ld_llm_decode_forward")


LLM_FUSED_CTL_WORDS = 512          # LD_LLM_FUSED_CTL_WORDS


def llm_layer_table_device(table, device):
    """The ld_llm_layer array as a device tensor (the persistent decode kernel reads the table itself)."""
    raw = bytes(memoryview(table).cast("B"))
    return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)


def llm_decode_forward_fused(table_dev, n_layers, emb, token, pos, x, qkv, att, gate, attn_ws, cos_t, sin_t, lnf_w, lnf_b, lnf_out,
                             head, logits, heads, Lmax, nsplit, rms_eps, ln_eps, ctl):
    """llm_decode_forward with all blocks in one persistent launch (ld_llm_fused.hip); ctl: int32 [LLM_FUSED_CTL_WORDS], zeroed
    before the first step of a decode.  Raises LandiffHipError(unsupported) outside the fused form's shapes."""
    B, hidden = x.shape
    for t in (x, qkv, att, gate, lnf_out, logits, head):
        assert t.is_contiguous()
    assert ctl.dtype == torch.int32 and ctl.numel() >= LLM_FUSED_CTL_WORDS
    check(_lib.load().ld_llm_decode_forward_fused(_ptr(table_dev), n_layers, _ptr(emb), _ptr(token), _ptr(pos), _ptr(x), _ptr(qkv),
                                                  _ptr(att), _ptr(gate), _ptr(attn_ws), _ptr(cos_t), _ptr(sin_t), _ptr(lnf_w),
                                                  _ptr(lnf_b), _ptr(lnf_out), _ptr(head), _ptr(logits), B, hidden, heads,
                                                  gate.shape[1], logits.shape[1], Lmax, nsplit, float(rms_eps), float(ln_eps),
                                                  _ptr(ctl), _stream()), "ld_llm_decode_forward_fused")


LLM_CHAIN_CTL_WORDS = 64 + 256 * 8 * 16     # LD_LLM_CHAIN_CTL_WORDS


def llm_decode_blocks_chained(table, pos_value, x, qkv, att, gate, attn_ws, cos_t, sin_t, heads, Lmax, nsplit, rms_eps, ctl, epoch,
                              stream0, stream1):
    """All blocks of one decode step as dependent launches alternating between two streams (ld_llm_decode_blocks_chained): the
    caller orders stream1 after the producers of the KV cache before the first step and the readers of x after both streams."""
    B, hidden = x.shape
    for t in (x, qkv, att, gate):
        assert t.is_contiguous()
    assert ctl.dtype == torch.int32 and ctl.numel() >= LLM_CHAIN_CTL_WORDS
    check(_lib.load().ld_llm_decode_blocks_chained(ctypes.addressof(table), len(table), int(pos_value), _ptr(x), _ptr(qkv), _ptr(att),
                                                   _ptr(gate), _ptr(attn_ws), _ptr(cos_t), _ptr(sin_t), B, hidden, heads,
                                                   gate.shape[1], Lmax, nsplit, float(rms_eps), _ptr(ctl), int(epoch),
                                                   ctypes.c_void_p(stream0.cuda_stream), ctypes.c_void_p(stream1.cuda_stream)),
          "ld_llm_decode_blocks_chained")


def llm_logits_to_probs(logits, probs, cfg_logits, guided, scale, temperature, pos=None, allowed=None,
                        top_k=None, top_p=None):
    V = probs.shape[-1]
    check(_lib.load().ld_llm_logits_to_probs(_ptr(logits), _ptr(probs), _ptr(cfg_logits), V, int(guided), float(scale),
                                             float(temperature), _ptr(pos), _ptr(allowed),
                                             allowed.stride(0) if allowed is not None else 0,
                                             int(top_k) if top_k is not None else 0,
                                             float(top_p) if top_p is not None else -1.0, _stream()),
          "ld_llm_logits_to_probs")


def llm_sample_advance(logits, probs, cfg_logits, guided, scale, temperature, pos, allowed, noise, forced, token, out_tokens,
                       out_count, sampled, emb, x, top_k=None, top_p=None):
    """probabilities -> draw (argmax(p / noise), noise ~ Exp(1): torch.multinomial's own formula) -> forced-token schedule,
    token record, position advance -> embedding rows of the next token in x: the tail of a decode step in one launch."""
    V = logits.shape[-1]
    B, D = x.shape
    check(_lib.load().ld_llm_sample_advance(_ptr(logits), _ptr(probs), _ptr(cfg_logits), V, int(guided), float(scale),
                                            float(temperature), _ptr(pos), _ptr(allowed),
                                            allowed.stride(0) if allowed is not None else 0,
                                            int(top_k) if top_k is not None else 0,
                                            float(top_p) if top_p is not None else -1.0, _ptr(noise), _ptr(forced),
                                            _ptr(token), _ptr(out_tokens), _ptr(out_count), _ptr(sampled), _ptr(emb),
                                            _ptr(x), B, D, _stream()), "ld_llm_sample_advance")


def llm_decode_advance(sampled, forced, pos, token, out_tokens, out_count):
    check(_lib.load().ld_llm_decode_advance(_ptr(sampled), _ptr(forced), _ptr(pos), _ptr(token), _ptr(out_tokens),
                                            _ptr(out_count), _stream()), "ld_llm_decode_advance")


# ------------------------------------------------------------------------------------------------
# normalisation
# ------------------------------------------------------------------------------------------------
def layernorm(x, w, b, out, eps, *, mod=None, mod_bstride=0, shift_img=0, scale_img=0, shift_txt=0, scale_txt=0,
              rows_per_batch=0, text_len=0):
    rows, D = x.shape
    assert x.stride(1) == 1 and out.stride(1) == 1
    check(_lib.load().ld_layernorm(_ptr(x), x.stride(0), int(x.dtype == torch.float32), _ptr(w), _ptr(b), _ptr(out),
                                   out.stride(0), int(out.dtype == torch.float32), rows, D, float(eps), _ptr(mod),
                                   mod_bstride, shift_img, scale_img, shift_txt, scale_txt, rows_per_batch, text_len,
                                   _stream()), "ld_layernorm")
    return out


def layernorm_mxfp8(x, w, b, q, scales, eps, *, mod=None, mod_bstride=0, shift_img=0, scale_img=0, shift_txt=0, scale_txt=0,
                    rows_per_batch=0, text_len=0):
    """layernorm(...) whose bf16 result is written as MXFP8: q uint8 [rows, D], scales uint8 [D/128, rows, 4]."""
    _bf16(x, "x")
    rows, D = x.shape
    assert x.stride(1) == 1 and q.dtype == torch.uint8 and q.shape == (rows, D) and q.stride(1) == 1
    assert scales.dtype == torch.uint8 and tuple(scales.shape) == (D // 128, rows, 4) and scales.is_contiguous()
    check(_lib.load().ld_layernorm_mxfp8(_ptr(x), x.stride(0), _ptr(w), _ptr(b), _ptr(q), q.stride(0), _ptr(scales),
                                         rows, rows, D, float(eps), _ptr(mod), mod_bstride, shift_img, scale_img,
                                         shift_txt, scale_txt, rows_per_batch, text_len, _stream()), "ld_layernorm_mxfp8")
    return q, scales


def feature_norm_cl(features, mean, std, out, T, C, P):
    """features [T,C,h,w] (fp32/bf16) -> out [T*P, C] bf16 = (x - mean[c]) / (std[c] + 1e-8)."""
    assert features.is_contiguous() and out.is_contiguous() and out.dtype == torch.bfloat16
    assert features.dtype in (torch.float32, torch.bfloat16) and mean.dtype == torch.float32 and std.dtype == torch.float32
    check(_lib.load().ld_feature_norm_cl(_ptr(features), int(features.dtype == torch.float32), _ptr(mean), _ptr(std), _ptr(out),
                                         T, C, P, _stream()), "ld_feature_norm_cl")


def feature_denorm(x, mean, std, out=None):
    """x bf16 [rows, C] -> bf16(float(x) * (std[c] + 1e-8) + mean[c]) (VideoVQ.denorm_features when a mean_std_path is configured)."""
    _bf16(x, "x")
    assert x.is_contiguous() and mean.dtype == torch.float32 and std.dtype == torch.float32
    out = x if out is None else out
    assert out.is_contiguous() and out.dtype == torch.bfloat16 and out.shape == x.shape
    check(_lib.load().ld_feature_denorm(_ptr(x), _ptr(mean), _ptr(std), _ptr(out), x.shape[0], x.shape[1], _stream()),
          "ld_feature_denorm")
    return out


def vq_nearest(x, codebook, idx, dim):
    """x bf16 [rows, >=dim], codebook fp32 [V, dim] -> idx int64 [rows] (first code at minimum Euclidean distance)."""
    _bf16(x, "x")
    assert codebook.dtype == torch.float32 and codebook.is_contiguous() and codebook.shape[1] == dim
    assert idx.dtype == torch.int64 and idx.is_contiguous() and x.stride(1) == 1
    check(_lib.load().ld_vq_nearest(_ptr(x), x.stride(0), _ptr(codebook), _ptr(idx), x.shape[0], codebook.shape[0], dim,
                                    _stream()), "ld_vq_nearest")


def gemm_qkv_heads(a, w, bias, q, k, vt, B, N, H, Npad, ln, eps=1e-6):
    """The DiT qkv Linear with the head split in its epilogue: a [B*N, K] -> q, k [B,H,Npad,64] (QK-LayerNorm with
    ln = (q_w, q_b, k_w, k_b)) and vt [B,H,64,Npad].  Rows [N, Npad) of q / k / vt are left as they are (keep them zero)."""
    _bf16(a, "a"); _bf16(w, "w")
    assert a.shape[0] == B * N and w.shape == (3 * H * 64, a.shape[1]) and a.stride(1) == 1 and w.is_contiguous()
    assert q.shape == (B, H, Npad, 64) and k.shape == q.shape and vt.shape == (B, H, 64, Npad)
    assert q.is_contiguous() and k.is_contiguous() and vt.is_contiguous()
    check(_lib.load().ld_gemm_qkv_heads(_ptr(a), a.stride(0), _ptr(w), _ptr(bias), B * N, a.shape[1], _ptr(q), _ptr(k), _ptr(vt),
                                        B, N, H, Npad, _ptr(ln[0]), _ptr(ln[1]), _ptr(ln[2]), _ptr(ln[3]), float(eps), _stream()),
          "ld_gemm_qkv_heads")


def gemm_qkv_heads_mxfp8(a8, sa, w8, sw, bias, q, k, vt, B, N, H, Npad, ln, eps=1e-6):
    """gemm_qkv_heads on MXFP8 operands: a8 uint8 [B*N, K] + scales sa [K/128, B*N, 4], w8 [3*H*64, K] + sw [K/128, 3*H*64, 4]."""
    assert a8.dtype == torch.uint8 and w8.dtype == torch.uint8 and a8.stride(1) == 1 and w8.is_contiguous()
    M, K = a8.shape
    assert M == B * N and w8.shape == (3 * H * 64, K) and tuple(sa.shape) == (K // 128, M, 4) and tuple(sw.shape) == (K // 128, 3 * H * 64, 4)
    assert sa.is_contiguous() and sw.is_contiguous() and sa.dtype == torch.uint8 and sw.dtype == torch.uint8
    assert q.shape == (B, H, Npad, 64) and k.shape == q.shape and vt.shape == (B, H, 64, Npad)
    assert q.is_contiguous() and k.is_contiguous() and vt.is_contiguous()
    check(_lib.load().ld_gemm_qkv_heads_mxfp8(_ptr(a8), a8.stride(0), _ptr(sa), _ptr(w8), _ptr(sw), _ptr(bias), M, K, _ptr(q), _ptr(k), _ptr(vt),
                                              B, N, H, Npad, _ptr(ln[0]), _ptr(ln[1]), _ptr(ln[2]), _ptr(ln[3]), float(eps), _stream()),
          "ld_gemm_qkv_heads_mxfp8")


def qkv_split(qkv, q, k, vt, B, N, H, Npad, *, ln=None, rope=None, eps=1e-6):
    """ln = (q_w, q_b, k_w, k_b) for the DiT QK-LayerNorm, or rope = (cos, sin) [N,32] fp32 for TiTok."""
    mode = 0 if ln is not None else 1
    lw = ln if ln is not None else (None,) * 4
    rp = rope if rope is not None else (None, None)
    check(_lib.load().ld_qkv_split(_ptr(qkv), _ptr(q), _ptr(k), _ptr(vt), B, N, H, Npad, mode, _ptr(lw[0]), _ptr(lw[1]),
                                   _ptr(lw[2]), _ptr(lw[3]), float(eps), _ptr(rp[0]), _ptr(rp[1]), _stream()),
          "ld_qkv_split")


def groupnorm_stats(x, stats, F, P, C, G):
    """stats [F, G, 2] float64 <- (sum, sum of squares) per frame-group and channel group; deterministic (fixed-order
    reduction of per-workgroup partials through a workspace from torch's caching allocator)."""
    assert stats.dtype == torch.float64 and stats.is_contiguous() and stats.numel() == F * G * 2
    lib = _lib.load()
    ws = torch.empty(F * int(lib.ld_groupnorm_stats_blocks(P)) * G * 2, device=stats.device, dtype=torch.float64)
    check(lib.ld_groupnorm_stats(_ptr(x), _ptr(stats), _ptr(ws), F, P, C, G, _stream()), "ld_groupnorm_stats")


GN_FOLD_BLOCKS = 256      # ld_norm.hip: LD_GN_FOLD_BLOCKS


def groupnorm_stats_from_conv(part, stats, P, C, G):
    """stats [1, G, 2] float64 <- the partial sums conv_cl(..., gn_partials=True) left behind (P output rows, C channels);
    deterministic (fixed-order fold + reduce)."""
    assert stats.dtype == torch.float64 and stats.is_contiguous() and stats.numel() == G * 2
    assert part.dtype == torch.float32 and part.is_contiguous() and part.numel() == (P + 63) // 64 * (C // 4) * 2
    ws = torch.empty(GN_FOLD_BLOCKS * G * 2, device=stats.device, dtype=torch.float64)
    check(_lib.load().ld_groupnorm_stats_from_conv(_ptr(part), _ptr(stats), _ptr(ws), P, C, G, _stream()),
          "ld_groupnorm_stats_from_conv")


def groupnorm_apply(x, out_padded, stats, gamma, beta, F, T, H, W, C, G, *, zy=None, zb=None, zshape=(1, 1, 1),
                    tpad=0, hpad=0, wpad=0, swish=True, eps=1e-6):
    check(_lib.load().ld_groupnorm_apply(_ptr(x), _ptr(out_padded), _ptr(stats), _ptr(gamma), _ptr(beta), _ptr(zy), _ptr(zb),
                                         F, T, H, W, C, G, zshape[0], zshape[1], zshape[2], tpad, hpad, wpad, int(swish),
                                         float(eps), _stream()), "ld_groupnorm_apply")


# ------------------------------------------------------------------------------------------------
# layout / elementwise
# ------------------------------------------------------------------------------------------------
def patchify(x, sem, out, p):
    B, T, C, H, W = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous()
    check(_lib.load().ld_patchify(_ptr(x), _ptr(sem), _ptr(out), B, T, C, H, W, p, _stream()), "ld_patchify")


def unpatchify_cfg(lin, x, out, p, c_out, c_skip, scale):
    _, T, C, H, W = x.shape
    check(_lib.load().ld_unpatchify_cfg(_ptr(lin), _ptr(x), _ptr(out), T, C, H, W, p, float(c_out), float(c_skip),
                                        float(scale), _stream()), "ld_unpatchify_cfg")


def axpbypcz(out, x, a, y=None, b=0.0, z=None, c=0.0):
    check(_lib.load().ld_axpbypcz(_ptr(out), _ptr(x), float(a), _ptr(y), float(b), _ptr(z), float(c), out.numel(),
                                  _stream()), "ld_axpbypcz")
    return out


def timestep_embedding(t, out, max_period=10000.0):
    B, dim = out.shape
    check(_lib.load().ld_timestep_embedding(_ptr(t), _ptr(out), B, dim, float(max_period), _stream()),
          "ld_timestep_embedding")


def place_cl(x, out_padded, F, Ti, Hi, Wi, Cin, Cout, *, mode=0, time_up=False, tpad=0, hpad=1, wpad=1):
    check(_lib.load().ld_place_cl(_ptr(x), _ptr(out_padded), F, Ti, Hi, Wi, Cin, Cout, mode, int(time_up), tpad, hpad,
                                  wpad, _stream()), "ld_place_cl")


def to_uint8(x, out, video, P):
    check(_lib.load().ld_to_uint8(_ptr(x), x.stride(-2), _ptr(out), _ptr(video), P, _stream()), "ld_to_uint8")


def latent_to_cl(x, out, T, C, H, W, Cpad, mul, src_tchw):
    check(_lib.load().ld_latent_to_cl(_ptr(x), _ptr(out), T, C, H, W, Cpad, float(mul), int(src_tchw), _stream()),
          "ld_latent_to_cl")


# ---- T5 text encoders -------------------------------------------------------------------------------
def t5_rmsnorm(x, w, out, eps):
    _bf16(x, "x"); _bf16(w, "w"); _bf16(out, "out")
    rows, D = x.shape
    check(_lib.load().ld_t5_rmsnorm(_ptr(x), _ptr(w), _ptr(out), rows, D, float(eps), _stream()), "ld_t5_rmsnorm")
    return out


def t5_attn(q, k, v, out, bias_table, bucket, H):
    """q, k, v, out: [N, ld] bf16 views (last dim contiguous, same row stride); bucket int32 [2N-1]."""
    N = q.shape[0]
    ld = q.stride(0)
    assert k.stride(0) == ld and v.stride(0) == ld and out.stride(0) == ld and bucket.dtype == torch.int32
    check(_lib.load().ld_t5_attn(_ptr(q), _ptr(k), _ptr(v), _ptr(out), ld, _ptr(bias_table), _ptr(bucket), N, H, _stream()),
          "ld_t5_attn")
    return out

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


def conv_cl(x_padded: torch.Tensor, w: torch.Tensor, T: int, H: int, W: int,
            out: torch.Tensor | None = None, **epi) -> torch.Tensor:
    """Channels-last conv.  x_padded [T+kT-1, H+kH-1, W+kW-1, Cin]; w [Cout, kT, kH, kW, Cin]."""
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
    check(lib.ld_conv_cl_bf16(_ptr(x_padded), _ptr(w), _ptr(out), out.stride(-2), T, H, W, Cin, Cout,
                              kT, kH, kW, ctypes.byref(e), _stream()), "ld_conv_cl_bf16")
    return out


def attn_fwd(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, out: torch.Tensor, Nq: int, Nk: int,
             scale: float, fid_q=None, fid_k=None, kt_min=None, kt_max=None) -> torch.Tensor:
    """q,k [B,H,Npad,64]; vt [B,H,64,Npad]; out [B,N,H*64] (bf16).  Optional frame mask arrays (int32)."""
    _bf16(q, "q"); _bf16(k, "k"); _bf16(vt, "vt"); _bf16(out, "out")
    B, H, Npad, D = q.shape
    assert D == 64 and k.shape == q.shape and tuple(vt.shape) == (B, H, 64, Npad)
    assert q.is_contiguous() and k.is_contiguous() and vt.is_contiguous() and out.stride(-1) == 1
    assert out.shape[0] == B and out.shape[2] == H * 64
    lib = _lib.load()
    check(lib.ld_attn_fwd_bf16(_ptr(q), _ptr(k), _ptr(vt), _ptr(out), B, H, Nq, Nk, Npad,
                               out.stride(0), out.stride(1), float(scale),
                               _ptr(fid_q), _ptr(fid_k), _ptr(kt_min), _ptr(kt_max), _stream()),
          "ld_attn_fwd_bf16")
    return out

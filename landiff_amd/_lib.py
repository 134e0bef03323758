"""ctypes loader for liblandiff_hip.so (the only compute backend; fails loudly if absent)."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_int, c_int32, c_int64, c_uint32, c_void_p, c_float, c_double

_HERE = os.path.dirname(os.path.abspath(__file__))
# LANDIFF_HIP_LIB: another build of the same library (tools/: A/B of compile-time variants); the product never sets it
LIB_PATH = os.environ.get("LANDIFF_HIP_LIB") or os.path.join(_HERE, "liblandiff_hip.so")


class LandiffHipError(RuntimeError):
    pass


class Epilogue(Structure):
    """Mirror of ld_epilogue_t (include/landiff_hip.h)."""

    _fields_ = [
        ("bias", c_void_p),
        ("act", c_int32),
        ("mul", c_void_p),
        ("ldmul", c_int64),
        ("resid", c_void_p),
        ("ldr", c_int64),
        ("resid_f32", c_int32),
        ("gate", c_void_p),
        ("gate_bstride", c_int64),
        ("gate_off_img", c_int64),
        ("gate_off_txt", c_int64),
        ("rows_per_batch", c_int32),
        ("text_len", c_int32),
        ("add2", c_void_p),
        ("ldadd", c_int64),
        ("out_f32", c_int32),
    ]


class LlmLayer(Structure):
    """Mirror of ld_llm_layer (include/landiff_hip.h): one decoder block's weights and its KV cache."""

    _fields_ = [(n, c_void_p) for n in ("wqkv", "wo", "w1", "w3", "w2", "n0", "n1", "k_cache", "v_cache")]


_lib = None
ABI_VERSION = 10         # LD_ABI_VERSION of include/landiff_hip.h that SIGNATURES below were written against

I64 = c_int64
I32 = c_int32
P = c_void_p

# name -> argtypes; every symbol declared in include/landiff_hip.h must be listed here
SIGNATURES: dict[str, list] = {
    "ld_version": [],
    "ld_last_error": [],
    "ld_gemm_bf16": [P, I64, P, P, I64, I64, I64, I64, POINTER(Epilogue), P],
    "ld_gemm_qkv_heads": [P, I64, P, P, I64, I64, P, P, P, I64, I64, I64, I64, P, P, P, P, c_float, P],
    "ld_conv_cl_bf16": [P, P, P, I64, I64, I64, I64, I64, I64, I64, I64, I64, POINTER(Epilogue), P],
    "ld_conv_cl_bf16_gn": [P, P, P, I64, I64, I64, I64, I64, I64, I64, I64, I64, POINTER(Epilogue), P, P],
    "ld_conv_gn_partials_size": [I64, I64],
    "ld_conv_route": [I64, I64, I64, I64, I64, I64, I64, I64],
    "ld_calib_mfma_bf16": [P, I64, P, I64, I64, POINTER(c_double), P],
    "ld_calib_stream_read": [P, I64, P, P],
    "ld_attn_fwd_bf16": [P, P, P, P, I64, I64, I64, I64, I64, I64, I64, c_float, P, P, P, P, P],
    "ld_attn_fwd_bf16_exact": [P, P, P, P, I64, I64, I64, I64, I64, I64, I64, c_float, P, P, P, P, P],
    "ld_attn_last_kernel": [],
    "ld_attn_last_fallbacks": [P, P],
    "ld_reset": [P],
    "ld_attn_queue_poke": [I32, c_uint32, P],
    "ld_gemv": [P, I64, I32, P, P, I32, P, P, I64, P, I64, I32, I64, I64, I64, I32, I32, P, c_float, P],
    "ld_rmsnorm_bf16": [P, P, P, I64, I64, c_float, P],
    "ld_layernorm_bf16_to_f32": [P, I64, P, P, P, I64, I64, c_float, P],
    "ld_llm_rope_append": [P, P, P, P, P, P, P, I64, I64, I64, I64, P],
    "ld_llm_kv_attn": [P, P, P, P, P, I64, I64, I64, I64, P, I64, P, P, P, P],
    "ld_llm_embed": [P, P, P, I64, I64, P],
    "ld_quantize_fp8": [P, I64, P, I64, P, I64, I64, P],
    "ld_gemm_fp8": [P, I64, P, P, P, P, I64, I64, I64, I64, P, P],
    "ld_quantize_mxfp8": [P, I64, P, I64, P, I64, I64, I64, P],
    "ld_layernorm_mxfp8": [P, I64, P, P, P, I64, P, I64, I64, I64, c_float, P, I64, I64, I64, I64, I64, I64, I64, P],
    "ld_gemm_mxfp8": [P, I64, P, P, P, P, I64, P, I64, I64, I64, I64, P, P],
    "ld_gemm_qkv_heads_mxfp8": [P, I64, P, P, P, P, I64, I64, P, P, P, I64, I64, I64, I64, P, P, P, P, c_float, P],
    "ld_feature_norm_cl": [P, I32, P, P, P, I64, I64, I64, P],
    "ld_feature_denorm": [P, P, P, P, I64, I64, P],
    "ld_vq_nearest": [P, I64, P, P, I64, I64, I64, P],
    "ld_llm_decode_forward": [P, I64, P, P, P, I32, P, P, P, P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64,
                              c_float, c_float, P],
    "ld_llm_logits_to_probs": [P, P, P, I64, I32, c_float, c_float, P, P, I64, I32, c_float, P],
    "ld_llm_decode_advance": [P, P, P, P, P, P, P],
    "ld_llm_sample_advance": [P, P, P, I64, I32, c_float, c_float, P, P, I64, I32, c_float, P, P, P, P, P, P, P, P, I64, I64, P],
    "ld_layernorm": [P, I64, I32, P, P, P, I64, I32, I64, I64, c_float, P, I64, I64, I64, I64, I64, I64, I64, P],
    "ld_qkv_split": [P, P, P, P, I64, I64, I64, I64, I32, P, P, P, P, c_float, P, P, P],
    "ld_groupnorm_stats_blocks": [I64],
    "ld_groupnorm_stats": [P, P, P, I64, I64, I64, I64, P],
    "ld_groupnorm_stats_from_conv": [P, P, P, I64, I64, I64, P],
    "ld_groupnorm_apply": [P, P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, I64, I64, I64, I64, I64, I32, c_float, P],
    "ld_patchify": [P, P, P, I64, I64, I64, I64, I64, I64, P],
    "ld_unpatchify_cfg": [P, P, P, I64, I64, I64, I64, I64, c_float, c_float, c_float, P],
    "ld_axpbypcz": [P, P, c_float, P, c_float, P, c_float, I64, P],
    "ld_timestep_embedding": [P, P, I64, I64, c_float, P],
    "ld_place_cl": [P, P, I64, I64, I64, I64, I64, I64, I32, I32, I64, I64, I64, P],
    "ld_to_uint8": [P, I64, P, P, I64, P],
    "ld_latent_to_cl": [P, P, I64, I64, I64, I64, I64, c_float, I32, P],
    "ld_t5_rmsnorm": [P, P, P, I64, I64, c_float, P],
    "ld_t5_attn": [P, P, P, P, I64, P, P, I64, I64, P],
}

# entry points that only the variants build exports (include/landiff_hip.h under LD_VARIANTS; landiff_amd/csrc/build.sh with
# LD_BUILD_VARIANTS=1 -> VARIANTS_LIB_PATH): bound when the loaded library has them
VARIANT_SIGNATURES: dict[str, list] = {
    "ld_llm_decode_blocks_chained": [P, I64, I32, P, P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, c_float, P, c_uint32, P, P],
    "ld_llm_decode_blocks_fused": [P, I64, P, P, P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, c_float, P, P],
    "ld_llm_decode_forward_fused": [P, I64, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64,
                                    c_float, c_float, P, P],
}
VARIANTS_LIB_PATH = os.path.join(_HERE, "variants", "liblandiff_hip_variants.so")


def has_variants() -> bool:
    """True when the loaded library is the variants build (LANDIFF_HIP_LIB pointed at it)."""
    return hasattr(load(), "ld_llm_decode_blocks_chained")


def load():
    """Load the shared library (once).  Raises LandiffHipError if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LandiffHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    lib.ld_version.restype = c_int
    if lib.ld_version() != ABI_VERSION:
        raise LandiffHipError(f"{LIB_PATH} reports ABI version {lib.ld_version()}, this binding was written for {ABI_VERSION}: "
                              "rebuild the library (__graft_entry__.build()) -- the argument lists would not match")
    sigs = dict(SIGNATURES)
    sigs.update({k: v for k, v in VARIANT_SIGNATURES.items() if hasattr(lib, k)})
    for name, argtypes in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = (c_char_p if name in ("ld_last_error", "ld_attn_last_kernel") else
                      ctypes.c_int64 if name in ("ld_groupnorm_stats_blocks", "ld_conv_gn_partials_size") else c_int)
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().ld_last_error()
        raise LandiffHipError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

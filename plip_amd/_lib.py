"""ctypes binding of libplipmi.so (include/plipmi.h).  No fallback: if the HIP
extension is missing or fails to load, importing callers get a RuntimeError."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libplipmi.so")

F32, BF16, F16 = 0, 1, 2
# plipmi_config.flags
FLAG_SEPARATE_LAYERNORM, FLAG_DENSE_LAST_BLOCK, FLAG_PACK_CAPTIONS, FLAG_VALU_ATTENTION, FLAG_TEXT_TOWER_F16 = 1, 2, 4, 8, 16
VISION, TEXT = 0, 1


class Config(C.Structure):
    # first member = sizeof(Config) as this binding knows it (plipmi_create copies that many bytes: the struct may grow)
    _fields_ = [(n, C.c_int32) for n in (
        "struct_size", "image_size", "patch_size", "v_width", "v_layers", "v_heads", "v_mlp", "vocab_size",
        "context_length", "t_width", "t_layers", "t_heads", "t_mlp", "projection_dim")] + [
        ("layer_norm_eps", C.c_float), ("compute_dtype", C.c_int32), ("max_batch", C.c_int32), ("flags", C.c_int32),
        ("graph_batch", C.c_int32), ("text_f16_layers", C.c_int32), ("pass_batch", C.c_int32)]


_LAYER_FIELDS = ("ln1_w", "ln1_b", "q_w", "q_b", "k_w", "k_b", "v_w", "v_b", "o_w", "o_b",
                 "ln2_w", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")


class LayerWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _LAYER_FIELDS]


class Weights(C.Structure):
    _fields_ = [
        ("v_class_embedding", C.c_void_p), ("v_patch_weight", C.c_void_p), ("v_pos_embedding", C.c_void_p),
        ("v_pre_ln_w", C.c_void_p), ("v_pre_ln_b", C.c_void_p), ("v_post_ln_w", C.c_void_p),
        ("v_post_ln_b", C.c_void_p), ("visual_projection", C.c_void_p), ("v_layers", C.POINTER(LayerWeights)),
        ("t_token_embedding", C.c_void_p), ("t_pos_embedding", C.c_void_p), ("t_final_ln_w", C.c_void_p),
        ("t_final_ln_b", C.c_void_p), ("text_projection", C.c_void_p), ("t_layers", C.POINTER(LayerWeights)),
    ]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("calls", C.c_int64), ("total_ms", C.c_double),
                ("flops", C.c_double), ("bytes", C.c_double)]


_vp, _i, _f = C.c_void_p, C.c_int, C.c_float
# kernel-level test entries and A/B hooks (include/plipmi_test.h): tests/, tools/ and bench.py's side fields only
TEST_SYMBOLS = {
    "plipmi_debug_hidden": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp]),
    "plipmi_gemm_nt": (_i, [_i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _f, _vp, _vp]),
    "plipmi_gemm_variant_name": (C.c_char_p, [_i]),
    "plipmi_test_force_gemm_tile": (_i, [_i]),
    "plipmi_test_remap_gemm_tile": (_i, [_i, _i]),
    "plipmi_test_fused_qkv_attention": (_i, [_i]),
    "plipmi_test_patch_gather": (_i, [_i]),
    "plipmi_test_reset_hooks": (None, []),
    "plipmi_gemm_variant_built": (_i, [_i, _i]),
    "plipmi_recode_planes": (_i, [_vp, _vp, C.c_size_t, _i, _i, _i, _vp]),
    "plipmi_gemm_nt_ln": (_i, [_i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp]),
    "plipmi_gemm_nt_ld": (_i, [_i, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _f, _vp, _vp]),
    "plipmi_gemm_nt_traced": (_i, [_i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp]),
    "plipmi_attention": (_i, [_i, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "plipmi_qkv_attention": (_i, [_i, _vp, _vp, _vp, _vp, _i, _f, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
}
# every symbol include/plipmi.h declares (the product interface): (restype, argtypes)
SYMBOLS = {
    "plipmi_create": (_i, [C.POINTER(Config), C.POINTER(Weights), _vp, C.POINTER(_vp)]),
    "plipmi_destroy": (None, [_vp]),
    "plipmi_version": (_i, []),
    "plipmi_last_error": (C.c_char_p, []),
    "plipmi_device_name": (C.c_char_p, [_vp]),
    "plipmi_encode_image": (_i, [_vp, _vp, _i, _vp, _i, _vp]),
    "plipmi_encode_image_u8": (_i, [_vp, _vp, _i, _vp, _i, _vp]),
    "plipmi_encode_text": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _vp]),
    "plipmi_check_async": (_i, [_vp]),
    "plipmi_set_latency_batch": (_i, [_vp, _i]),
    "plipmi_set_graph_batch": (_i, [_vp, _i]),
    "plipmi_get_pass_batch": (_i, [_vp]),
    "plipmi_clone": (_i, [_vp, C.POINTER(_vp)]),
    "plipmi_streams_overlap": (_i, [_vp, _vp, _vp, C.POINTER(_f)]),
    "plipmi_l2_normalize": (_i, [_vp, _vp, _i, _i, _vp]),
    "plipmi_logits": (_i, [_vp, _vp, _i, _vp, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "plipmi_topk": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "plipmi_resize_crop_u8": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "plipmi_similarity_topk": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "plipmi_set_text_packing": (_i, [_vp, _i]),
    "plipmi_profile_enable": (_i, [_vp, _i]),
    "plipmi_profile_read": (_i, [_vp, C.POINTER(KernelStat), _i, C.POINTER(_i)]),
}

_lib = None


def load():
    """Load libplipmi.so once.  Fails loudly -- there is no CPU/eager fallback."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # torch ships its own libamdhip64; load it FIRST so libplipmi.so binds to the same HIP runtime instance
        # (two runtimes in one process do not share devices/streams: plipmi_create then sees "no device")
        import torch  # noqa: F401
    except ImportError:  # pragma: no cover - the host layer needs torch anyway
        pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP engine is not built. Run `python -m plip_amd.build` "
            "(needs hipcc / ROCm). plip_amd has no fallback path.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise RuntimeError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in {**SYMBOLS, **TEST_SYMBOLS}.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().plipmi_last_error().decode(errors="replace")


class PlipmiError(RuntimeError):
    pass


ERR_TOKEN_ID = 5     # include/plipmi.h PLIPMI_ERR_TOKEN_ID


def check(rc: int, what: str) -> None:
    if rc == ERR_TOKEN_ID:      # what the reference's embedding lookup raises on an out-of-range id (plip.py:68)
        raise IndexError(f"{what}: {last_error()}")
    if rc != 0:
        raise PlipmiError(f"{what} failed (code {rc}): {last_error()}")

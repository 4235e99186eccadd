"""ctypes binding of libmoge_hip.so (include/moge_hip.h).  The product path: there is NO fallback - if the HIP
library is missing or does not load, importing this module raises."""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must be imported first: libmoge_hip.so binds to the libamdhip64.so.7 torch has loaded)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libmoge_hip.so")

MOGE_MAX_TAPS = 8
MOGE_LEVELS = 5
FP32, FP16, FP16_HALF = 0, 1, 2        # moge_precision: FP16 = fp32 weights under autocast (fp32 residual stream), FP16_HALF = model.half() (fp16 residual stream)
HEAD_POINTS, HEAD_NORMAL, HEAD_MASK, HEAD_SCALE = 1, 2, 4, 8
FORCE_PROJECTION, APPLY_MASK = 1, 2
REMAP = {"linear": 0, "sinh": 1, "exp": 2, "sinh_exp": 3}
RESAMPLER = {"conv_transpose": 0, "bilinear": 1, "nearest": 2, "pixel_shuffle": 3}      # moge_resampler (x2 up-samplers, modules.py:139-181)
RES_NORM = {"none": 0, "layer_norm": 1, "group_norm": 2, "instance_norm": 3}            # moge_res_norm (modules.py:47-60)
ACTIVATION = {"relu": 0, "leaky_relu": 1, "silu": 2, "elu": 3}                          # moge_activation (modules.py:31-40)
ERR_NONFINITE = -5
KC_NAMES = ["gemm", "attn", "conv", "norm", "pre", "post", "recover", "gemm_pp"]
ABI_VERSION = 5


class MogeConfig(C.Structure):
    _fields_ = [("embed_dim", C.c_int32), ("depth", C.c_int32), ("num_heads", C.c_int32), ("n_taps", C.c_int32),
                ("taps", C.c_int32 * MOGE_MAX_TAPS), ("dims", C.c_int32 * MOGE_LEVELS),
                ("neck_res_blocks", C.c_int32 * MOGE_LEVELS), ("head_res_blocks", C.c_int32 * MOGE_LEVELS),
                ("heads", C.c_int32), ("scale_hidden", C.c_int32), ("remap_output", C.c_int32),
                ("neck_resamplers", C.c_int32 * (MOGE_LEVELS - 1)), ("head_resamplers", C.c_int32 * (MOGE_LEVELS - 1)),
                ("neck_in_norm", C.c_int32), ("neck_hidden_norm", C.c_int32), ("head_in_norm", C.c_int32), ("head_hidden_norm", C.c_int32),
                ("neck_activation", C.c_int32), ("head_activation", C.c_int32), ("neck_hidden_mult", C.c_int32), ("head_hidden_mult", C.c_int32)]


MOGE_V1_MAX_UP = 4


class MogeV1Config(C.Structure):
    _fields_ = [("embed_dim", C.c_int32), ("depth", C.c_int32), ("num_heads", C.c_int32), ("n_taps", C.c_int32),
                ("taps", C.c_int32 * MOGE_MAX_TAPS), ("dim_proj", C.c_int32), ("n_up", C.c_int32), ("dim_upsample", C.c_int32 * MOGE_V1_MAX_UP),
                ("num_res_blocks", C.c_int32), ("last_conv_channels", C.c_int32), ("remap_output", C.c_int32), ("mask_threshold", C.c_float),
                ("hidden_mult", C.c_int32), ("res_block_norm", C.c_int32), ("last_res_blocks", C.c_int32), ("last_conv_size", C.c_int32)]


class TensorDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("numel", C.c_int64)]


class Outputs(C.Structure):
    _fields_ = [("points", C.c_void_p), ("depth", C.c_void_p), ("normal", C.c_void_p), ("mask_prob", C.c_void_p),
                ("mask", C.c_void_p), ("intrinsics", C.c_void_p), ("metric_scale", C.c_void_p),
                ("focal", C.c_void_p), ("shift", C.c_void_p)]


class Profile(C.Structure):
    _fields_ = [("ms", C.c_double * 8), ("flops", C.c_double * 8), ("bytes", C.c_double * 8), ("launches", C.c_int64 * 8)]


class TestGemmArgs(C.Structure):
    """moge_test_gemm_args (tests only): one GEMM through a chosen fused epilogue."""
    _fields_ = [("precision", C.c_int32), ("kind", C.c_int32), ("act", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("A", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("out", C.c_void_p), ("ln_mr", C.c_void_p), ("ln_c", C.c_void_p),
                ("wu", C.c_void_p), ("wv", C.c_void_p), ("u0", C.c_float), ("u1", C.c_float), ("v0", C.c_float), ("v1", C.c_float),
                ("pixW", C.c_int32), ("pixH", C.c_int32), ("Cout", C.c_int32),
                ("xres", C.c_void_p), ("gamma", C.c_void_p), ("x16_out", C.c_void_p), ("ln_part_out", C.c_void_p),
                ("q_out", C.c_void_p), ("k_out", C.c_void_p), ("v_out", C.c_void_p), ("nh", C.c_int32), ("Ntok", C.c_int32), ("qscale", C.c_float)]


class TestConvArgs(C.Structure):
    """moge_test_conv_args (tests only): one 3x3 conv through the pieces the decoder fuses into it."""
    _fields_ = [("precision", C.c_int32), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32),
                ("relu_in", C.c_int32), ("act", C.c_int32), ("up2", C.c_int32),
                ("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("add", C.c_void_p), ("side", C.c_void_p), ("side_w", C.c_void_p),
                ("wu", C.c_void_p), ("wv", C.c_void_p), ("u0", C.c_float), ("u1", C.c_float), ("v0", C.c_float), ("v1", C.c_float),
                ("w2", C.c_void_p), ("bias2", C.c_void_p), ("y", C.c_void_p), ("dot_w", C.c_void_p), ("dot_rows", C.c_int32)]


class TestCt3Args(C.Structure):
    """moge_test_ct3_args (tests only): ConvTranspose2d + 3x3 through the fused path of the fp16 decoder."""
    _fields_ = [("precision", C.c_int32), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32), ("no_border", C.c_int32),
                ("x", C.c_void_p), ("wt", C.c_void_p), ("bt", C.c_void_p), ("w3", C.c_void_p), ("b3", C.c_void_p), ("side", C.c_void_p), ("side_w", C.c_void_p),
                ("wu", C.c_void_p), ("wv", C.c_void_p), ("u0", C.c_float), ("u1", C.c_float), ("v0", C.c_float), ("v1", C.c_float), ("y", C.c_void_p)]


class MogeError(RuntimeError):
    pass


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -m moge_amd.build` (hipcc, gfx950). "
                          "moge_amd has no CPU / PyTorch fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, i32, i64, f32p = C.c_void_p, C.c_int, C.c_int64, C.c_void_p
    sig = {
        "moge_abi_version": (C.c_int, []),
        "moge_last_error": (C.c_char_p, []),
        "moge_create": (C.c_int, [C.POINTER(MogeConfig), i32, C.POINTER(vp)]),
        "moge_create_v1": (C.c_int, [C.POINTER(MogeV1Config), i32, C.POINTER(vp)]),
        "moge_v1_forward": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, i32, C.POINTER(Outputs), vp]),
        "moge_v1_infer": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, i32, vp, i32, C.POINTER(Outputs), vp]),
        "moge_destroy": (None, [vp]),
        "moge_load_weights": (C.c_int, [vp, C.POINTER(TensorDesc), i32, vp]),
        "moge_alloc_master": (C.c_int, [vp]),
        "moge_master_blob": (C.c_int, [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]),
        "moge_master_ready": (C.c_int, [vp]),
        "moge_broadcast_weights": (C.c_int, [vp, vp, i32, vp]),
        "moge_set_precision": (C.c_int, [vp, i32, vp]),
        "moge_set_onnx_compatible_mode": (C.c_int, [vp, i32]),
        "moge_workspace_bytes": (C.c_int, [vp, i32, i32, i32, i32, i32, C.POINTER(C.c_size_t)]),
        "moge_forward": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, i32, C.POINTER(Outputs), vp]),
        "moge_infer": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, i32, vp, i32, C.POINTER(Outputs), vp]),
        "moge_postprocess": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, vp, i32, C.POINTER(Outputs), vp]),
        "moge_depth_edge_mask": (C.c_int, [vp, vp, vp, i32, i32, i32, C.c_float, vp, vp]),
        "moge_cast_f16": (C.c_int, [vp, vp, C.c_int64, vp]),
        "moge_sync": (C.c_int, [vp, vp]),
        "moge_profile_enable": (C.c_int, [vp, i32]),
        "moge_profile_read": (C.c_int, [vp, C.POINTER(Profile), i32]),
        "moge_debug_tap": (C.c_int, [vp, C.c_char_p, vp, i64, C.POINTER(i64), vp]),
        "moge_tune_set": (None, [C.c_char_p, i32]),
        "moge_test_gemm": (C.c_int, [i32, f32p, f32p, f32p, f32p, i32, i32, i32, i32, vp]),
        "moge_test_gemm_ex": (C.c_int, [C.POINTER(TestGemmArgs), vp]),
        "moge_test_layernorm": (C.c_int, [i32, f32p, f32p, f32p, f32p, i32, i32, vp]),
        "moge_test_attention": (C.c_int, [i32, f32p, f32p, f32p, f32p, i32, i32, i32, vp]),
        "moge_test_conv3x3": (C.c_int, [i32, f32p, f32p, f32p, f32p, i32, i32, i32, i32, i32, i32, vp]),
        "moge_test_conv_ex": (C.c_int, [C.POINTER(TestConvArgs), vp]),
        "moge_test_convt2x2": (C.c_int, [i32, f32p, f32p, f32p, f32p, i32, i32, i32, i32, i32, vp]),
        "moge_test_ct3": (C.c_int, [C.POINTER(TestCt3Args), vp]),
        "moge_test_preprocess": (C.c_int, [f32p, f32p, i32, i32, i32, i32, i32, vp]),
        "moge_test_resize_bicubic_aa": (C.c_int, [f32p, f32p, i32, i32, i32, i32, i32, vp]),
        "moge_test_groupnorm_relu": (C.c_int, [i32, f32p, f32p, f32p, f32p, i32, i32, i32, i32, i32, vp]),
        "moge_test_norm_act": (C.c_int, [i32, f32p, f32p, f32p, f32p, i32, i32, i32, i32, i32, i32, i32, vp]),
        "moge_test_posembed": (C.c_int, [f32p, f32p, i32, i32, i32, vp]),
        "moge_test_recover": (C.c_int, [f32p, vp, f32p, i32, i32, i32, f32p, f32p, vp, vp]),
        "moge_align_l1": (C.c_int, [f32p, f32p, f32p, i32, i32, C.c_float, f32p, f32p, vp, vp]),
        "moge_align_l1_anchored": (C.c_int, [f32p, f32p, f32p, i32, i32, i32, vp, vp, i32, C.c_float, f32p, f32p, vp, vp]),
        "moge_align_select": (C.c_int, [f32p, vp, i32, i32, f32p, vp, vp]),
        "moge_align_lstsq": (C.c_int, [f32p, f32p, f32p, i32, i32, f32p, f32p, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    if lib.moge_abi_version() != ABI_VERSION:
        raise ImportError("libmoge_hip.so ABI version mismatch")
    return lib


lib = _load()
EXPORTS = ["moge_abi_version", "moge_last_error", "moge_create", "moge_create_v1", "moge_v1_forward", "moge_v1_infer", "moge_destroy", "moge_load_weights", "moge_alloc_master",
           "moge_master_blob", "moge_master_ready", "moge_broadcast_weights", "moge_set_precision", "moge_set_onnx_compatible_mode", "moge_workspace_bytes", "moge_forward", "moge_infer",
           "moge_postprocess", "moge_depth_edge_mask", "moge_cast_f16", "moge_sync", "moge_profile_enable", "moge_profile_read", "moge_debug_tap", "moge_tune_set", "moge_test_gemm",
           "moge_test_gemm_ex", "moge_test_layernorm", "moge_test_attention", "moge_test_conv3x3", "moge_test_conv_ex", "moge_test_convt2x2", "moge_test_ct3", "moge_test_preprocess",
           "moge_test_resize_bicubic_aa", "moge_test_groupnorm_relu", "moge_test_norm_act", "moge_test_posembed", "moge_test_recover",
           "moge_align_l1", "moge_align_l1_anchored", "moge_align_select", "moge_align_lstsq"]


def check(code: int) -> None:
    if code == 0:
        return
    msg = (lib.moge_last_error() or b"").decode(errors="replace")
    if code == ERR_NONFINITE:
        raise ValueError(msg or "Residuals are not finite in the initial point.")      # what scipy raises in the reference
    raise MogeError(f"libmoge_hip error {code}: {msg}")


def tune(key: str, value: int) -> None:
    """A/B switch of the library (tests / tools): same keys as the MOGE_<KEY> environment variables."""
    lib.moge_tune_set(key.encode(), int(value))


def stream_ptr(device=None) -> int:
    return int(torch.cuda.current_stream(device).cuda_stream)


class DevView:
    """Zero-copy torch view of a raw device buffer (via __cuda_array_interface__)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2, "strides": None}

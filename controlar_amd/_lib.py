"""ctypes binding of libcontrolar_hip.so (include/controlar_hip.h).

This is the stub a reference maintainer would add next to autoregressive/models/generate.py
(see INTEGRATION.md).  There is NO fallback: if the shared library is missing or does not
export every declared symbol the import raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libcontrolar_hip.so")
# The same sources compiled with -DCAR_DEV_KNOBS (csrc/build.sh): the A/B and profiling switches behind CAR_* environment variables exist ONLY there.
# Loaded when CONTROLAR_DEV_LIB=1 is in the environment or on request (`load(dev=True)`, `Engine(..., dev=True)`: tools/, the schedule-invariance tests).
DEV_LIB_PATH = os.path.join(_HERE, "csrc", "libcontrolar_hip_dev.so")

CAR_ABI_VERSION = 2
CAR_F32, CAR_BF16 = 0, 1
CAR_DT_F32, CAR_DT_BF16, CAR_DT_I32, CAR_DT_I64, CAR_DT_U8 = 0, 1, 2, 3, 4
CAR_RESIZE_NEAREST, CAR_RESIZE_BICUBIC_AC = 0, 1


class CarConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("mode", C.c_int32),
        ("dim", C.c_int32), ("n_layer", C.c_int32), ("n_head", C.c_int32), ("ffn_hidden", C.c_int32), ("vocab_size", C.c_int32),
        ("cls_token_num", C.c_int32), ("block_size", C.c_int32), ("caption_dim", C.c_int32),
        ("norm_eps", C.c_float), ("rope_base", C.c_float),
        ("vit_hidden", C.c_int32), ("vit_layers", C.c_int32), ("vit_heads", C.c_int32), ("vit_mlp", C.c_int32),
        ("vit_patch", C.c_int32), ("vit_pos_grid", C.c_int32), ("vit_ln_eps", C.c_float), ("resize_mode", C.c_int32),
        ("codebook_size", C.c_int32), ("codebook_dim", C.c_int32), ("z_channels", C.c_int32), ("vq_ch", C.c_int32),
        ("vq_num_res_blocks", C.c_int32), ("vq_n_mult", C.c_int32), ("vq_ch_mult", C.c_int32 * 8), ("gn_eps", C.c_float),
        ("vit_variant", C.c_int32), ("model_type", C.c_int32), ("num_classes", C.c_int32), ("stream_priority", C.c_int32), ("decode_weight_fp8", C.c_int32), ("kv_cache_fp8", C.c_int32), ("reserved", C.c_int32 * 2),
    ]


class CarT5Config(C.Structure):
    _fields_ = [
        ("vocab_size", C.c_int32), ("d_model", C.c_int32), ("d_kv", C.c_int32), ("num_heads", C.c_int32), ("d_ff", C.c_int32),
        ("num_layers", C.c_int32), ("rel_buckets", C.c_int32), ("rel_max_distance", C.c_int32), ("ln_eps", C.c_float),
        ("reserved", C.c_int32 * 7),
    ]


class CarSampling(C.Structure):
    _fields_ = [
        ("cfg_scale", C.c_float), ("cfg_interval", C.c_int32), ("temperature", C.c_float), ("top_k", C.c_int32),
        ("top_p", C.c_float), ("sample_logits", C.c_int32), ("seed", C.c_uint64), ("control_strength", C.c_float),
        ("first_valid_hint", C.c_int32), ("reserved", C.c_int32 * 3),
    ]


class CarStats(C.Structure):
    _fields_ = [
        ("decode_ms", C.c_double), ("prefill_ms", C.c_double), ("decode_steps", C.c_int64), ("decode_algo_bytes", C.c_int64),
        ("decode_kernels_per_step", C.c_int32), ("graph_used", C.c_int32), ("dev_knobs_active", C.c_int32), ("reserved", C.c_int32 * 5),
    ]


# name -> (restype, argtypes): exactly the entry points declared in include/controlar_hip.h
SYMBOLS = {
    "car_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(CarConfig)]),
    "car_destroy": (None, [C.c_void_p]),
    "car_last_error": (C.c_char_p, [C.c_void_p]),
    "car_abi_version": (C.c_int, []),
    "car_build_id": (C.c_char_p, []),
    "car_load_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.c_int32]),
    "car_finalize_weights": (C.c_int, [C.c_void_p]),
    "car_check_errors": (C.c_int, [C.c_void_p]),
    "car_export_packed": (C.c_int, [C.c_void_p, C.c_char_p]),
    "car_import_packed": (C.c_int, [C.c_void_p, C.c_char_p]),
    "car_t5_configure": (C.c_int, [C.c_void_p, C.POINTER(CarT5Config)]),
    "car_t5_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "car_encode_control": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "car_generate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                               C.POINTER(CarSampling), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "car_generate_c2i": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(CarSampling), C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    "car_sample_logits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(CarSampling), C.c_int32, C.c_void_p, C.c_void_p]),
    "car_canny": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "car_vq_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "car_vq_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "car_get_stats": (C.c_int, [C.c_void_p, C.POINTER(CarStats)]),
    "car_debug_pack_decode_weight": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "car_debug_f32_to_e4m3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "car_debug_control_tokens": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]),
}

_lib = None
_lib_dev = None


def load(dev=None) -> C.CDLL:
    """Loads the HIP library; raises (never falls back) when it is absent or incomplete.  `dev`: the development build with the CAR_* switches
    (default: only when CONTROLAR_DEV_LIB=1)."""
    global _lib, _lib_dev
    if dev is None:
        dev = os.environ.get("CONTROLAR_DEV_LIB") == "1"
    if dev and _lib_dev is not None:
        return _lib_dev
    if not dev and _lib is not None:
        return _lib
    path = DEV_LIB_PATH if dev else LIB_PATH
    if not os.path.exists(path):
        raise ImportError(f"{path} not found — build it with `python __graft_entry__.py build` "
                          "(controlar_amd has no CPU or PyTorch fallback)")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype, fn.argtypes = res, args
    if lib.car_abi_version() != CAR_ABI_VERSION:
        raise ImportError("libcontrolar_hip.so ABI version mismatch")
    if dev:
        _lib_dev = lib
        return lib
    _lib = lib
    return lib

"""ctypes binding of libscenerf_hip.so (the C ABI declared in include/scenerf_hip.h).

There is NO fallback: if the shared library is missing or an entry point fails, a RuntimeError is
raised.  The product never routes through the CPU oracle or eager PyTorch for the hot path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

# torch first: libscenerf_hip.so must bind to the HIP runtime torch has already loaded (one runtime per process;
# loading /opt/rocm's copy before torch's bundled one leaves the library without a visible device).
import torch  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
_TAG = os.environ.get("SRF_LIB_TAG", "")      # development only: a variant library built with SRF_LIB_TAG / SRF_EXTRA_FLAGS (build.py)
LIB_PATH = os.path.join(HERE, "csrc", "libscenerf_hip%s.so" % ("_" + _TAG if _TAG else ""))

N_SCALES = 5
D_LATENT = 2480
D_HIDDEN = 512
D_XENC = 48
TILE_ROWS = 128
W_STREAM_BLOCKS = (3 * D_XENC + D_LATENT) // 16 + 2 * ((512 + D_LATENT) // 16) + 10 * (512 // 16)   # scenerf_hip.h
ABI_VERSION = 10

vp = C.c_void_p


class Cfg(C.Structure):
    _fields_ = [
        ("n_pts_uni", C.c_int32), ("n_gaussians", C.c_int32), ("n_pts_per_gaussian", C.c_int32),
        ("n_samples", C.c_int32), ("sphere_W", C.c_int32), ("sphere_H", C.c_int32),
        ("max_sample_depth", C.c_float), ("uni_step", C.c_float), ("base_std", C.c_float),
        ("som_sigma", C.c_float), ("gauss_floor", C.c_float), ("kl_std_floor", C.c_float),
        ("v_min", C.c_float), ("v_fov", C.c_float), ("h_min", C.c_float), ("h_fov", C.c_float),
        ("map_C", C.c_int32 * N_SCALES), ("map_H", C.c_int32 * N_SCALES), ("map_W", C.c_int32 * N_SCALES),
        ("div_H", C.c_int32 * N_SCALES), ("div_W", C.c_int32 * N_SCALES),
        ("precision", C.c_int32),
        ("map_chw", C.c_int32 * N_SCALES),
        ("fused_min_rows", C.c_int32), ("fwd_kernel", C.c_int32), ("flags", C.c_uint32),
    ]


class MlpWeights(C.Structure):
    _fields_ = [
        ("d_out", C.c_int32),
        ("w_in", vp), ("b_in", vp),
        ("w_h", vp * 4), ("b_h", vp * 4),
        ("w_fc0", vp * 3), ("b_fc0", vp * 3),
        ("w_out", vp), ("b_out", vp),
        ("w_fc0_t", vp * 3), ("w_fc1_t", vp * 3), ("w_z_t", vp * N_SCALES),
        ("w_stream", vp),
        ("clear", vp), ("clear_floats", C.c_int64),
    ]


RESNETFC_MAX_BLOCKS = 8


class ResnetFCNet(C.Structure):
    """scenerf_resnetfc (scenerf_hip.h): a ResnetFC of any block count / width (fp32 GEMM path)."""
    _fields_ = [("n_blocks", C.c_int32), ("d_hidden", C.c_int32), ("d_out_pad", C.c_int32),
                ("w_in", vp), ("b_in", vp),
                ("w_z", vp * RESNETFC_MAX_BLOCKS), ("b_z", vp * RESNETFC_MAX_BLOCKS),
                ("w_fc0", vp * RESNETFC_MAX_BLOCKS), ("b_fc0", vp * RESNETFC_MAX_BLOCKS),
                ("w_fc1", vp * RESNETFC_MAX_BLOCKS), ("b_fc1", vp * RESNETFC_MAX_BLOCKS),
                ("w_out", vp), ("b_out", vp)]


class ResnetFCNetT(C.Structure):
    """scenerf_resnetfc_t: the transposed operands of the generic net's input-gradient GEMMs."""
    _fields_ = [("w_fc0_t", vp * RESNETFC_MAX_BLOCKS), ("w_fc1_t", vp * RESNETFC_MAX_BLOCKS), ("w_out_t", vp), ("w_z_t", vp * N_SCALES)]


class ResnetFCActs(C.Structure):
    _fields_ = [("hz", vp * RESNETFC_MAX_BLOCKS), ("n", vp * RESNETFC_MAX_BLOCKS), ("h_fin", vp)]


class ResnetFCGrads(C.Structure):
    _fields_ = [("w_in", vp), ("b_in", vp), ("w_z", vp),
                ("w_fc0", vp * RESNETFC_MAX_BLOCKS), ("b_fc0", vp * RESNETFC_MAX_BLOCKS),
                ("w_fc1", vp * RESNETFC_MAX_BLOCKS), ("b_fc1", vp * RESNETFC_MAX_BLOCKS),
                ("w_out", vp), ("b_out", vp)]


class MlpParams(C.Structure):
    _fields_ = [
        ("d_out", C.c_int32),
        ("lin_in_w", vp), ("lin_in_b", vp), ("lin_out_w", vp), ("lin_out_b", vp),
        ("fc0_w", vp * 3), ("fc0_b", vp * 3), ("fc1_w", vp * 3), ("fc1_b", vp * 3), ("linz_w", vp * 3), ("linz_b", vp * 3),
    ]


class MlpGrads(C.Structure):
    _fields_ = [
        ("w_in", vp), ("b_in", vp),
        ("w_fc0", vp * 3), ("b_fc0", vp * 3),
        ("w_fc1", vp * 3), ("b_fc1", vp * 3),
        ("w_z", vp), ("b_z", vp), ("w_out", vp), ("b_out", vp), ("w_in_dense", vp),
    ]


class MlpActs(C.Structure):
    _fields_ = [("H", vp * 4), ("Nn", vp * 3), ("h0pre", vp), ("logits", vp), ("sign_bits", vp), ("x3_ready", C.c_int32)]


class AdamWTensor(C.Structure):
    _fields_ = [("p", vp), ("g", vp), ("m", vp), ("v", vp), ("numel", C.c_int64), ("g_cols", C.c_int32), ("g_ld", C.c_int32),
                ("step", C.c_int64)]


class ProfRec(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int32), ("total_ms", C.c_float),
                ("flops", C.c_double), ("bytes", C.c_double)]


i32 = C.c_int
_PROTOS = {
    "scenerf_hip_abi_version": (C.c_int, []),
    "scenerf_hip_last_error": (C.c_char_p, []),
    "scenerf_hip_clear_last_error": (C.c_int, []),
    "scenerf_hip_stream_capture_id": (C.c_int, [vp, C.POINTER(C.c_ulonglong)]),
    "scenerf_hip_stream_create_lowest_priority": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "scenerf_hip_stream_destroy": (C.c_int, [vp]),
    "scenerf_hip_depth_errors": (C.c_int, [vp, vp, vp, C.c_int64, C.c_float, C.c_float, vp, vp]),
    "scenerf_hip_prepare": (C.c_int, [C.POINTER(Cfg), vp]),
    "scenerf_hip_maps_chw_to_hwc": (C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
    "scenerf_hip_grads_hwc_to_chw": (C.c_int, [vp, vp, i32, i32, i32, vp]),
    "scenerf_hip_fill_zero": (C.c_int, [vp, C.c_int64, i32, vp]),
    "scenerf_hip_ray_setup": (C.c_int, [C.POINTER(Cfg), vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp]),
    "scenerf_hip_encode_points": (C.c_int, [C.POINTER(Cfg), vp, i32, i32, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp]),
    "scenerf_hip_gather_features": (C.c_int, [C.POINTER(Cfg), C.POINTER(vp * N_SCALES), vp, i32, vp, vp, vp, vp, vp]),
    "scenerf_hip_mlp_pack": (C.c_int, [C.POINTER(Cfg), C.POINTER(MlpParams), C.POINTER(MlpWeights), vp]),
    "scenerf_hip_mlp_forward": (C.c_int, [C.POINTER(Cfg), C.POINTER(MlpWeights), vp, vp, vp, i32, C.POINTER(MlpActs), vp]),
    "scenerf_hip_mlp_backward": (C.c_int, [C.POINTER(Cfg), C.POINTER(MlpWeights), C.POINTER(MlpGrads), vp, vp, vp, vp, vp,
                                           i32, C.POINTER(MlpActs), vp, vp, vp, C.POINTER(vp * N_SCALES), vp]),
    "scenerf_hip_tsdf_integrate": (C.c_int, [vp, vp, vp, C.POINTER(C.c_int32 * 3), C.POINTER(C.c_float * 3), C.c_double,
                                             C.POINTER(C.c_float * 9), C.POINTER(C.c_float * 16), C.POINTER(C.c_double * 16), vp, vp,
                                             i32, i32, C.c_float, C.c_float, i32, vp]),
    "scenerf_hip_ray_tail_forward": (C.c_int, [C.POINTER(Cfg)] + [vp] * 5 + [i32] + [vp] * 13 + [vp]),
    "scenerf_hip_ray_tail_backward": (C.c_int, [C.POINTER(Cfg)] + [vp] * 3 + [i32] + [vp] * 24 + [vp]),
    "scenerf_hip_adamw_step": (C.c_int, [i32, C.POINTER(AdamWTensor), C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, vp]),
    "scenerf_hip_adamw_step_dev": (C.c_int, [i32, C.POINTER(AdamWTensor), vp, C.c_float, C.c_float, C.c_float, C.c_float, vp]),
    "scenerf_hip_sphere_map_build": (C.c_int, [vp, vp, C.c_int64, i32, i32, i32, vp, vp, vp]),
    "scenerf_hip_sphere_resample_forward": (C.c_int, [vp, C.c_int64, i32, i32, vp, i32, i32, vp, vp]),
    "scenerf_hip_sphere_resample_backward": (C.c_int, [vp, C.c_int64, i32, i32, vp, vp, i32, i32, vp, vp]),
    "scenerf_hip_sphere_resample_forward_nhwc": (C.c_int, [vp, C.c_int64, i32, i32, i32, vp, i32, i32, vp, vp]),
    "scenerf_hip_sphere_resample_backward_nhwc": (C.c_int, [vp, C.c_int64, i32, i32, i32, vp, vp, i32, i32, vp, vp]),
    "scenerf_hip_loss_side_forward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "scenerf_hip_loss_side_backward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp]),
    "scenerf_hip_source_loss_forward": (C.c_int, [vp] * 7 + [i32] + [vp] * 4 + [C.c_float] + [vp] * 3 + [i32] * 3 + [C.c_float] * 3 + [vp] * 7 + [vp]),
    "scenerf_hip_source_loss_backward": (C.c_int, [vp] * 9 + [i32, i32] + [C.c_float] * 3 + [vp] * 4 + [vp]),
    "scenerf_hip_mlp_feature_grads": (C.c_int, [C.POINTER(Cfg), C.POINTER(MlpWeights), vp, vp, vp, i32, vp,
                                                C.POINTER(vp * N_SCALES), vp]),
    "scenerf_hip_gaussian_sample_sort": (C.c_int, [C.POINTER(Cfg), vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp]),
    "scenerf_hip_composite_forward": (C.c_int, [vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "scenerf_hip_composite_backward": (C.c_int, [vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "scenerf_hip_raysom_forward": (C.c_int, [C.POINTER(Cfg), vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp]),
    "scenerf_hip_sampler_backward": (C.c_int, [C.POINTER(Cfg), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp]),
    "scenerf_hip_pixels_to_sphere": (C.c_int, [vp, vp, C.c_float, C.c_float, C.c_float, C.c_float, i32, i32, C.c_int64, vp, vp, vp]),
    "scenerf_hip_test_acos_atan2": (C.c_int, [vp, vp, C.c_int64, vp, vp, vp]),
    "scenerf_hip_resnetfc_forward": (C.c_int, [C.POINTER(Cfg), C.POINTER(ResnetFCNet), vp, vp, vp, i32, vp, vp, vp, vp, vp]),
    "scenerf_hip_resnetfc_forward_train": (C.c_int, [C.POINTER(Cfg), C.POINTER(ResnetFCNet), vp, vp, vp, i32, C.POINTER(ResnetFCActs), vp, vp, vp]),
    "scenerf_hip_resnetfc_backward": (C.c_int, [C.POINTER(Cfg), C.POINTER(ResnetFCNet), C.POINTER(ResnetFCNetT), C.POINTER(ResnetFCGrads),
                                                vp, vp, vp, vp, vp, i32, C.POINTER(ResnetFCActs), vp, vp, vp, vp, vp, vp]),
    "scenerf_hip_test_gemm_nt": (C.c_int, [i32, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]),
    "scenerf_hip_test_chunk_table": (C.c_int, [C.POINTER(Cfg), i32, vp, i32]),
    "scenerf_hip_test_set_tuning": (C.c_int, [i32, i32]),
    "scenerf_hip_test_gemm_tn": (C.c_int, [i32, vp, vp, i32, i32, i32, i32, vp, vp, vp]),
    "scenerf_hip_profile_enable": (C.c_int, [i32]),
    "scenerf_hip_profile_collect": (C.c_int, [C.POINTER(ProfRec), i32]),
}
EXPORTED_SYMBOLS = tuple(_PROTOS)

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """dlopen the in-tree library and type its entry points.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libscenerf_hip.so is not built (%s). Run `python -m scenerf_amd.build` (needs hipcc). "
            "There is no CPU / eager fallback for the SceneRF hot path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    v = lib.scenerf_hip_abi_version()
    if v != ABI_VERSION:
        raise RuntimeError("libscenerf_hip.so ABI %d != binding ABI %d: rebuild" % (v, ABI_VERSION))
    if os.environ.get("SRF_TUNING"):   # development only (same-box A/B): "warm_wide,dfeat_delay_us" (scenerf_hip_test_set_tuning; -1 = default)
        lib.scenerf_hip_test_set_tuning(*[int(x) for x in os.environ["SRF_TUNING"].split(",")])
    _lib = lib
    return lib


FUSED_MIN_ROWS_DEFAULT = 4096    # SCENERF_FUSED_MIN_ROWS_DEFAULT
FLAG_NO_FUSED_BWD, FLAG_NO_WGRAD_TR, FLAG_DFEAT_PER_SCALE, FLAG_WGRAD_OVERLAP, FLAG_WIDE_BWD, FLAG_WIDE_ANY_M, FLAG_DFEAT_GEMM = 1, 2, 4, 8, 16, 32, 64   # SCENERF_FLAG_*
FLAG_UNIFORM_ONLY = 128
FLAG_WIDE_BWD_STAGED = 256
FLAG_PACK_FORWARD, FLAG_PACK_REST = 512, 1024
FLAG_BWD_CHAIN_ONLY, FLAG_BWD_GRADS_ONLY = 2048, 4096
ADAMW_SCRATCH = 65        # SCENERF_ADAMW_SCRATCH: words behind [lr, t] in scenerf_hip_adamw_step_dev's hyper
WIN_LD = 256            # SCENERF_WIN_LD: row stride of scenerf_mlp_grads.w_in


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().scenerf_hip_last_error()
        raise RuntimeError("%s failed (code %d): %s" % (what, code, (msg or b"").decode("utf-8", "replace")))


def ptr(t) -> Optional[int]:
    """device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def profile_collect(cap: int = 256):
    lib = load()
    arr = (ProfRec * cap)()
    n = lib.scenerf_hip_profile_collect(arr, cap)
    return [dict(name=arr[i].name.decode(), launches=arr[i].launches, total_ms=arr[i].total_ms,
                 flops=arr[i].flops, bytes=arr[i].bytes) for i in range(n)]

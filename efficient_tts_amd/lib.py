"""ctypes binding of libefts_hip.so (the C ABI declared in include/efts_abi.h).

The library is the product: there is NO CPU / PyTorch fallback.  If the shared object is
missing or a call fails, an exception is raised with ``efts_last_error()``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EFTS_LIB", os.path.join(HERE, "libefts_hip.so"))   # EFTS_LIB: kernel experiments only

ABI_VERSION = 601         # EFTS_ABI_VERSION of the include/efts_abi.h this binding mirrors; load() refuses any other library
RC_PLAN_INTS = 42
GAP = 2
GUARD_LO = 8
GUARD_HI = 144
TILE_M = 128
ACT_NONE, ACT_LEAKY, ACT_RELU, ACT_TANH = 0, 1, 2, 3
ACT_BWD_BIAS_PARTS = 16
# EFTS_ACTFN_* (include/efts_abi.h): torch.nn module name -> (id, names of the module's scalar parameters in p0 / p1 order with torch's defaults)
ACTFN = {
    "Identity": (0, ()), "ReLU": (1, ()), "LeakyReLU": (2, (("negative_slope", 0.01),)), "ELU": (3, (("alpha", 1.0),)),
    "CELU": (4, (("alpha", 1.0),)), "SELU": (5, ()), "GELU": (6, ()), "SiLU": (8, ()), "Mish": (9, ()), "Tanh": (10, ()), "Sigmoid": (11, ()),
    "Softplus": (12, (("beta", 1.0), ("threshold", 20.0))), "Hardtanh": (13, (("min_val", -1.0), ("max_val", 1.0))), "ReLU6": (13, ()),
    "Hardswish": (14, ()), "Hardsigmoid": (15, ()), "Softsign": (16, ()), "Tanhshrink": (17, ()), "LogSigmoid": (18, ()),
}


def actfn(name: str, params: dict):
    """(EFTS_ACTFN id, p0, p1) of torch.nn.<name>(**params), or None when the library has no form of it (learnable or non-pointwise
    modules: PReLU, RReLU, Softmax, GLU, ...)"""
    if name not in ACTFN:
        return None
    aid, keys = ACTFN[name]
    extra = set(params) - {k for k, _ in keys} - {"inplace"} - ({"approximate"} if name == "GELU" else set())
    if extra:
        return None
    if name == "GELU":
        approx = params.get("approximate", "none")
        if approx not in ("none", "tanh"):
            return None
        return (7 if approx == "tanh" else 6), 0.0, 0.0
    if name == "ReLU6":
        return aid, 0.0, 6.0
    vals = [float(params.get(k, d)) for k, d in keys] + [0.0, 0.0]
    return aid, vals[0], vals[1]
TILING_AUTO, TILING_GENERIC, TILING_WIDE, TILING_NARROW, TILING_RESIDENT, TILING_SMALLM = 0, 1, 2, 3, 4, 5

i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class GemmArgs(C.Structure):
    """mirror of `struct efts_gemm_args` (include/efts_abi.h)"""
    _fields_ = [
        ("a", vp), ("lda", i64), ("a_batch_stride", i64),
        ("b", vp), ("ldb", i64), ("b_tap_stride", i64), ("b_batch_stride", i64),
        ("split", i32), ("taps", i32), ("m", i32), ("n", i32), ("nchunk", i32), ("batch", i32),
        ("alpha", f32), ("act", i32), ("slope", f32),
        ("bias", vp), ("resid", vp), ("ldr", i64), ("resid_batch_stride", i64),
        ("rowmask", vp), ("rowmask_batch_stride", i64),
        ("out_f32", vp), ("ldo", i64), ("out_batch_stride", i64),
        ("out_bf16", vp), ("ldob", i64), ("outb_batch_stride", i64),
        ("out_split", i32), ("batch2", i32),
        ("a_batch2_stride", i64), ("b_batch2_stride", i64), ("out_batch2_stride", i64),
        ("dilation", i32), ("plane_act", i32), ("plane_slope", f32),
        ("out_bf16_lo", vp), ("tiling", i32), ("sign_mask", vp), ("soft_index", vp), ("key_len", vp), ("query_len", vp), ("drop_p", f32), ("drop_seed", C.c_uint32),
        ("sqerr_target", vp), ("ld_target", i64), ("target_batch_stride", i64), ("sqerr_part", vp),
    ]


class FrameLinearArgs(C.Structure):
    """mirror of efts_frame_linear_args (include/efts_abi.h)"""
    _fields_ = [("x", vp), ("w", vp), ("ldw", i64), ("split", i32), ("bias", vp), ("act", i32), ("slope", f32),
                ("B", i32), ("T", i32), ("Tp", i32), ("cin", i32), ("n", i32), ("y_f32", vp), ("ldo", i64), ("y", vp), ("y_lo", vp),
                ("ldy", i64), ("y_split", i32), ("max_workgroups", i32)]


class ExpandArgs(C.Structure):
    """mirror of efts_expand_args (include/efts_abi.h)"""
    _fields_ = [("e", vp), ("text_len", vp), ("mel_len", vp), ("sigma", f32), ("v", vp), ("ldv", i64),
                ("B", i32), ("T1", i32), ("T1p", i32), ("T2", i32), ("T2p", i32), ("n", i32),
                ("alpha_out", vp), ("y_f32", vp), ("ldo", i64), ("y", vp), ("y_lo", vp), ("ldy", i64), ("y_split", i32)]


class ResConv5Args(C.Structure):
    """mirror of `struct efts_resconv5_args` (include/efts_abi.h)"""
    _fields_ = [
        ("x", vp), ("x_lo", vp), ("ldx", i64), ("x_f32", vp), ("ldr", i64),
        ("w", vp), ("ldw", i64), ("w_tap_stride", i64),
        ("split", i32), ("m", i32), ("n", i32), ("nchunk", i32),
        ("bias", vp), ("slope", f32), ("rowmask", vp),
        ("y_f32", vp), ("ldo", i64), ("y", vp), ("y_lo", vp), ("ldy", i64), ("y_split", i32),
        ("plan", C.POINTER(i32)), ("taps", i32), ("no_residual", i32), ("sign_bits", vp),
        ("act_bwd_sign", vp), ("act_bwd_bias_part", vp), ("act_bwd_bias_rows", i32), ("act_bwd_slope", f32), ("kernel", i32),
    ]


class WgradItem(C.Structure):
    """mirror of `struct efts_wgrad_item` (include/efts_abi.h)"""
    _fields_ = [("dz_plane", vp), ("ldz", i64), ("x_plane", vp), ("ldx", i64), ("v", vp), ("g", vp), ("dw_or_dv", vp), ("dg", vp),
                ("bias_part", vp), ("dbias", vp), ("nparts", i32), ("reserved", i32)]


WGRAD_MAX_ITEMS = 8

_SIGS = {
    "efts_version": (i32, []),
    "efts_last_error": (C.c_char_p, []),
    "efts_device_check": (i32, []),
    "efts_gemm": (i32, [C.POINTER(GemmArgs), vp]),
    "efts_resconv5": (i32, [C.POINTER(ResConv5Args), vp]),
    "efts_resconv5_multi": (i32, [C.POINTER(ResConv5Args), i32, vp]),
    "efts_resconv5_plan": (i32, [i32, i32, i32, C.POINTER(i32), i32]),
    "efts_resconv5_bias_rows": (i32, [i32, i32]),
    "efts_pack_weight": (i32, [vp, vp, vp, vp, i64, i32, i32, i32, i32, vp]),
    "efts_row_masks": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "efts_row_masks_pair": (i32, [vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "efts_embed": (i32, [vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, vp]),
    "efts_embed_conv": (i32, [vp, vp, vp, vp, vp, f32, vp, vp, i64, i32, i32, i32, i32, i32, i32, i32, vp]),
    "efts_pack_rows": (i32, [vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, vp]),
    "efts_attn_soft_index": (i32, [vp, i64, vp, vp, vp, vp, i32, i32, i32, vp]),
    "efts_imv_scan": (i32, [vp, vp, vp, vp, i32, i32, vp]),
    "efts_aligned_positions": (i32, [vp, vp, vp, f32, f32, vp, vp, i32, i32, i32, vp]),
    "efts_duration_target": (i32, [vp, vp, vp, f32, i32, vp, i32, i32, vp]),
    "efts_reconst_alpha": (i32, [vp, vp, vp, f32, vp, vp, i64, i32, i32, i32, i32, vp]),
    "efts_pack_vt": (i32, [vp, i64, vp, i64, i32, i32, i32, i32, vp]),
    "efts_cumsum_rows": (i32, [vp, vp, i32, i32, vp]),
    "efts_imv_align": (i32, [vp, vp, vp, f32, f32, i32, vp, vp, vp, i32, i32, i32, vp]),
    "efts_expand": (i32, [C.POINTER(ExpandArgs), vp]),
    "efts_duration_positions": (i32, [vp, i64, vp, f32, i32, vp, vp, i32, i32, vp]),
    "efts_bf16_round": (i32, [vp, vp, i64, i32, vp]),
    "efts_layernorm_rows": (i32, [vp, vp, vp, f32, vp, vp, vp, i64, i32, i32, i32, f32, C.c_uint32, vp, vp]),
    "efts_layernorm_dot": (i32, [vp, vp, vp, f32, vp, vp, vp, i32, f32, vp, i32, i32, f32, C.c_uint32, vp, vp]),
    "efts_losses_workspace_bytes": (C.c_size_t, []),
    "efts_masked_losses": (i32, [vp, i64, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "efts_losses_from_parts": (i32, [vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    # training step
    "efts_pack_weight_t": (i32, [vp, vp, i64, i32, i32, i32, i32, vp]),
    "efts_pack_weights_grouped": (i32, [vp, i32, vp, i64, i64, i32, i32, i32, i32, i32, vp]),
    "efts_loss_bwd": (i32, [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, vp, i32, i32, i32, i32, i32, i32, vp]),
    "efts_act_bwd": (i32, [vp, vp, vp, vp, f32, i32, vp, vp, i64, i32, vp, i32, i32, vp]),
    "efts_act_bwd_dropout": (i32, [vp, vp, vp, vp, f32, i32, vp, vp, i64, i32, vp, i32, i32, f32, C.c_uint32, vp]),
    "efts_act_apply": (i32, [vp, vp, vp, i32, f32, f32, vp, vp, i64, i32, i32, i32, f32, C.c_uint32, vp]),
    "efts_act_grad": (i32, [vp, vp, vp, i32, f32, f32, vp, vp, i64, i32, vp, i32, i32, f32, C.c_uint32, vp]),
    "efts_pack_t": (i32, [vp, i64, vp, i64, i64, i32, i32, i32, i32, i32, i32, vp]),
    "efts_wgrad_reduce": (i32, [vp, i32, vp, vp, vp, vp, i32, i32, i32, vp]),
    "efts_wgrad_grouped_part_bytes": (i64, [i32, i32, i32, i32, i32, i32, i32]),
    "efts_wgrad_tn_grouped": (i32, [C.POINTER(WgradItem), i32, vp, i32, i32, i32, i32, i32, i32, vp]),
    "efts_wgrad_reduce_grouped": (i32, [C.POINTER(WgradItem), i32, vp, i32, i32, i32, i32, i32, i32, vp]),
    "efts_layernorm_bwd": (i32, [vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, i64, i32, vp, vp, vp, vp, vp, i32, i32, f32, C.c_uint32, vp, vp]),
    "efts_alpha_bwd": (i32, [vp, vp, vp, vp, vp, f32, vp, vp, i32, i32, i32, vp]),
    "efts_e_bwd": (i32, [vp, vp, vp, vp, vp, f32, vp, vp, i32, i32, i32, vp]),
    "efts_imv_bwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "efts_attn_bwd": (i32, [vp, i64, vp, vp, vp, vp, vp, i64, vp, i64, i32, i32, i32, i32, vp]),
    "efts_embed_bwd": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "efts_frame_linear": (i32, [C.POINTER(FrameLinearArgs), vp]),
    "efts_sumsq_workspace_bytes": (C.c_size_t, []),
    "efts_sumsq": (i32, [vp, i64, vp, vp, vp]),
    "efts_scale_unless_one": (i32, [vp, i64, vp, vp]),
    "efts_adam_amsgrad": (i32, [vp, vp, vp, vp, vp, i64, vp, f32, f32, f32, f32, f32, f32, f32, i32, vp]),
    "efts_adam_hyper": (i32, [f32, f32, f32, i32, C.POINTER(f32)]),
    "efts_adam_amsgrad_dev": (i32, [vp, vp, vp, vp, vp, i64, vp, f32, f32, vp, f32, f32, f32, f32, vp]),
    "efts_store_words": (i32, [vp, C.POINTER(C.c_uint32), i32, vp]),
    # log-mel front-end
    "efts_frame_pack": (i32, [vp, i64, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, vp]),
    "efts_logmel": (i32, [vp, i64, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "efts_frame_pack_dit": (i32, [vp, i64, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, i32, vp]),
    "efts_logmel_dit": (i32, [vp, i64, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "efts_logmel_fft": (i32, [vp, i64, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "efts_logmel_fft_pcm16": (i32, [vp, i64, f32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    # vocoder
    "efts_mean_act_rows": (i32, [vp, vp, vp, i64, f32, f32, vp, i64, vp, i64, i32, i32, i32, vp]),
}

_lib: Optional[C.CDLL] = None


class EftsError(RuntimeError):
    pass


def load() -> C.CDLL:
    """dlopen libefts_hip.so and bind every symbol of include/efts_abi.h (no GPU needed)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EftsError(
            f"{LIB_PATH} is missing: the HIP extension is the product path and has no fallback. "
            "Build it with `python -m efficient_tts_amd.build` (needs hipcc, gfx950).")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    have = lib.efts_version()
    if have != ABI_VERSION:
        raise EftsError(f"{LIB_PATH} reports ABI revision {have}, this binding mirrors revision {ABI_VERSION} of include/efts_abi.h: "
                        "argument blocks would be misread.  Rebuild with `python -m efficient_tts_amd.build --force`.")
    _lib = lib
    return lib


def exported_symbols():
    return list(_SIGS)


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().efts_last_error().decode(errors="replace")
        if rc in (-1, -2, -3):
            raise ValueError(f"{what}: {msg} (code {rc})")
        raise EftsError(f"{what}: {msg} (code {rc})")


_device_ok = False


def require_device() -> None:
    """Fail loudly unless a gfx950 device is current (also opts kernels into large LDS)."""
    global _device_ok
    if not _device_ok:
        check(load().efts_device_check(), "efts_device_check")
        _device_ok = True

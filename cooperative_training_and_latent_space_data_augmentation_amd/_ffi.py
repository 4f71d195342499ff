"""ctypes binding of libctl_hip.so (include/ctl_hip.h).  There is NO fallback: if the HIP library is missing or a call
fails, the product path raises -- it never routes through PyTorch ops or the CPU oracle."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libctl_hip.so")
ABI_VERSION = 11                     # CTL_ABI_VERSION of include/ctl_hip.h this binding was written against
RED_BLOCKS = 512                     # CTL_RED_BLOCKS of ctl_hip.h; checked against the library's compiled value (ctl_red_blocks) at load

# enums of ctl_hip.h
IN_PLAIN, IN_UP2, IN_ZINS2, IN_C4 = 0, 1, 2, 3
ACT_NONE, ACT_LEAKY, ACT_SIGMOID = 0, 1, 2
EPI_BIAS, EPI_ACCUM, EPI_RES, EPI_STATS, EPI_BNBWD, EPI_TAILBWD = 1, 2, 4, 8, 16, 32
(OP_CONV, OP_WGRAD, OP_WGRAD_REDUCE, OP_PACK, OP_BN_FINALIZE, OP_BN_EVAL, OP_BN_ACT, OP_BWD_REDUCE, OP_BN_BWD_FINALIZE,
 OP_BWD_APPLY, OP_CHAN_SUM_FINALIZE, OP_SUMPOOL2, OP_SIGMOID_BWD, OP_ZERO, OP_COPY, OP_PACK_BATCH, OP_WGRAD_REDUCE_BATCH, OP_DROPOUT2D, OP_BN_REPLAY,
 OP_WGRAD_GROUP) = range(1, 21)
WGRAD_GROUP_MAX = 8                  # members of one grouped weight-gradient launch (CTL_OP_WGRAD_GROUP)
OP_MAX_T = 14

CONV_DTYPE = np.dtype([
    ("n", "<i4"), ("hin", "<i4"), ("win", "<i4"), ("cin", "<i4"), ("hout", "<i4"), ("wout", "<i4"), ("cout", "<i4"),
    ("ks", "<i4"), ("stride", "<i4"), ("pad", "<i4"), ("in_mode", "<i4"), ("pro_affine", "<i4"), ("pro_slope", "<f4"),
    ("epi_flags", "<i4"), ("epi_act", "<i4"), ("epi_slope", "<f4"), ("out_h", "<i4"), ("out_w", "<i4"),
    ("out_sy", "<i4"), ("out_sx", "<i4"), ("nsub", "<i4"), ("out_sub", "<i4"), ("groups", "<i4"), ("dt", "<i4")])
DT_BF16, DT_X16, DT_Y16, DT_RES16, DT_X3 = 1, 2, 4, 8, 16
PACK_X3 = 16                         # or-ed into the mode word of a pack record (CTL_PACK_X3)
OP_DTYPE = np.dtype([("kind", "<i4"), ("i", "<i4", (27,)), ("f", "<f4", (4,)), ("slot", "<i4", (OP_MAX_T,)),
                     ("off", "<i8", (OP_MAX_T,)), ("l", "<i8", (4,))], align=True)


class CtlError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        self._lib = None

    def load(self):
        if self._lib is not None:
            return self._lib
        if not os.path.exists(LIB_PATH):
            raise CtlError(f"HIP extension not built: {LIB_PATH} is missing. Run `python -c 'import __graft_entry__ as g; "
                           "g.build()'` (or `make -C <package>/csrc`). There is no fallback path.")
        # PyTorch first: its wheel carries its own libamdhip64.so, and the extension must bind to THAT runtime (the one that owns the
        # device, the streams and the memory).  Loaded before torch, the extension pulled in /opt/rocm's copy instead and every launch
        # failed with "no ROCm-capable device is detected" (build() followed by smoke() in one process).
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        lib.ctl_last_error.restype = C.c_char_p
        lib.ctl_version.restype = C.c_int
        lib.ctl_launch_count.restype = C.c_ulonglong
        for name in ("ctl_conv_wpack_floats", "ctl_conv_stats_floats", "ctl_wgrad_partial_floats",
                     "ctl_wgrad_bias_partial_floats", "ctl_latent_score_ws_floats", "ctl_latent_mask_apply_ws_floats",
                     "ctl_rescale_intensity_ws_floats", "ctl_sizeof_op", "ctl_sizeof_conv", "ctl_latent_mask_fused_ws_floats", "ctl_conv_wpack_floats_x3"):
            getattr(lib, name).restype = C.c_size_t
        p, i32, i64, f32, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint64
        sig = {
            "ctl_conv_wpack_floats": [i32, i32, i32],
            "ctl_conv_stats_floats": [p], "ctl_conv_stats_blocks": [p],
            "ctl_wgrad_splits": [p], "ctl_wgrad_partial_floats": [p], "ctl_wgrad_bias_partial_floats": [p],
            "ctl_pack_weights": [p, p, i32, i32, i32, i64, i64, i64, i64, i32, p],
            "ctl_conv_forward": [p] * 12, "ctl_conv_forward_ex": [p] * 16, "ctl_conv_pool_ok": [p],
            "ctl_conv_wgrad": [p] * 8, "ctl_conv_wgrad_ex": [p] * 10,
            "ctl_wgrad_group_class": [p, i32], "ctl_wgrad_group_plan": [p, i32, p], "ctl_conv_wgrad_group": [i32] + [p] * 11,
            "ctl_wgrad_reduce": [p, p, p, p, i64, i64, i64, i64, p, i32, p],
            "ctl_confusion_hist": [p, p, i64, i32, p, p],
            "ctl_rescale_intensity": [p, p, p, i32, i64, f32, f32, f32, p],
            "ctl_noise_clamp": [p, p, u64, f32, f32, f32, p, i64, p],
            "ctl_rescale_intensity_ws_floats": [i32],
            "ctl_crop_or_pad": [p, p, i32, i32, i32, i32, i32, i32, p],
            "ctl_bn_finalize": [p, i32, i32, i64, p, p, f32, f32, i32, p, p, p, p, p, p, p, i32, p],
            "ctl_bn_finalize_ex": [p, i32, i32, i64, p, p, f32, f32, i32, p, p, p, p, p, p, p, p, i32, p],
            "ctl_bn_replay_running": [p, p, p, p, i32, f32, p],
            "ctl_bn_eval_coeffs": [i32, p, p, p, p, f32, p, p, i32, p],
            "ctl_bn_act": [p, p, p, f32, p, i64, i32, i32, p],
            "ctl_bwd_reduce": [i32, p, p, p, p, p, f32, i64, i32, p, i32, p], "ctl_red_blocks": [],
            "ctl_bn_bwd_finalize": [p, i32, i64, p, p, p, p, p, p, i32, i32, i32, p],
            "ctl_bn_bwd_finalize_ex": [p, i32, i64, p, p, p, p, p, p, i32, i32, i32, C.c_uint32, p],
            "ctl_bwd_apply": [i32, p, p, p, p, p, f32, p, i64, i32, p, p, i32, p],
            "ctl_chan_sum_finalize": [p, i32, p, i32, p],
            "ctl_sumpool2": [p, p, i32, i32, i32, i32, i32, p],
            "ctl_sigmoid_bwd": [p, p, p, i64, p],
            "ctl_softmax_t_fwd": [p, f32, p, i64, i32, p],
            "ctl_softmax_t_bwd": [p, p, f32, p, i64, i32, p],
            "ctl_onehot": [p, p, i64, i32, p],
            "ctl_ce2d_fwd": [p, p, i64, i32, p, p, p],
            "ctl_ce2d_bwd": [p, p, p, i64, i32, p, p],
            "ctl_mse_fwd": [p, p, i64, f32, p, p, p],
            "ctl_mse_bwd": [p, p, p, i64, f32, p, p],
            "ctl_argmax_c": [p, p, i64, i32, p],
            "ctl_latent_score_ws_floats": [i32, i32, i32, i32],
            "ctl_latent_score": [i32, p, p, p, i32, i32, i32, p],
            "ctl_latent_mask_apply": [i32, p, p, p, i32, p, p, p, p, i32, i32, i32, p],
            "ctl_latent_mask_apply_ws_floats": [i32, i32, i32, i32],
            "ctl_latent_mask_fused_ws_floats": [i32, i32, i32, i32],
            "ctl_latent_mask_fused": [i32, p, p, p, i32, p, p, p, p, p, i32, i32, i32, p],
            "ctl_dropout2d": [p, p, u64, f32, p, p, i32, i32, i32, p],
            "ctl_uniform": [p, i64, u64, p],
            "ctl_step_tick": [p, p],
            "ctl_spin": [i32, p],
            "ctl_dropout2d_ex": [p, p, u64, p, f32, p, p, p, i32, i32, i32, p],
            "ctl_dropout2d_dt": [p, p, u64, p, f32, p, p, i32, i32, i32, C.c_uint32, p],
            "ctl_uniform_dev": [p, i64, u64, p, p],
            "ctl_adam_dev": [p, p, p, p, i64, f32, f32, f32, f32, p, f32, p],
            "ctl_adam": [p, p, p, p, i64, f32, f32, f32, f32, i32, f32, p],
            "ctl_accumulate": [p, p, i32, i64, p],
            "ctl_plan_run": [p, i32, p, i32, p],
            "ctl_prof_start": [C.c_char_p], "ctl_prof_start_sampled": [C.c_char_p, i32], "ctl_prof_stop": [p, C.c_size_t],
            "ctl_pack_weights_batched": [p, p, p, i32, i64, p], "ctl_wgrad_reduce_batched": [p, p, p, i32, i64, p],
            "ctl_pack_weights_bf16_batched": [p, p, p, i32, i64, p],
            "ctl_pack_weights_x3_batched": [p, p, p, i32, i64, p], "ctl_conv_wpack_floats_x3": [i32, i32, i32],
            "ctl_bwd_reduce_rows": [i32, i64, i32],
            "ctl_bn_act_dt": [p, p, p, f32, p, i64, i32, i32, C.c_uint32, p],
            "ctl_bwd_reduce_dt": [i32, p, p, p, p, p, f32, i64, i32, p, i32, C.c_uint32, p, p],
            "ctl_bwd_apply_dt": [i32, p, p, p, p, p, f32, p, i64, i32, p, p, i32, C.c_uint32, p],
            "ctl_sumpool2_dt": [p, p, i32, i32, i32, i32, i32, C.c_uint32, p],
        }
        for name, args in sig.items():
            getattr(lib, name).argtypes = args
        if lib.ctl_version() != ABI_VERSION or lib.ctl_sizeof_op() != OP_DTYPE.itemsize or lib.ctl_sizeof_conv() != CONV_DTYPE.itemsize:
            raise CtlError(f"ABI mismatch: library version {lib.ctl_version()} vs binding {ABI_VERSION}, sizeof(ctl_op)={lib.ctl_sizeof_op()} "
                           f"vs {OP_DTYPE.itemsize}, sizeof(ctl_conv)={lib.ctl_sizeof_conv()} vs {CONV_DTYPE.itemsize}")
        if lib.ctl_red_blocks() != RED_BLOCKS:
            raise CtlError(f"ABI mismatch: the library was built with CTL_RED_BLOCKS={lib.ctl_red_blocks()}, this binding sizes scratch for {RED_BLOCKS}")
        self._lib = lib
        return lib

    def __getattr__(self, name):
        return getattr(self.load(), name)


lib = _Lib()

# every symbol include/ctl_hip.h declares (checked by tests/test_cabi.py without a GPU)
EXPORTED = ["ctl_version", "ctl_last_error", "ctl_conv_wpack_floats", "ctl_conv_stats_floats", "ctl_conv_stats_blocks",
            "ctl_pack_weights", "ctl_conv_forward", "ctl_conv_forward_ex", "ctl_conv_pool_ok", "ctl_wgrad_splits", "ctl_wgrad_partial_floats",
            "ctl_wgrad_bias_partial_floats", "ctl_conv_wgrad", "ctl_conv_wgrad_ex", "ctl_wgrad_reduce", "ctl_bn_finalize", "ctl_bn_finalize_ex", "ctl_bn_replay_running", "ctl_bn_eval_coeffs",
            "ctl_bn_act", "ctl_bwd_reduce", "ctl_bn_bwd_finalize", "ctl_bn_bwd_finalize_ex", "ctl_bwd_apply", "ctl_chan_sum_finalize", "ctl_sumpool2",
            "ctl_sigmoid_bwd", "ctl_softmax_t_fwd", "ctl_softmax_t_bwd", "ctl_onehot", "ctl_ce2d_fwd", "ctl_ce2d_bwd",
            "ctl_mse_fwd", "ctl_mse_bwd", "ctl_argmax_c", "ctl_latent_score_ws_floats", "ctl_latent_score",
            "ctl_latent_mask_apply", "ctl_latent_mask_apply_ws_floats", "ctl_dropout2d", "ctl_uniform", "ctl_adam", "ctl_plan_run", "ctl_sizeof_op",
            "ctl_sizeof_conv", "ctl_prof_start", "ctl_prof_start_sampled", "ctl_prof_stop", "ctl_pack_weights_batched",
            "ctl_wgrad_reduce_batched", "ctl_confusion_hist", "ctl_rescale_intensity_ws_floats", "ctl_rescale_intensity",
            "ctl_noise_clamp", "ctl_crop_or_pad", "ctl_step_tick", "ctl_spin", "ctl_dropout2d_ex", "ctl_dropout2d_dt", "ctl_uniform_dev", "ctl_adam_dev",
            "ctl_latent_mask_fused_ws_floats", "ctl_latent_mask_fused", "ctl_accumulate", "ctl_pack_weights_bf16_batched",
            "ctl_bn_act_dt", "ctl_bwd_reduce_dt", "ctl_bwd_apply_dt", "ctl_sumpool2_dt", "ctl_bwd_reduce_rows", "ctl_red_blocks",
            "ctl_launch_count", "ctl_conv_wpack_floats_x3", "ctl_pack_weights_x3_batched", "ctl_wgrad_group_class", "ctl_wgrad_group_plan",
            "ctl_conv_wgrad_group"]


def prof_start(kernel_filter: str = "", every: int = 1) -> None:
    """Bracket the launches whose id contains `kernel_filter` with HIP events on their own stream (every `every`-th one)."""
    check(lib.ctl_prof_start_sampled(kernel_filter.encode(), int(every)), "ctl_prof_start_sampled")


def prof_stop() -> dict:
    """{kernel id: dict(launches, ms, flops, bytes)} for the launches bracketed since prof_start."""
    buf = C.create_string_buffer(1 << 18)
    check(lib.ctl_prof_stop(buf, len(buf)), "ctl_prof_stop")
    out = {}
    for line in buf.value.decode().splitlines():
        parts = line.split()
        out[parts[0]] = {k: float(v) for k, v in (kv.split("=") for kv in parts[1:])}
    return out


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib.ctl_last_error()
        raise CtlError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def conv_desc(**kw) -> np.ndarray:
    """Build a ctl_conv record (numpy scalar array of CONV_DTYPE) with defaults for the plain, non-scattered case."""
    d = np.zeros((), dtype=CONV_DTYPE)
    d["pad"] = 1 if kw.get("ks", 3) in (3, 4) else 0
    d["stride"] = 1
    d["nsub"] = 1
    d["groups"] = 1
    d["out_sy"] = d["out_sx"] = 1
    for k, v in kw.items():
        d[k] = v
    if "out_h" not in kw:
        d["out_h"] = d["hout"]
    if "out_w" not in kw:
        d["out_w"] = d["wout"]
    return d


def desc_ptr(d: np.ndarray) -> int:
    return d.ctypes.data

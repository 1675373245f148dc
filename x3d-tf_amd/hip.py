"""ctypes binding of libx3d_hip.so (C ABI in include/x3d_hip.h).

There is no fallback: if the library is missing or a call fails this raises.  torch is used only for
device memory (``tensor.data_ptr()``) and the current HIP stream.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("X3D_HIP_LIB") or os.path.join(_HERE, "libx3d_hip.so")   # X3D_HIP_LIB: A/B builds (tools/build_variant.sh)

ABI_VERSION = 133   # X3D_ABI_VERSION of the include/x3d_hip.h the signatures below were written against
F32, BF16, F16 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_SWISH, ACT_SIGMOID = 0, 1, 2, 3
EPI_STORE, EPI_ADD, EPI_ADD_STRIDED, EPI_SWISH_BWD = 0, 1, 2, 3

_vp, _i, _f, _d, _ll = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_longlong


class X3DHipError(RuntimeError):
    pass


class PwFwdArgs(C.Structure):
    _fields_ = [("x", _vp), ("w", _vp), ("y", _vp), ("stats", _vp), ("in_scale_shift", _vp),
                ("in_gate", _vp), ("in_act", _i), ("N", _i), ("Cin", _i), ("Cout", _i), ("T", _i),
                ("H", _i), ("W", _i), ("stride", _i), ("dtype", _i), ("w_panel", _vp),
                ("in_add", _vp), ("in_add_scale_shift", _vp), ("in_store", _vp),
                ("out_scale_shift", _vp), ("out_add", _vp), ("out_add_scale_shift", _vp), ("out_act", _i)]


class PwDgradArgs(C.Structure):
    _fields_ = [("g", _vp), ("yraw", _vp), ("coef", _vp), ("w", _vp), ("dx", _vp), ("epi", _i),
                ("add", _vp), ("braw", _vp), ("b_scale_shift", _vp), ("gate", _vp), ("nc_sums", _vp),
                ("N", _i), ("Cin", _i), ("Cout", _i), ("T", _i), ("H", _i), ("W", _i), ("dtype", _i),
                ("w_panel", _vp), ("coef_fold", _vp)]


class PwBwdArgs(C.Structure):
    _fields_ = [("g", _vp), ("yraw", _vp), ("coef", _vp), ("w_panel", _vp), ("dx", _vp), ("epi", _i),
                ("add", _vp), ("braw", _vp), ("b_scale_shift", _vp), ("gate", _vp), ("nc_sums", _vp),
                ("x", _vp), ("dw", _vp),
                ("N", _i), ("Cin", _i), ("Cout", _i), ("T", _i), ("H", _i), ("W", _i), ("dtype", _i),
                ("tail_c", _vp), ("tail_r", _vp), ("tail_sums_c", _vp), ("tail_sums_r", _vp),
                ("rc_panel", _vp), ("rc_c0", _vp), ("rc_sums", _vp), ("x_stride", _i), ("xH", _i), ("xW", _i),
                ("dw_slab", _vp), ("dw_slab_parts", _i), ("coef_fold", _vp)]


class DwReduceJob(C.Structure):
    _fields_ = [("slab", _vp), ("dw", _vp), ("parts", _i), ("elems", _i)]


class EvalViewsArgs(C.Structure):
    _fields_ = [("video", _vp), ("out", _vp), ("F", _i), ("H", _i), ("W", _i), ("T", _i), ("views", _i),
                ("crops", _i), ("size", _i), ("mean", _f * 3), ("std", _f * 3), ("dtype", _i)]


class TrainClipArgs(C.Structure):
    _fields_ = [("video", _vp), ("out", _vp), ("F", _i), ("H", _i), ("W", _i), ("T", _i), ("rate", _i), ("start", _i),
                ("jitter", _f), ("size", _i), ("y0", _i), ("x0", _i), ("flip", _i), ("mean", _f * 3), ("std", _f * 3),
                ("dtype", _i)]


class BnEvalItem(C.Structure):
    _fields_ = [("gamma", _vp), ("beta", _vp), ("moving_mean", _vp), ("moving_var", _vp), ("scale_shift", _vp),
                ("mean_invstd", _vp), ("C", _i)]


class PwPackItem(C.Structure):
    _fields_ = [("w", _vp), ("fwd_panel", _vp), ("dgrad_panel", _vp), ("Cout", _i), ("Cin", _i)]


class PwWgradArgs(C.Structure):
    _fields_ = [("g", _vp), ("yraw", _vp), ("coef", _vp), ("x", _vp), ("in_scale_shift", _vp),
                ("in_gate", _vp), ("in_act", _i), ("dw", _vp), ("N", _i), ("Cin", _i), ("Cout", _i),
                ("T", _i), ("H", _i), ("W", _i), ("stride", _i), ("dtype", _i), ("dw_slab", _vp), ("dw_slab_parts", _i),
                ("coef_fold", _vp)]


class BnBwdFold(C.Structure):
    """x3d_bn_bwd_fold: the BatchNorm-backward finalize folded into its consumers (coef_fold of the backward argument structs)."""
    _fields_ = [("sums", _vp), ("count", _d), ("mean_invstd", _vp), ("gamma", _vp), ("dgamma", _vp), ("dbeta", _vp),
                ("coef_out", _vp)]


# address -> BnBwdFold: argument structs refer to a fold by address (tools that walk a plan's pointers follow it).  Weak values:
# the plan (or the ops wrapper) that built a fold keeps it alive for as long as its launches exist; an entry whose owner is gone
# disappears with it instead of accumulating -- and instead of handing dispatch._struct_pointers stale device addresses.
import weakref  # noqa: E402
FOLDS = weakref.WeakValueDictionary()


def fold_address(f: BnBwdFold) -> int:
    """Address to put into an argument struct's `coef_fold`; the caller keeps `f` alive for as long as launches use it."""
    a = C.addressof(f)
    FOLDS[a] = f
    return a


class BnFold(C.Structure):
    _fields_ = [("stats", _vp), ("count", _d), ("gamma", _vp), ("beta", _vp), ("moving_mean", _vp), ("moving_var", _vp),
                ("eps", _f), ("momentum", _f), ("update_moving", _i), ("scale_shift", _vp), ("mean_invstd", _vp)]


class Dw3dFwdArgs(C.Structure):
    _fields_ = [("x", _vp), ("w", _vp), ("y", _vp), ("in_scale_shift", _vp), ("in_act", _i),
                ("stats", _vp), ("pool", _vp), ("N", _i), ("C", _i), ("T", _i), ("H", _i), ("W", _i),
                ("stride", _i), ("dtype", _i), ("in_bn", C.POINTER(BnFold))]


class Dw3dBwdArgs(C.Structure):
    _fields_ = [("dv", _vp), ("braw", _vp), ("coef_nc", _vp), ("araw", _vp), ("a_scale_shift", _vp),
                ("w", _vp), ("ga", _vp), ("a_sums", _vp), ("dw", _vp), ("N", _i), ("C", _i), ("T", _i),
                ("H", _i), ("W", _i), ("stride", _i), ("dtype", _i)]


class SeBnbBwdArgs(C.Structure):
    _fields_ = [("nc_sums", _vp), ("pool_sums", _vp), ("P", _d), ("b_scale_shift", _vp),
                ("b_mean_invstd", _vp), ("gamma_b", _vp), ("w1", _vp), ("b1", _vp), ("w2", _vp),
                ("b2", _vp), ("gate", _vp), ("hidden", _vp), ("dw1", _vp), ("db1", _vp), ("dw2", _vp),
                ("db2", _vp), ("dgamma_b", _vp), ("dbeta_b", _vp), ("coef_nc", _vp), ("scratch", _vp),
                ("N", _i), ("C", _i), ("Wd", _i), ("reduce", DwReduceJob * 2)]


_SIGS = {
    "x3d_version": ([], _i),
    "x3d_last_error": ([], C.c_char_p),
    "x3d_stem_s_nthwc_supported": ([_i, _i, _i, _i], _i),
    "x3d_stem_s_fwd": ([_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp], _i),
    "x3d_stem_s_wgrad": ([_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp], _i),
    "x3d_dwt_fwd": ([_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp], _i),
    "x3d_dwt_bwd": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp], _i),
    "x3d_stem_fused_supported": ([_i] * 9, _i),
    "x3d_stem_fwd": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp], _i),
    "x3d_stem_bwd": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp], _i),
    "x3d_stats_replicas": ([], _i),
    "x3d_stats_stride": ([_i], _ll),
    "x3d_bn_finalize": ([_vp, _d, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _vp, _i, _vp], _i),
    "x3d_bn_eval_coef_batched": ([_vp, _i, _f, _vp], _i),
    "x3d_bn_bwd_finalize": ([_vp, _d, _vp, _vp, _vp, _vp, _vp, _i, _vp], _i),
    "x3d_pw_fwd": ([C.POINTER(PwFwdArgs), _vp], _i),
    "x3d_pw_fwd_tail_supported": ([C.POINTER(PwFwdArgs)], _i),
    "x3d_pw_dgrad": ([C.POINTER(PwDgradArgs), _vp], _i),
    "x3d_pw_wgrad": ([C.POINTER(PwWgradArgs), _vp], _i),
    "x3d_pw_wgrad_dw_parts": ([C.POINTER(PwWgradArgs)], _i),
    "x3d_pw_bwd_supported": ([C.POINTER(PwBwdArgs)], _i),
    "x3d_pw_bwd": ([C.POINTER(PwBwdArgs), _vp], _i),
    "x3d_pw_bwd_dw_parts": ([C.POINTER(PwBwdArgs)], _i),
    "x3d_pw_coef_fold_supported": ([C.POINTER(PwDgradArgs), C.POINTER(PwWgradArgs), C.POINTER(PwBwdArgs)], _i),
    "x3d_dw_slab_reduce": ([C.POINTER(DwReduceJob), _i, _vp], _i),
    "x3d_pw_bwd_rc_panel_elems": ([_i, _i], _ll),
    "x3d_pw_bwd_rc_sums_elems": ([_i, _i], _ll),
    "x3d_pw_bwd_rc_prepare": ([_vp, _vp, _vp, _vp, _i, _i, _i, _vp], _i),
    "x3d_pw_bwd_rc_finish": ([_vp, _vp, _vp, _vp, _i, _i, _i, _vp], _i),
    "x3d_bn_bwd_finalize_rc": ([_vp, _d, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp], _i),
    "x3d_pw_kernel_name": ([C.POINTER(PwFwdArgs), C.POINTER(PwDgradArgs), C.POINTER(PwWgradArgs), C.POINTER(PwBwdArgs),
                            C.c_char_p, _i], _i),
    "x3d_pw_panel_elems": ([_i, _i], _ll),
    "x3d_pw_pack_weights": ([_vp, _i, _i, _vp], _i),
    "x3d_dw3d_fwd": ([C.POINTER(Dw3dFwdArgs), _vp], _i),
    "x3d_dw3d_bwd": ([C.POINTER(Dw3dBwdArgs), _vp], _i),
    "x3d_dw3d_kernel_name": ([C.POINTER(Dw3dFwdArgs), C.POINTER(Dw3dBwdArgs), C.c_char_p, _i], _i),
    "x3d_se_fwd": ([_vp, _d, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp], _i),
    "x3d_se_bnb_bwd": ([C.POINTER(SeBnbBwdArgs), _vp], _i),
    "x3d_tail_fwd": ([_vp, _vp, _vp, _vp, _vp, _i, _i, _ll, _i, _vp], _i),
    "x3d_tail_fwd_bn": ([_vp, C.POINTER(BnFold), _vp, C.POINTER(BnFold), _vp, _i, _i, _ll, _i, _vp], _i),
    "x3d_tail_bwd": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _ll, _i, _vp], _i),
    "x3d_relu_bn_bwd_reduce": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _ll, _i, _vp], _i),
    "x3d_pool_fwd": ([_vp, _vp, _vp, _i, _i, _ll, _i, _vp], _i),
    "x3d_subsample2": ([_vp, _vp, _ll, _i, _i, _i, _vp], _i),
    "x3d_dense_fwd": ([_vp, _vp, _f, _vp, _vp, _vp, _i, _i, _i, _i, _vp], _i),
    "x3d_dense_bwd": ([_vp, _vp, _i, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _vp], _i),
    "x3d_softmax_xent": ([_vp, _vp, _vp, _vp, _vp, _f, _i, _i, _vp], _i),
    "x3d_view_mean": ([_vp, _vp, _i, _i, _i, _vp], _i),
    "x3d_sgd_nesterov": ([_vp, _vp, _vp, _vp, _f, _f, _f, _f, _ll, _vp], _i),
    "x3d_adam": ([_vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _f, _f, _ll, _ll, _vp], _i),
    "x3d_all_finite": ([_vp, _ll, _vp, _vp], _i),
    "x3d_l2_sumsq": ([_vp, _vp, _vp, _ll, _vp], _i),
    "x3d_nthwc_to_ncthw": ([_vp, _i, _vp, _i, _i, _i, _ll, _vp], _i),
    "x3d_eval_views": ([C.POINTER(EvalViewsArgs), _vp], _i),
    "x3d_train_clip": ([C.POINTER(TrainClipArgs), _vp], _i),
    "x3d_train_resized_hw": ([_i, _i, _f, C.POINTER(_i), C.POINTER(_i)], _i),
    "x3d_crc32c": ([C.c_char_p, C.c_size_t, C.c_uint32], C.c_uint32),
}

_lib = None


def exported_symbols():
    """Names include/x3d_hip.h declares (and this binding expects)."""
    return sorted(_SIGS)


def load(path=None):
    """dlopen libx3d_hip.so and type every entry point.  Raises if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise X3DHipError(
            f"{p} not found: build it with `python x3d-tf_amd/build.py` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for the X3D hot path.")
    lib = C.CDLL(p)
    lib.x3d_version.restype = _i
    have = int(lib.x3d_version())
    if have != ABI_VERSION:     # a stale .so (they ship out of band, git-ignored) would take shifted arguments silently
        raise X3DHipError(f"{p} has ABI version {have}, this binding needs {ABI_VERSION}: rebuild it "
                          "(`python x3d-tf_amd/build.py`)")
    for name, (argtypes, restype) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = restype
    if path is None:
        _lib = lib
    return lib


def dw3d_kernel_name(args):
    """Instantiation x3d_dw3d_fwd / x3d_dw3d_bwd would launch for this argument struct (no launch)."""
    buf = C.create_string_buffer(128)
    fwd = C.byref(args) if isinstance(args, Dw3dFwdArgs) else None
    bwd = C.byref(args) if isinstance(args, Dw3dBwdArgs) else None
    rc = load().x3d_dw3d_kernel_name(fwd, bwd, buf, 128)
    if rc != 0:
        raise X3DHipError(load().x3d_last_error().decode())
    return buf.value.decode()


def pw_kernel_name(args):
    """Instantiation x3d_pw_fwd / x3d_pw_dgrad / x3d_pw_wgrad / x3d_pw_bwd would launch for this argument struct (no
    launch, no GPU needed)."""
    buf = C.create_string_buffer(160)
    slots = [None, None, None, None]
    for i, kind in enumerate((PwFwdArgs, PwDgradArgs, PwWgradArgs, PwBwdArgs)):
        if isinstance(args, kind):
            slots[i] = C.byref(args)
    rc = load().x3d_pw_kernel_name(*slots, buf, 160)
    if rc != 0:
        raise X3DHipError(load().x3d_last_error().decode())
    return buf.value.decode()


def kernel_name(args):
    """pw_kernel_name / dw3d_kernel_name by struct type."""
    return dw3d_kernel_name(args) if isinstance(args, (Dw3dFwdArgs, Dw3dBwdArgs)) else pw_kernel_name(args)


def dtype_code(dt):
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    if dt == torch.float16:
        return F16
    raise X3DHipError(f"unsupported activation dtype {dt}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def check(status, what=""):
    if status != 0:
        msg = load().x3d_last_error()
        raise X3DHipError(f"{what} failed ({status}): {msg.decode() if msg else ''}")


def call(name, *args):
    """Call a plain-argument entry point on the current stream."""
    lib = load()
    check(getattr(lib, name)(*args, stream_ptr()), name)


def call_struct(name, struct):
    lib = load()
    check(getattr(lib, name)(C.byref(struct), stream_ptr()), name)


def stats_layout(c: int):
    """(replicas, stride in doubles) of a statistics accumulator for c channels (include/x3d_hip.h)."""
    lib = load()
    return int(lib.x3d_stats_replicas()), int(lib.x3d_stats_stride(int(c)))

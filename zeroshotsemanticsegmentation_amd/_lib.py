"""ctypes binding of libszn_hip.so (the C-ABI declared in include/szn.h).

The product path has no CPU fallback: if the HIP library is missing or a call fails, `SznError` is raised.
Every wrapper takes torch tensors (device memory owners) and passes raw pointers + the current HIP stream.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SZN_LIB_PATH") or os.path.join(_HERE, "lib", "libszn_hip.so")   # (override: A/B builds in tools/)

SZN_F32, SZN_BF16, SZN_F16 = 0, 1, 2


class SznError(RuntimeError):
    pass


class CallResult(C.Structure):
    """szn_call_result_t: what a call reports back (rows written into its colsum slab, fraction of the dense tiles it executed)"""
    _fields_ = [("colsum_rows", C.c_int), ("work_fraction", C.c_float)]


class ConvDesc(C.Structure):
    """szn_conv_desc_t.  Every instance owns a CallResult (`d.res`) and points `result` at it, so that after szn_conv2d_fwd / _dgrad /
    _wgrad / _wgrad_adam `d.res.colsum_rows` / `d.res.work_fraction` hold what THAT call decided (no thread-local "last call" state)."""
    _fields_ = [(n, C.c_int) for n in (
        "dtype", "B", "Hi", "Wi", "Ci", "Ho", "Wo", "Co", "KH", "KW", "pad", "ldi", "ldo", "ldg", "relu", "out_f32")] + [
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("colsum", C.c_void_p), ("pool_out", C.c_void_p),
        ("colsum_slab", C.c_void_p), ("colsum_slab_rows", C.c_int), ("pool_code", C.c_void_p), ("pool_only", C.c_int),
        ("cb_on", C.c_int), ("cb_rect", C.c_int * 4), ("cb_const", C.c_int * 4), ("reserved_cus", C.c_int),
        ("dw_lp", C.c_void_p), ("dw_lp_dtype", C.c_int), ("result", C.POINTER(CallResult))]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.res = CallResult(0, 1.0)
        self.result = C.pointer(self.res)


class DeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("arch", C.c_char * 32), ("compute_units", C.c_int),
                ("wavefront", C.c_int), ("lds_bytes_per_block", C.c_int), ("hbm_bytes", C.c_int64),
                ("clock_mhz", C.c_int)]


class AdamArgs(C.Structure):
    """szn_adam_args_t (include/szn.h): state and hyper-parameters of the Adam step fused into szn_conv2d_wgrad_adam"""
    _fields_ = [("param", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("w_lp", C.c_void_p),
                ("w_lp_dtype", C.c_int), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("step", C.c_int), ("grad_scale", C.c_float), ("grad_optional", C.c_int)]


MAX_CLASSES = 256           # SZN_MAX_CLASSES


class ClassSet(C.Structure):
    """szn_class_set (include/szn.h): bit k % 64 of w[k // 64] = class k"""
    _fields_ = [("w", C.c_uint64 * 4)]


def class_set(classes):
    """iterable of class indices -> a byref(ClassSet) argument for the *_k entry points (None = the empty set)"""
    cs = ClassSet()
    any_ = False
    for k in classes or []:
        k = int(k)
        if not 0 <= k < MAX_CLASSES:
            raise SznError("class index %d outside [0, %d)" % (k, MAX_CLASSES))
        cs.w[k >> 6] |= 1 << (k & 63)
        any_ = True
    return C.byref(cs) if any_ else None


_P, _I, _L, _F, _U64, _SZ = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_uint64, C.c_size_t
_D = C.POINTER(ConvDesc)
_IP = C.POINTER(C.c_int)
_CS = C.POINTER(ClassSet)

# name -> (restype, argtypes); must list every symbol of include/szn.h (tests/test_abi.py checks that)
SIGNATURES = {
    "szn_last_error": (C.c_char_p, []),
    "szn_last_kernel": (C.c_char_p, []),
    "szn_prev_kernel": (C.c_char_p, []),
    "szn_version": (_I, []),
    "szn_knob_count": (_I, []),
    "szn_knob_name": (C.c_char_p, [_I]),
    "szn_device_info": (_I, [_I, C.POINTER(DeviceInfo)]),
    "szn_stream_create_cu_mask": (_I, [_I, C.POINTER(C.c_uint32), C.POINTER(C.c_void_p)]),
    "szn_stream_destroy": (_I, [C.c_void_p]),
    "szn_conv2d_fwd": (_I, [_D, _P, _P, _P, _P, _P, _P, _P]),
    "szn_pack_weight_dgrad": (_I, [_I, _I, _I, _I, _I, _P, _P, _P]),
    "szn_pack_weight_dgrad_batch": (_I, [_I, _I, _P, _P, _P, _P, _P, _P]),
    "szn_conv2d_dgrad": (_I, [_D, _P, _P, _P, _P, _P, _P]),
    "szn_conv2d_dgrad_border_region": (_I, [_D, _P]),
    "szn_conv2d_dgrad_border_finish": (_I, [_D, _P, _P, _P, _P, _P, _P, _P]),
    "szn_conv2d_dgrad_gemm_workspace_bytes": (C.c_size_t, [_D]),
    "szn_conv2d_dgrad_gemm": (_I, [_D, _P, _P, _P, _P]),
    "szn_conv2d_dgrad_gemm_native_supported": (_I, [_D]),
    "szn_conv2d_dgrad_gemm_native_workspace_bytes": (C.c_size_t, [_D]),
    "szn_conv2d_dgrad_gemm_native": (_I, [_D, _P, _P, _P, _P]),
    "szn_conv2d_wgrad": (_I, [_D, _P, _P, _P, _I, _P]),
    "szn_conv2d_wgrad_adam_supported": (_I, [_D]),
    "szn_conv2d_wgrad_adam": (_I, [_D, _P, _P, _P, C.POINTER(AdamArgs), _P]),
    "szn_bias_grad": (_I, [_I, _L, _I, _I, _P, _P, _I, _P]),
    "szn_bias_grad_slab": (_I, [_I, _L, _I, _I, _P, _P, _I, _P, _I, _IP, _P]),
    "szn_colsum_reduce_batch": (_I, [_I, _P, _P, _P, _P, _P]),
    "szn_gemm_proj_fwd": (_I, [_I, _L, _I, _I, _I, _P, _P, _P, _P, _P]),
    "szn_gemm_proj_dgrad": (_I, [_I, _L, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "szn_gemm_proj_wgrad": (_I, [_I, _L, _I, _I, _I, _P, _P, _P, _I, _P]),
    "szn_conv1_1_fwd": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "szn_conv1_1_fwd_c": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "szn_conv1_1_wgrad_c": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P, _P]),
    "szn_conv1_1_wgrad_workspace_bytes": (_SZ, [_I, _I, _I, _I, _I]),
    "szn_conv1_1_wgrad": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P]),
    "szn_conv1_1_wgrad_reads": (_I, [_I, _I, _I, _I, _I, C.POINTER(C.c_int)]),
    "szn_maxpool2x2_ceil_fwd": (_I, [_I, _I, _I, _I, _I, _P, _P, _P]),
    "szn_maxpool2x2_ceil_bwd": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _IP, _P]),
    "szn_maxpool2x2_ceil_fwd_code": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "szn_maxpool2x2_ceil_bwd_code": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _IP, _P]),
    "szn_maxpool2x2_ceil_bwd_code_gather": (_I, [_I, _I, _I, _I, _I, _P, _P, _I, _I, _P, _P, _P, _P, _P, _I, _IP, _P]),
    "szn_maxpool2x2_ceil_bwd_code_cb": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _IP, _P, _I, _P, _P, _P]),
    "szn_conv2d_wgrad_cb_region": (_I, [_D, _P]),
    "szn_bilinear_up32_crop_fwd": (_I, [_I] * 9 + [_P, _P, _P]),
    "szn_bilinear_up32_crop_bwd": (_I, [_I] * 9 + [_P, _P, _P]),
    "szn_bilinear_up_crop_fwd": (_I, [_I] * 10 + [_P, _P, _P]),
    "szn_bilinear_up_crop_bwd": (_I, [_I] * 10 + [_P, _P, _P]),
    "szn_bilinear_up2_nhwc_fwd": (_I, [_I] * 5 + [_P, _P, _P]),
    "szn_bilinear_up2_nhwc_bwd": (_I, [_I] * 5 + [_P, _P, _P]),
    "szn_deconv64s32_fwd": (_I, [_I] * 9 + [_P, _P, _P, _P]),
    "szn_deconv64s32_dgrad": (_I, [_I] * 9 + [_P, _P, _P, _P]),
    "szn_deconv64s32_wgrad": (_I, [_I] * 9 + [_P, _P, _P, _I, _P]),
    "szn_seenmask_head_workspace_bytes": (_SZ, [_I] * 6),
    "szn_seenmask_head": (_I, [_I] * 8 + [_P, _P, _P, _I, _U64] + [_P] * 8),
    "szn_seenmask_head_k": (_I, [_I] * 8 + [_P, _P, _P, _I, _CS] + [_P] * 8),
    "szn_seenmask_score_wgrad_workspace_bytes": (_SZ, [_L, _I]),
    "szn_seenmask_score_wgrad": (_I, [_I, _L, _I, _I, _P, _P, _P, _P, _P, _P]),
    "szn_loss_workspace_bytes": (_SZ, [_I, _I, _I]),
    "szn_cosine_loss_fwd": (_I, [_I] * 5 + [_P] * 8),
    "szn_cosine_loss_bwd": (_I, [_I] * 5 + [_P] * 8),
    "szn_mse_loss_fwd": (_I, [_I] * 5 + [_P] * 8),
    "szn_mse_loss_bwd": (_I, [_I] * 5 + [_P] * 8),
    "szn_ce2d_fwd": (_I, [_I] * 4 + [_P, _P, _P, _I, _P, _P, _P, _P, _P]),
    "szn_ce2d_bwd": (_I, [_I] * 4 + [_P, _P, _P, _I, _P, _P, _P, _P]),
    "szn_embed_argmax": (_I, [_I] * 5 + [_P, _P, _I, _U64, _P, _P, _P, _P]),
    "szn_embed_argmax_k": (_I, [_I] * 5 + [_P, _P, _I, _CS, _P, _P, _P, _P]),
    "szn_confusion_hist": (_I, [_L, _I, _P, _P, _U64, _P, _P]),
    "szn_confusion_hist_k": (_I, [_L, _I, _P, _P, _CS, _P, _P]),
    "szn_fused_head_workspace_bytes": (_SZ, [_I] * 5),
    "szn_fused_head": (_I, [_I] * 10 + [_P] * 6 + [_I, _P, _P, _P]),
    "szn_fused_head_strided": (_I, [_I] * 11 + [_P] * 6 + [_I, _P, _P, _P]),
    "szn_fused_head_prepare": (_I, [_I, _I, _P, _P, _P]),
    "szn_fused_head_prepared": (_I, [_I] * 11 + [_P] * 6 + [_I, _P, _P, _P]),
    "szn_adam_step": (_I, [_L, _P, _P, _P, _P, _F, _F, _F, _F, _F, _I, _F, _P, _I, _P]),
    "szn_sgd_momentum_step": (_I, [_L, _P, _P, _P, _F, _F, _F, _I, _F, _P, _I, _P]),
    "szn_adam_step_g16": (_I, [_L, _P, _P, _I, _P, _P, _F, _F, _F, _F, _F, _I, _F, _P, _I, _P]),
    "szn_sgd_momentum_step_g16": (_I, [_L, _P, _P, _I, _P, _F, _F, _F, _I, _F, _P, _I, _P]),
    "szn_grad_check_finite": (_I, [_L, _P, _P, _P]),
    "szn_adam_step_scaled": (_I, [_L, _P, _P, _P, _P, _F, _F, _F, _F, _F, _P, _F, _P, _I, _P]),
    "szn_sgd_momentum_step_scaled": (_I, [_L, _P, _P, _P, _F, _F, _F, _P, _F, _P, _I, _P]),
    "szn_loss_scale_update": (_I, [_P, _F, _F, _I, _F, _F, _P]),
    "szn_cast": (_I, [_I, _I, _L, _P, _P, _P]),
    "szn_band_remap": (_I, [_I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "szn_band_fold": (_I, [_I, _I, _I, _I, _I, _P, _I, _P, _I, _P]),
    "szn_dropout2d_mask": (_I, [_L, _F, _U64, _U64, _P, _P]),
    "szn_proj_fp8_workspace_bytes": (_SZ, [_L, _I, _I]),
    "szn_proj_fp8_fwd": (_I, [_I, _I, _L, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "szn_proj_fp8_bwd_workspace_bytes": (_SZ, [_L, _I, _I]),
    "szn_proj_fp8_dgrad": (_I, [_I, _I, _L, _I, _I, _I, _P, _P, _P, _I, _I, _P, _I, _I, _P, _I, _P, _P]),
    "szn_proj_fp8_wgrad": (_I, [_I, _I, _L, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "szn_image_u8_to_bgr_f32": (_I, [_I, _I, _I, _P, C.POINTER(C.c_double), _P, _P]),
}

_lib = None


def load():
    """Load libszn_hip.so (built in-tree by __graft_entry__.build() / csrc/Makefile). Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SznError("libszn_hip.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(the product path has no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise SznError("%s failed (%d): %s" % (what, rc, load().szn_last_error().decode()))


def last_kernel():
    """name of the kernel the library launched last on this thread (which specialised path the dispatcher took)"""
    return load().szn_last_kernel().decode()


def rows_out():
    """a fresh int out-parameter (colsum_rows_out of the pool / bias-gradient entry points): pass C.byref(r), read r.value"""
    return C.c_int(0)


def prev_kernel():
    """the launch before last_kernel() (e.g. the GEMM kernel in front of a split-K epilogue)"""
    return load().szn_prev_kernel().decode()


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def dtype_code(t):
    if t == torch.float32:
        return SZN_F32
    if t == torch.bfloat16:
        return SZN_BF16
    if t == torch.float16:
        return SZN_F16
    raise SznError("unsupported dtype %s" % t)


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    check(rc, name)

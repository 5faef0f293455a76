"""ctypes binding of libartspeech_hip.so (include/artspeech_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call fails, this raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AS_LIB_PATH") or os.path.join(_HERE, "lib", "libartspeech_hip.so")   # override: kernel experiments
_lib = None

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_sz = ctypes.c_size_t
c_f = ctypes.c_float

_SIGNATURES = {
    "as_abi_version": (c_i, []),
    "as_prof_enable": (c_i, [c_i]),
    "as_prof_collect": (c_i, [c_p, c_p, c_p, c_p, c_i]),
    "as_mas_workspace_bytes": (c_sz, [c_i, c_i, c_i]),
    "as_mas_f32": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_sz, c_p]),
    "as_softmax_mas_f32": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_sz, c_p]),
}

AS_MAX_TAPS = 25


class ConvGemmArgs(ctypes.Structure):
    _fields_ = [("Wh", ctypes.c_void_p), ("W", ctypes.c_void_p), ("X", ctypes.c_void_p), ("Xh", ctypes.c_void_p), ("Y", ctypes.c_void_p),
                ("Yh", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("res", ctypes.c_void_p), ("meta", ctypes.c_void_p),
                ("ws", ctypes.c_void_p), ("ws_bytes", ctypes.c_size_t),
                ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32), ("T", ctypes.c_int32),
                ("Kp", ctypes.c_int32),
                ("ldx", ctypes.c_int32), ("ldy", ctypes.c_int32), ("ldr", ctypes.c_int32),
                ("act", ctypes.c_int32), ("div_sqrt2", ctypes.c_int32), ("in_act", ctypes.c_int32),
                ("transpose_out", ctypes.c_int32), ("yh_lrelu", ctypes.c_int32), ("n_prod", ctypes.c_int32),
                ("dh", ctypes.c_int32 * AS_MAX_TAPS), ("dw", ctypes.c_int32 * AS_MAX_TAPS),
                ("in_slope", ctypes.c_float), ("act_slope", ctypes.c_float), ("acc_scale", ctypes.c_float),
                ("n_groups", ctypes.c_int32), ("group_cols", ctypes.c_int32)]


_SIGNATURES.update({
    "as_make_meta": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p]),
    "as_conv_gemm_f32": (c_i, [ctypes.POINTER(ConvGemmArgs), c_p]),
    "as_conv_gemm_workspace_bytes": (c_sz, [ctypes.POINTER(ConvGemmArgs)]),
    "as_split_f16x2_bytes": (c_sz, [c_i, c_i]),
    "as_split_f16x2_f32": (c_i, [c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_p]),
    "as_prep_weight_f16x2_bytes": (c_sz, [c_i, c_i, c_i, c_i]),
    "as_prep_weight_f16x2_host": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    "as_embed_groups_f32": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_i, c_p]),
    "as_channel_layernorm_groups_f32": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_f, c_i, c_p, c_i, c_p]),
    "as_channel_layernorm_split_f32": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_f, c_i, c_p, c_p]),
    "as_relpos_attention_groups_f32": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_i, c_p, c_i, c_p]),
    "as_embed_f32": (c_i, [c_p, c_p, c_i, c_i, c_i, c_f, c_p, c_i, c_p]),
    "as_channel_layernorm_f32": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_f, c_i, c_p, c_i, c_p]),
    "as_adain_f32": (c_i, [c_p, c_i, c_i, c_p, c_i, c_p, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_i, c_p]),
    "as_adain_split_f32": (c_i, [c_p, c_i, c_i, c_p, c_i, c_p, c_i, c_i, c_i, c_p, c_p]),
    "as_linear_rows_f32": (c_i, [c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_p, c_i, c_p]),
    "as_durations_f32": (c_i, [c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_p]),
    "as_expand_f32": (c_i, [c_p, c_i, c_i, c_p, c_i, c_i, c_p, c_i, c_p]),
    "as_ref_features_f32": (c_i, [c_p, c_i, c_i, c_p, c_p, c_i, c_i, c_p, c_p, c_i, c_p]),
    "as_crop_f32": (c_i, [c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_i, c_i, c_i, c_p]),
    "as_rows_to_images_f32": (c_i, [c_p, c_i, c_p, c_i, c_i, c_p, c_p, c_i, c_i, c_p]),
    "as_dwconv_down_f32": (c_i, [c_p, c_i, c_p, c_p, c_i, c_p, c_i, c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "as_frame_signal_f32": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p]),
    "as_spec_power_f32": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_p]),
    "as_log_norm_f32": (c_i, [c_p, c_i, c_i, c_i, c_f, c_f, c_f, c_p, c_i, c_p]),
    "as_xl_attention_f32": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_f, c_p, c_i, c_i, c_p, c_i, c_p]),
    "as_glu_dwconv_bn_swish_f32": (c_i, [c_p, c_i, c_i, c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_i, c_p]),
    "as_lstm_step0_f32": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_p]),
    "as_bn_lrelu_maxpool_rows_f32": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_f, c_p, c_i, c_i, c_i, c_p]),
    "as_avgpool_down_f32": (c_i, [c_p, c_i, c_p, c_p, c_i, c_p, c_i, c_p, c_p, c_i, c_i, c_p, c_i, c_i, c_i, c_i, c_p]),
    "as_im2col_valid_f32": (c_i, [c_p, c_i, c_p, c_p, c_i, c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "as_mean_pool_f32": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_p, c_i, c_p]),
    "as_relpos_attention_f32": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_i, c_p, c_i, c_p]),
    "as_bilstm_f32": (c_i, [c_p, c_i, c_p, c_i, c_i, c_p]),
    "as_interleave_phases_f32": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_p, c_i, c_p]),
    "as_mean3_f32": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_i, c_p]),
})


AS_MAX_LSTM_JOBS = 4


class BiLstmJob(ctypes.Structure):
    _fields_ = [("gx_tm", ctypes.c_void_p), ("whh_t", ctypes.c_void_p), ("out", ctypes.c_void_p),
                ("ldg", ctypes.c_int32), ("ldo", ctypes.c_int32)]


class HipLibraryError(RuntimeError):
    pass


def lib():
    """The loaded library (raises HipLibraryError if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                f"{LIB_PATH} not found: build it with `python -m artspeech_amd._build` "
                "(or __graft_entry__.build()).  There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError if the header and the .so disagree
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        kind = "invalid argument" if rc < 0 else f"hipError_t {rc}"
        raise HipLibraryError(f"{what} failed: {kind}")


def ptr(t):
    """Device pointer of a contiguous CUDA(HIP) tensor, or NULL for None."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HipLibraryError("expected a tensor in GPU memory (the HIP path has no CPU fallback)")
    if not t.is_contiguous():
        raise HipLibraryError("expected a contiguous tensor")
    return t.data_ptr()


def stream():
    """the current HIP stream of the current device.  Tensors of another device are rejected by device_guard (models.py runs
    every forward under `torch.cuda.device(model device)`), so a launch never mixes a stream of one GPU with memory of another."""
    return torch.cuda.current_stream().cuda_stream

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
    "as_device_status": (c_i, [c_i]),
    "as_device_status_raise_for_test": (c_i, [c_i, c_p]),
    "as_set_range_probe": (c_i, [c_i]),
    "as_prof_enable": (c_i, [c_i]),
    "as_prof_collect": (c_i, [c_p, c_p, c_p, c_p, c_i]),
    "as_prof_hint": (c_i, [ctypes.c_double, ctypes.c_double]),
    "as_prof_bracket_overhead": (c_i, [c_p, c_p]),
    "as_prof_mfma_sustained": (c_i, [c_p, c_p, c_p]),
    "as_mas_workspace_bytes": (c_sz, [c_i, c_i, c_i]),
    "as_mas_f32": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_sz, c_p]),
    "as_softmax_mas_f32": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_sz, c_p]),
}

AS_MAX_TAPS = 25
AS_ABI_VERSION = 8          # include/artspeech_hip.h


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
                ("n_groups", ctypes.c_int32), ("group_cols", ctypes.c_int32), ("range_probe", ctypes.c_int32), ("status", ctypes.c_void_p),
                ("Xh2", ctypes.c_void_p), ("K2", ctypes.c_int32), ("src_col", ctypes.c_void_p), ("N_in", ctypes.c_int32), ("ileave_u", ctypes.c_int32),
                ("slab_tr", ctypes.c_int32), ("n_valid", ctypes.c_void_p)]


_SIGNATURES.update({
    "as_make_meta": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p]),
    "as_conv_gemm_f32": (c_i, [ctypes.POINTER(ConvGemmArgs), c_p]),
    "as_conv_gemm_workspace_bytes": (c_sz, [ctypes.POINTER(ConvGemmArgs)]),
    "as_conv_gemm_multi_f32": (c_i, [ctypes.POINTER(ConvGemmArgs), c_i, c_p]),
    "as_conv_gemm_multi_tile": (c_i, [ctypes.POINTER(ConvGemmArgs), c_i]),
    "as_conv_gemm_multi_workspace_bytes": (c_sz, [ctypes.POINTER(ConvGemmArgs)]),
    "as_conv_gemm_plan": (c_i, [ctypes.POINTER(ConvGemmArgs), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32),
                                ctypes.POINTER(ctypes.c_int32)]),
    "as_split_f16x2_bytes": (c_sz, [c_i, c_i]),
    "as_split_f16x2_f32": (c_i, [c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_p]),
    "as_prep_weight_f16x2_bytes": (c_sz, [c_i, c_i, c_i, c_i]),
    "as_prep_weight_f16x2_host": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    "as_prep_weight_f16x2_sc_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_i]),
    "as_prep_weight_f16x2_sc_host": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    "as_embed_groups_f32": (c_i, [c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_i, c_p]),
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
    "as_stem_pool_image_f32": (c_i, [c_p, c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p]),
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
    "as_dwconv_down_image_f32": (c_i, [c_p, c_i, c_p, c_p, c_i, c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_p]),
    "as_avgpool_down_image_f32": (c_i, [c_p, c_i, c_p, c_p, c_i, c_p, c_i, c_p, c_p, c_i, c_i, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p]),
    "as_im2col_valid_image_f32": (c_i, [c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    "as_mean_pool_f32": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_p, c_i, c_p]),
    "as_relpos_attention_f32": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_i, c_p, c_i, c_p]),
    "as_relpos_attention_image_f32": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_i, c_p, c_i, c_p, c_p]),
    "as_bilstm_f32": (c_i, [c_p, c_i, c_p, c_i, c_i, c_p]),
    "as_bilstm_cluster_bytes": (ctypes.c_size_t, [c_i, c_i]),
    "as_bilstm_cluster_f32": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_p, ctypes.c_size_t, c_p]),
    "as_interleave_phases_f32": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_p, c_i, c_p]),
    "as_mean3_f32": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_i, c_p]),
})


class LnArgs(ctypes.Structure):
    _fields_ = [("gamma", c_p), ("beta", c_p), ("gamma2", c_p), ("beta2", c_p), ("n_split", ctypes.c_int32), ("eps", ctypes.c_float),
                ("relu", ctypes.c_int32), ("yh", c_p)]


class AdainArgs(ctypes.Structure):
    _fields_ = [("x", c_p), ("ldx", ctypes.c_int32), ("C", ctypes.c_int32), ("gb", c_p), ("gb_off", c_p), ("ldgb", ctypes.c_int32),
                ("gb_sc", ctypes.c_int32), ("col_off", c_p), ("src_off", c_p), ("U", ctypes.c_int32), ("N", ctypes.c_int32),
                ("lrelu", ctypes.c_int32), ("yh", c_p), ("pool_w", c_p), ("pool_b", c_p), ("x_up", c_p), ("ld_up", ctypes.c_int32),
                ("col_w", c_p)]


class DownArgs(ctypes.Structure):
    """AsDownArgs (include/artspeech_hip.h): one tower down-sampling step of as_down_multi_f32"""
    _fields_ = [("kind", ctypes.c_int32), ("x", c_p), ("ldx", ctypes.c_int32), ("in_off", c_p), ("in_w", c_p), ("Hin", ctypes.c_int32),
                ("y", c_p), ("ldy", ctypes.c_int32), ("out_off", c_p), ("out_w", c_p), ("Hout", ctypes.c_int32), ("w", c_p), ("bias", c_p),
                ("kh", ctypes.c_int32), ("pool_h", ctypes.c_int32), ("Kp", ctypes.c_int32), ("res", c_p), ("ldr", ctypes.c_int32),
                ("B", ctypes.c_int32), ("C", ctypes.c_int32), ("max_out", ctypes.c_int32), ("lrelu", ctypes.c_int32), ("yh", c_p),
                ("n_out", ctypes.c_int32)]


class ResPairArgs(ctypes.Structure):
    """AsResPairArgs (include/artspeech_hip.h): one residual step of the vocoder's ResBlock1 as one launch"""
    _fields_ = [("x", c_p), ("ldx", ctypes.c_int32), ("y", c_p), ("ldy", ctypes.c_int32), ("w1", c_p), ("w2", c_p), ("b1", c_p), ("b2", c_p),
                ("scale1", ctypes.c_float), ("scale2", ctypes.c_float), ("C", ctypes.c_int32), ("N", ctypes.c_int32), ("k", ctypes.c_int32),
                ("dil", ctypes.c_int32), ("slope", ctypes.c_float), ("col_off", c_p), ("B", ctypes.c_int32), ("max_w", ctypes.c_int32),
                ("add1", c_p), ("add2", c_p), ("ld_add", ctypes.c_int32), ("out_div", ctypes.c_float), ("yh", c_p), ("yh_slope", ctypes.c_float),
                ("x_u", ctypes.c_int32), ("x_bias", c_p)]


class ModelCfg(ctypes.Structure):
    _fields_ = [("hidden_dim", ctypes.c_int32), ("dim_in", ctypes.c_int32), ("style_dim", ctypes.c_int32), ("n_mels", ctypes.c_int32),
                ("n_token", ctypes.c_int32), ("reserved", ctypes.c_int32), ("stats", ctypes.c_float * 24)]


class Batch(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("tok_lens", ctypes.POINTER(ctypes.c_int32)), ("ref_lens", ctypes.POINTER(ctypes.c_int32)),
                ("frames", ctypes.POINTER(ctypes.c_int32))]


class ForwardIO(ctypes.Structure):
    _fields_ = [("tokens", c_p), ("mel", c_p), ("ld_mel", ctypes.c_int32), ("f0_raw", c_p), ("ema_raw", c_p), ("ld_ema", ctypes.c_int32),
                ("forced_dur", c_p), ("mel_out", c_p), ("ld_out", ctypes.c_int32), ("duration", c_p), ("dur_i", c_p), ("frame_off", c_p),
                ("style", c_p), ("feat12", c_p), ("ld_feat", ctypes.c_int32), ("t_en", c_p), ("a_en", c_p), ("ld_en", ctypes.c_int32),
                ("F0", c_p), ("N", c_p), ("EMA", c_p), ("ld_pred", ctypes.c_int32), ("frame_cap", ctypes.c_int32), ("segs", c_p)]


class HostIO(ctypes.Structure):                 # as_host_io: HOST pointers (as_lanes_submit_host)
    _fields_ = [("tokens", c_p), ("mel", c_p), ("ld_mel", ctypes.c_int32), ("f0_raw", c_p), ("ema_raw", c_p), ("ld_ema", ctypes.c_int32),
                ("forced_dur", c_p), ("mel_out", c_p), ("ld_out", ctypes.c_int32), ("frame_cap", ctypes.c_int32), ("frame_off", c_p)]


AS_MOD_FORWARD_A, AS_MOD_FORWARD_B, AS_MOD_ENCODER, AS_MOD_STYLE, AS_MOD_DURATION, AS_MOD_ARTS, AS_MOD_DECODER, AS_MOD_FORWARD_B_CAP = range(8)
_pB, _pIO = ctypes.POINTER(Batch), ctypes.POINTER(ForwardIO)
_SIGNATURES.update({
    "as_adain_image_f32": (c_i, [ctypes.POINTER(AdainArgs), c_p]),
    "as_conv_gemm_multi_post_f32": (c_i, [ctypes.POINTER(ConvGemmArgs), ctypes.POINTER(AdainArgs), ctypes.POINTER(ctypes.c_int32),
                                          ctypes.POINTER(LnArgs), c_i, c_p]),
    "as_down_multi_f32": (c_i, [ctypes.POINTER(DownArgs), c_i, c_p]),
    "as_respair_f32": (c_i, [ctypes.POINTER(ResPairArgs), c_p]),
    "as_xl_attention_image_f32": (c_i, [c_p, c_i, c_p, c_p, c_i, c_i, c_i, ctypes.c_float, c_p, c_i, c_i, c_p, c_i, c_p, c_p]),
    "as_mean3_image_f32": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, ctypes.c_float, c_p, c_p]),
    "as_conv_post_f32": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_i, ctypes.c_float, c_i, c_p, c_p, c_p]),
    "as_rows_image_f32": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p]),
    "as_project_cols_f32": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_p, c_i, c_p]),
    "as_pointwise_small_f32": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_p, c_i, c_p, c_p, c_i, c_p, c_p]),
    "as_model_create": (c_i, [c_p, c_sz, ctypes.POINTER(ModelCfg), ctypes.POINTER(c_p)]),
    "as_model_destroy": (c_i, [c_p]),
    "as_plan_create": (c_i, [c_p, ctypes.POINTER(c_p)]),
    "as_plan_destroy": (c_i, [c_p]),
    "as_plan_set_serial": (c_i, [c_p, c_i]),
    "as_plan_set_merge": (c_i, [c_p, c_i]),
    "as_plan_set_timing": (c_i, [c_p, c_i]),
    "as_plan_set_operand_mode": (c_i, [c_p, c_i]),
    "as_plan_phase_ms": (c_i, [c_p, ctypes.POINTER(ctypes.c_float), c_i]),
    "as_plan_set_layout_cap": (c_i, [c_p, c_i]),
    "as_plan_layout_flushes": (c_i, [c_p]),
    "as_plan_layout_count": (c_i, [c_p]),
    "as_plan_reset_layouts": (c_i, [c_p]),
    "as_bilstm_cluster_test_hooks": (c_i, [c_i, c_i]),
    "as_module_workspace_bytes": (c_sz, [c_p, c_p, c_i, _pB]),
    "as_encoder_forward": (c_i, [c_p, c_p, c_i, _pB, c_p, c_p, c_i, c_p, c_sz, c_p]),
    "as_style_forward": (c_i, [c_p, c_p, _pB, c_p, c_i, c_p, c_p, c_i, c_p, c_i, c_p, c_p, c_sz, c_p]),
    "as_duration_forward": (c_i, [c_p, c_p, _pB, c_p, c_p, c_i, c_p, c_p, c_sz, c_p]),
    "as_arts_forward": (c_i, [c_p, c_p, _pB, c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_sz, c_p]),
    "as_decoder_forward": (c_i, [c_p, c_p, _pB, c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_p, c_sz, c_p]),
    "as_forward_test_begin": (c_i, [c_p, c_p, _pB, _pIO, c_p, c_sz, c_p]),
    "as_forward_test_finish": (c_i, [c_p, c_p, _pB, _pIO, c_p, c_sz, c_p, c_sz, c_p]),
    "as_forward_test": (c_i, [c_p, c_p, _pB, _pIO, c_p, c_sz, c_p, c_sz, ctypes.POINTER(ctypes.c_int32), c_p]),
    "as_lanes_create": (c_i, [c_p, c_i, ctypes.POINTER(c_p)]),
    "as_lanes_destroy": (c_i, [c_p]),
    "as_lanes_count": (c_i, [c_p]),
    "as_lanes_next": (c_i, [c_p]),
    "as_lanes_stream": (c_p, [c_p, c_i]),
    "as_lanes_submit": (c_i, [c_p, _pB, _pIO, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "as_lanes_wait": (c_i, [c_p, c_i]),
    "as_lanes_set_coalesce": (c_i, [c_p, c_i]),
    "as_lanes_set_debug": (c_i, [c_p, c_i]),
    "as_lanes_submit_host": (c_i, [c_p, _pB, ctypes.POINTER(HostIO), ctypes.POINTER(ctypes.c_int32)]),
    "as_model_get_cfg": (c_i, [c_p, c_p]),
    "as_lanes_flush": (c_i, [c_p]),
    "as_lanes_merged_calls": (ctypes.c_int64, [c_p, c_i]),
    "as_lanes_set_graph_cap": (c_i, [c_p, c_i]),
    "as_lanes_set_layout_cap": (c_i, [c_p, c_i]),
    "as_lanes_reserve": (c_i, [c_p, c_sz, c_sz]),
    "as_lanes_stats": (c_i, [c_p, c_i, ctypes.POINTER(ctypes.c_int64)]),
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
            if os.environ.get("AS_LIB_PATH") and not hasattr(handle, name):
                continue                        # (an experiment / older build named explicitly: A/B runs of scripts/exp)
            fn = getattr(handle, name)          # AttributeError if the header and the .so disagree
            fn.restype = res
            fn.argtypes = args
        if handle.as_abi_version() != AS_ABI_VERSION and not os.environ.get("AS_LIB_PATH"):
            raise HipLibraryError(f"{LIB_PATH} has ABI version {handle.as_abi_version()}, this binding is written for {AS_ABI_VERSION} "
                                  "(include/artspeech_hip.h): rebuild with `python -m artspeech_amd._build`")
        _lib = handle
    return _lib


AS_EDEVICE = -3
STATUS_NAMES = ("clustered LSTM hand-over timed out", "MAS band hand-over timed out", "token id outside [0, n_token)",
                "non-finite accumulator (an operand beyond fp16's range, or a non-finite input)",
                "a layout the kernels cannot serve: an utterance wider than the column descriptors (AS_META_MAX_W) or than its caller said, "
                "or (as_lanes debug mode) device buffers that changed while their submission was waiting for its group",
                "the predicted durations add up to more frames than the capacity the caller named (as_forward_io.frame_cap)")


def device_status(clear=False):
    """Names of the device-side failures raised on the current device since the last clear (as_device_status)."""
    bits = lib().as_device_status(int(clear))
    return [n for k, n in enumerate(STATUS_NAMES) if bits >> k & 1]


def check(rc, what):
    if rc != 0:
        if rc == AS_EDEVICE:
            raise HipLibraryError(f"{what} failed: a kernel reported {device_status()} (as_device_status); results since then are invalid")
        kind = "an output buffer or workspace is too small" if rc == -2 else "invalid argument" if rc < 0 else f"hipError_t {rc}"
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
    """the current HIP stream of the current device.  ops._p rejects tensors of any other device and models.py / mas.py run every
    call under `torch.cuda.device(model device)`, so a launch never mixes a stream of one GPU with memory of another."""
    return torch.cuda.current_stream().cuda_stream

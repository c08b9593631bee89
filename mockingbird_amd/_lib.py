"""ctypes binding of libmbhip.so (the C ABI declared in include/mbhip.h).

There is NO fallback: if the shared object is missing or fails to load the
import raises, and every call that returns a negative code raises
``MbHipError`` with the library's message.  The oracle under ``oracle/`` is
test infrastructure only and is never imported from here.
"""
import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("MBHIP_LIB", _HERE / "libmbhip.so"))


class MbHipError(RuntimeError):
    pass


MB_GAN_MAX_UPS, MB_GAN_MAX_KERNELS, MB_GAN_MAX_DIL = 8, 4, 4


class ConvArgs(C.Structure):
    _fields_ = [
        ("d_x", C.c_void_p), ("d_wpacked", C.c_void_p), ("d_bias", C.c_void_p),
        ("d_res", C.c_void_p), ("d_post_scale", C.c_void_p), ("d_post_shift", C.c_void_p),
        ("d_y", C.c_void_p),
        ("x_bstride", C.c_longlong), ("y_bstride", C.c_longlong), ("res_bstride", C.c_longlong),
        ("batch", C.c_int), ("c_in", C.c_int), ("c_out", C.c_int), ("t_in", C.c_int),
        ("t_out", C.c_int),
        ("ksize", C.c_int), ("dilation", C.c_int), ("pad", C.c_int), ("up", C.c_int),
        ("in_act", C.c_int), ("in_slope", C.c_float), ("in_scale", C.c_float),
        ("out_act", C.c_int), ("out_scale", C.c_float), ("accumulate", C.c_int),
        ("in_repeat", C.c_int), ("transpose_out", C.c_int), ("d_gate", C.c_void_p),
        ("d_valid", C.c_void_p), ("valid_mul", C.c_int), ("down", C.c_int), ("out_slope", C.c_float),
    ]


class ConvF16Args(C.Structure):
    _fields_ = [
        ("d_x", C.c_void_p), ("d_wpacked", C.c_void_p), ("d_bias", C.c_void_p), ("d_res", C.c_void_p),
        ("d_y", C.c_void_p),
        ("x_bstride", C.c_longlong), ("y_bstride", C.c_longlong), ("res_bstride", C.c_longlong),
        ("batch", C.c_int), ("c_in", C.c_int), ("c_out", C.c_int), ("t_in", C.c_int), ("t_out", C.c_int),
        ("ksize", C.c_int), ("dilation", C.c_int), ("pad", C.c_int), ("up", C.c_int),
        ("in_act", C.c_int), ("in_slope", C.c_float),
        ("out_act", C.c_int), ("out_scale", C.c_float),
        ("accumulate", C.c_int), ("in_repeat", C.c_int), ("y_f32", C.c_int),
        ("d_valid", C.c_void_p), ("valid_mul", C.c_int),
    ]


class ResPairF16Args(C.Structure):
    _fields_ = [
        ("d_x", C.c_void_p), ("d_y", C.c_void_p), ("d_wpacked", C.c_void_p), ("d_b1", C.c_void_p),
        ("d_b2", C.c_void_p),
        ("batch", C.c_int), ("channels", C.c_int), ("t", C.c_int), ("ksize", C.c_int), ("dilation", C.c_int),
        ("slope", C.c_float), ("out_scale", C.c_float), ("accumulate", C.c_int),
        ("d_valid", C.c_void_p), ("valid_mul", C.c_int),
    ]


class ResStageF16Args(C.Structure):
    _fields_ = [
        ("d_x", C.c_void_p), ("d_y", C.c_void_p), ("d_wpacked", C.c_void_p), ("d_bias", C.c_void_p),
        ("batch", C.c_int), ("channels", C.c_int), ("t", C.c_int), ("num_kernels", C.c_int), ("num_dilations", C.c_int),
        ("ksize", C.c_int * 4), ("dilation", (C.c_int * 4) * 4),
        ("slope", C.c_float), ("out_scale", C.c_float), ("accumulate", C.c_int),
        ("d_valid", C.c_void_p), ("valid_mul", C.c_int),
    ]


class Ppg2MelNetConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("bnf_dim", "spk_dim", "enc_dim", "down0", "down1", "num_mels", "postnet_layers",
                                       "postnet_dim", "postnet_ksize")]


class Ppg2MelConfig(C.Structure):
    _fields_ = [
        ("enc_dim", C.c_int), ("num_mels", C.c_int), ("frames_per_step", C.c_int), ("attention_rnn_dim", C.c_int),
        ("decoder_rnn_dim", C.c_int), ("n_prenet", C.c_int), ("prenet_dims", C.c_int * 4),
        ("num_mixtures", C.c_int), ("encoder_down_factor", C.c_int), ("num_decoder_rnn_layer", C.c_int),
        ("concat_context_to_last", C.c_int),
    ]


MB_F32, MB_F16, MB_F64 = 0, 1, 2
MB_PCM16_SNDFILE, MB_PCM16_ENCODE16, MB_PCM16_SAVE_WAV = 0, 1, 2


class GanConfig(C.Structure):
    _fields_ = [
        ("kind", C.c_int), ("num_mels", C.c_int), ("upsample_initial_channel", C.c_int),
        ("num_upsamples", C.c_int),
        ("upsample_rates", C.c_int * MB_GAN_MAX_UPS),
        ("upsample_kernel_sizes", C.c_int * MB_GAN_MAX_UPS),
        ("num_kernels", C.c_int),
        ("resblock_kernel_sizes", C.c_int * MB_GAN_MAX_KERNELS),
        ("num_dilations", C.c_int),
        ("resblock_dilations", (C.c_int * MB_GAN_MAX_DIL) * MB_GAN_MAX_KERNELS),
        ("top_k", C.c_int),
        ("interp_ups", C.c_int),
        ("resblock_type", C.c_int),
    ]


class ResPairSplitArgs(C.Structure):
    _fields_ = [
        ("d_x", C.c_void_p), ("d_y", C.c_void_p), ("d_wpacked", C.c_void_p), ("d_b1", C.c_void_p), ("d_b2", C.c_void_p),
        ("batch", C.c_int), ("channels", C.c_int), ("t", C.c_int), ("ksize", C.c_int), ("dilation", C.c_int),
        ("slope", C.c_float), ("out_scale", C.c_float), ("unscale1", C.c_float), ("unscale2", C.c_float),
        ("accumulate", C.c_int), ("d_valid", C.c_void_p), ("valid_mul", C.c_int),
    ]


class ConvSplitTmArgs(C.Structure):
    _fields_ = [
        ("d_x", C.c_void_p), ("d_y", C.c_void_p), ("d_wpacked", C.c_void_p), ("d_bias", C.c_void_p), ("d_res", C.c_void_p),
        ("batch", C.c_int), ("t", C.c_int), ("c_in", C.c_int), ("c_out", C.c_int),
        ("ksize", C.c_int), ("dilation", C.c_int), ("pad", C.c_int),
        ("in_slope", C.c_float), ("unscale", C.c_float), ("out_scale", C.c_float),
        ("out_act", C.c_int), ("accumulate", C.c_int), ("d_valid", C.c_void_p), ("valid_mul", C.c_int),
        ("d_gate", C.c_void_p), ("d_post_scale", C.c_void_p), ("d_post_shift", C.c_void_p),
        ("x_bstride", C.c_longlong), ("x_row_stride", C.c_int), ("x_split", C.c_int), ("d_ysplit", C.c_void_p),
    ]


class WaveRNNConfig(C.Structure):
    _fields_ = [
        ("rnn_dims", C.c_int), ("fc_dims", C.c_int), ("bits", C.c_int), ("pad", C.c_int),
        ("n_upsample", C.c_int), ("upsample_factors", C.c_int * 4),
        ("feat_dims", C.c_int), ("compute_dims", C.c_int), ("res_out_dims", C.c_int),
        ("res_blocks", C.c_int), ("mode", C.c_int),
    ]


class WaveRNNBatchPlan(C.Structure):
    _fields_ = [("n_utt", C.c_int), ("n_folds", C.c_int), ("seq_len", C.c_int), ("fold_stride", C.c_int),
                ("workspace_bytes", C.c_size_t)]


class WaveRNNPlan(C.Structure):
    _fields_ = [
        ("frames", C.c_int), ("total_len", C.c_int), ("n_folds", C.c_int), ("seq_len", C.c_int),
        ("fold_stride", C.c_int), ("workspace_bytes", C.c_size_t),
    ]


class TacoConfig(C.Structure):
    _fields_ = [
        ("n_mels", C.c_int), ("project_dims", C.c_int), ("decoder_dims", C.c_int),
        ("lstm_dims", C.c_int), ("max_r", C.c_int), ("r", C.c_int),
        ("postnet_dims", C.c_int), ("postnet_K", C.c_int), ("num_highways", C.c_int),
        ("lsa_kernel", C.c_int), ("lsa_filters", C.c_int),
        ("has_encoder", C.c_int), ("num_chars", C.c_int), ("embed_dims", C.c_int), ("encoder_dims", C.c_int),
        ("encoder_K", C.c_int), ("speaker_dims", C.c_int), ("style_dims", C.c_int),
        ("has_gst", C.c_int), ("gst_tokens", C.c_int), ("gst_heads", C.c_int), ("gst_n_convs", C.c_int),
        ("gst_width", C.c_int), ("gst_filters", C.c_int * 8), ("dropout", C.c_float),
    ]


# name -> (restype, argtypes); must list every symbol include/mbhip.h declares
# (tests/test_abi.py checks this table against the header and the .so).
_PP = C.POINTER(C.c_void_p)
SIGNATURES = {
    "mb_last_error": (C.c_char_p, []),
    "mb_abi_version": (C.c_int, []),
    "mb_conv1d_packed_floats": (C.c_size_t, [C.c_int] * 4),
    "mb_conv1d_pack": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_void_p]),
    "mb_conv1d": (C.c_int, [C.POINTER(ConvArgs), C.c_void_p]),
    "mb_conv1d_range_events": (C.c_longlong, [C.c_int]),
    "mb_conv1d_f16_packed_halves": (C.c_size_t, [C.c_int] * 4),
    "mb_conv1d_f16_pack": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p]),
    "mb_conv1d_f16": (C.c_int, [C.POINTER(ConvF16Args), C.c_void_p]),
    "mb_resblock_pair_f16_supported": (C.c_int, [C.c_int] * 3),
    "mb_resblock_pair_f16_packed_halves": (C.c_size_t, [C.c_int] * 2),
    "mb_resblock_pair_f16_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "mb_resblock_pair_f16": (C.c_int, [C.POINTER(ResPairF16Args), C.c_void_p]),
    "mb_resblock_pair_split_supported": (C.c_int, [C.c_int] * 3),
    "mb_resblock_pair_split_packed_halves": (C.c_size_t, [C.c_int] * 2),
    "mb_resblock_pair_split_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mb_resblock_pair_split": (C.c_int, [C.POINTER(ResPairSplitArgs), C.c_void_p]),
    "mb_conv_split_tm_supported": (C.c_int, [C.c_int] * 4),
    "mb_conv_split_tm_packed_halves": (C.c_size_t, [C.c_int] * 3),
    "mb_conv_split_tm_pack": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mb_conv_split_tm": (C.c_int, [C.POINTER(ConvSplitTmArgs), C.c_void_p]),
    "mb_maxpool2_tm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mb_conv_c1_tm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "mb_highway_tm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]),
    "mb_f32_cm_to_tm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mb_f32_tm_to_cm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mb_resblock_stage_f16_supported": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "mb_resblock_stage_f16_efficiency": (C.c_float, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "mb_resblock_stage_f16_packed_halves": (C.c_size_t, [C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "mb_resblock_stage_f16_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "mb_resblock_stage_f16": (C.c_int, [C.POINTER(ResStageF16Args), C.c_void_p]),
    "mb_resblock_stage_f32_supported": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "mb_resblock_stage_f32_efficiency": (C.c_float, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "mb_resblock_stage_f32_packed_halves": (C.c_size_t, [C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "mb_resblock_stage_f32_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "mb_resblock_stage_f32": (C.c_int, [C.POINTER(ResStageF16Args), C.c_void_p]),
    "mb_f32_to_f16_tm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mb_f16_tm_to_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mb_gan_num_weights": (C.c_int, [C.POINTER(GanConfig)]),
    "mb_gan_weight_numel": (C.c_size_t, [C.POINTER(GanConfig), C.c_int]),
    "mb_gan_create": (C.c_int, [C.POINTER(GanConfig), _PP, C.c_int, _PP]),
    "mb_gan_create_ex": (C.c_int, [C.POINTER(GanConfig), _PP, C.c_int, C.c_int, _PP]),
    "mb_gan_dtype": (C.c_int, [C.c_void_p]),
    "mb_gan_destroy": (None, [C.c_void_p]),
    "mb_gan_hop": (C.c_int, [C.c_void_p]),
    "mb_gan_out_samples": (C.c_longlong, [C.c_void_p, C.c_int]),
    "mb_gan_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "mb_gan_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_size_t, C.c_void_p]),
    "mb_gan_forward_ragged": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_size_t, C.c_void_p]),
    "mb_gan_forward_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_size_t, C.c_void_p]),
    "mb_wavernn_num_weights": (C.c_int, [C.POINTER(WaveRNNConfig)]),
    "mb_wavernn_weight_numel": (C.c_size_t, [C.POINTER(WaveRNNConfig), C.c_int]),
    "mb_wavernn_plan_generate_batch": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int,
                                                 C.POINTER(WaveRNNBatchPlan), C.POINTER(C.c_int)]),
    "mb_wavernn_generate_batch": (C.c_int, [C.c_void_p, C.POINTER(WaveRNNBatchPlan), C.POINTER(C.c_int), _PP,
                                            C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mb_wavernn_finish_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "mb_wavernn_finish": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_double, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_void_p,
                                    C.c_size_t, C.c_void_p]),
    "mb_wave_workspace_bytes": (C.c_size_t, []),
    "mb_wave_peak_normalize": (C.c_int, [C.c_void_p, C.c_int, C.c_longlong, C.c_double, C.c_void_p, C.c_size_t,
                                         C.c_void_p]),
    "mb_wave_pack_pcm16": (C.c_int, [C.c_void_p, C.c_int, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_size_t, C.c_void_p]),
    "mb_wave_finish_batch_workspace_bytes": (C.c_size_t, [C.c_int]),
    "mb_wave_finish_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_double, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_size_t, C.c_void_p]),
    "mb_wavernn_create": (C.c_int, [C.POINTER(WaveRNNConfig), _PP, C.c_int, _PP]),
    "mb_wavernn_destroy": (None, [C.c_void_p]),
    "mb_wavernn_plan_generate": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.POINTER(WaveRNNPlan)]),
    "mb_wavernn_generate": (C.c_int, [C.c_void_p, C.POINTER(WaveRNNPlan), C.c_void_p, C.c_void_p,
                                      C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_size_t, C.c_void_p]),
    "mb_wavernn_loop_path": (C.c_int, [C.c_int] * 7),
    "mb_diag_lookup": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int]),
    "mb_wavernn_debug_noise": (C.c_int, [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mb_wavernn_debug_noise_mol": (C.c_int, [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mb_wavernn_last_loop_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "mb_wavernn_last_path": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mb_wavernn_bench_kernel": (C.c_int, [C.c_void_p, C.POINTER(WaveRNNPlan), C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_float),
                                          C.POINTER(C.c_double), C.c_void_p]),
    "mb_ppg2mel_net_num_weights": (C.c_int, [C.POINTER(Ppg2MelNetConfig)]),
    "mb_ppg2mel_net_weight_numel": (C.c_size_t, [C.POINTER(Ppg2MelNetConfig), C.c_int]),
    "mb_ppg2mel_net_create": (C.c_int, [C.POINTER(Ppg2MelNetConfig), _PP, C.c_int, _PP]),
    "mb_ppg2mel_net_destroy": (None, [C.c_void_p]),
    "mb_ppg2mel_net_t_enc": (C.c_int, [C.POINTER(Ppg2MelNetConfig), C.c_int]),
    "mb_ppg2mel_net_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "mb_ppg2mel_net_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_size_t, C.c_void_p]),
    "mb_ppg2mel_net_postnet": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t,
                                         C.c_void_p]),
    "mb_ppg2mel_num_weights": (C.c_int, [C.POINTER(Ppg2MelConfig)]),
    "mb_ppg2mel_weight_numel": (C.c_size_t, [C.POINTER(Ppg2MelConfig), C.c_int]),
    "mb_ppg2mel_create": (C.c_int, [C.POINTER(Ppg2MelConfig), _PP, C.c_int, _PP]),
    "mb_ppg2mel_destroy": (None, [C.c_void_p]),
    "mb_ppg2mel_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "mb_ppg2mel_last_loop_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "mb_ppg2mel_last_loop_launches": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "mb_ppg2mel_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                    C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p,
                                    C.c_size_t, C.c_void_p]),
    "mb_taco_num_weights": (C.c_int, [C.POINTER(TacoConfig)]),
    "mb_taco_weight_numel": (C.c_size_t, [C.POINTER(TacoConfig), C.c_int]),
    "mb_taco_create": (C.c_int, [C.POINTER(TacoConfig), _PP, C.c_int, _PP]),
    "mb_taco_destroy": (None, [C.c_void_p]),
    "mb_taco_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "mb_taco_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                 C.c_int, C.c_float, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.c_size_t, C.c_void_p]),
    "mb_taco_last_postnet_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "mb_taco_last_loop_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "mb_taco_last_loop_form": (C.c_int, [C.c_void_p]),
    "mb_taco_last_loop_f16": (C.c_int, [C.c_void_p]),
    "mb_taco_loop_form": (C.c_int, [C.c_int] * 9 + [C.POINTER(C.c_int)]),
    "mb_taco_encode_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "mb_taco_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                 C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mb_maximum_path": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                  C.c_int, C.c_void_p]),
}

_lib = None


ABI_VERSION = 4  # include/mbhip.h: MB_ABI_VERSION


def lib():
    """Load (once) and return the ctypes library; raises if it is not built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise MbHipError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or python -m mockingbird_amd.build). "
                "There is no CPU fallback.")
        # torch first: its bundled HIP runtime must be the one this process binds (libmbhip.so only
        # borrows device memory / streams from it); loading ours first can leave two runtimes around.
        import torch  # noqa: F401
        l = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if l.mb_abi_version() != ABI_VERSION:  # a stale .so: argument meanings / packed images differ (include/mbhip.h)
            raise MbHipError(f"{LIB_PATH} answers ABI version {l.mb_abi_version()}, this package needs {ABI_VERSION}: rebuild it "
                             "(python -m mockingbird_amd.build --force)")
        _lib = l
    return _lib


def check(rc, what=""):
    if rc is not None and rc < 0:
        msg = lib().mb_last_error()
        raise MbHipError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
    return rc


def ptr(t):
    """Device/host pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def fresh_seed() -> int:
    """A 62-bit key for the on-device counter RNGs drawn from torch's GLOBAL (CPU) generator: unseeded calls
    vary, and torch.manual_seed(...) -- the toolbox's "random seed" box, hifigan.load_model's manual_seed
    (control/toolbox/__init__.py:240-243, hifigan/inference.py:39) -- makes a run reproducible, as it does for the
    reference, whose F.dropout (pre_net.py:23,26) / Categorical.sample (fatchord_version.py:224-226) consume
    that generator."""
    import torch
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


def host_ptr_array(tensors):
    """(void*[]) over contiguous float32 CPU tensors; keeps them alive via the return."""
    arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr

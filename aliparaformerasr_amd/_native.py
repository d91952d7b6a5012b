"""ctypes binding of libparaformer_hip.so (include/paraformer_hip.h).

The library is the product; this module only declares its entry points.  There is
no CPU fallback: if the shared object is missing, loading fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libparaformer_hip.so")

PF_OK = 0
PF_ERR_INVALID_ARG = -1
PF_ERR_DEVICE = -2
PF_ERR_IO = -3
PF_ERR_FORMAT = -4
PF_ERR_CAPACITY = -5
PF_ERR_UNSUPPORTED = -6
PF_ERR_DISPOSED = -7
PF_ERR_TOKENS = -8
PF_ERR_NULL_SAMPLES = -9
PF_ERR_RECOGNITION = -10


class PfEngineConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("device", C.c_int32),
        ("weights_path", C.c_char_p), ("weights_host", C.c_void_p), ("weights_device", C.c_void_p),
        ("weights_bytes", C.c_int64),
        ("mvn_path", C.c_char_p), ("cmvn_shift", C.POINTER(C.c_float)), ("cmvn_scale", C.POINTER(C.c_float)),
        ("cmvn_dim", C.c_int32),
        ("fs", C.c_int32), ("n_mels", C.c_int32), ("lfr_m", C.c_int32), ("lfr_n", C.c_int32),
        ("snip_edges", C.c_int32), ("dither", C.c_float), ("window", C.c_char_p), ("use_itn", C.c_int32),
        ("frame_length_ms", C.c_int32), ("frame_shift_ms", C.c_int32), ("dither_seed", C.c_int32),
        ("math_mode", C.c_int32), ("reserved", C.c_int32 * 3),
    ]


class PfGemmDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("relu", C.c_int32), ("out_kind", C.c_int32), ("a_blocked", C.c_int32), ("tile_rows", C.c_int32),
        ("scale_cols", C.c_int32), ("scale", C.c_float),
        ("bias", C.POINTER(C.c_float)), ("resid", C.POINTER(C.c_float)), ("add2", C.POINTER(C.c_float)),
    ]


class PfGemmRcDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("M", C.c_int32), ("K", C.c_int32), ("a_blocked", C.c_int32), ("T", C.c_int32),
        ("fsmn_k", C.c_int32),
        ("bias", C.POINTER(C.c_float)), ("resid", C.POINTER(C.c_float)), ("fsmn_v", C.POINTER(C.c_float)),
        ("fsmn_w", C.POINTER(C.c_float)), ("ln_gamma", C.POINTER(C.c_float)), ("ln_beta", C.POINTER(C.c_float)),
        ("short_input", C.c_int32), ("split_k", C.c_int32),
    ]


class PfBatchOut(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("l_cap", C.c_int32), ("logits_cap", C.c_int64), ("cif_peak_cap", C.c_int64),
        ("token_ids", C.POINTER(C.c_int64)), ("token_num", C.POINTER(C.c_int32)),
        ("logits", C.POINTER(C.c_float)), ("cif_peak", C.POINTER(C.c_float)),
        ("L", C.c_int32), ("V", C.c_int32), ("cif_peak_len", C.c_int32), ("reserved", C.c_int32),
    ]


# name -> (restype, argtypes); kept in one table so tests can check that every symbol the
# header declares is exported.
_P = C.POINTER
_vp = C.c_void_p
_f = _P(C.c_float)
_i32 = _P(C.c_int32)
_i64 = _P(C.c_int64)
_cpp = _P(C.c_char_p)
SIGNATURES = {
    "pf_version": (C.c_int, []),
    "pf_last_error": (C.c_char_p, []),
    "pf_engine_create": (C.c_int, [_P(PfEngineConfig), _P(_vp)]),
    "pf_engine_destroy": (None, [_vp]),
    "pf_engine_info": (C.c_int, [_vp, _i32, _i32, _i32, _i32]),
    "pf_frontend_num_frames": (C.c_int, [_vp, C.c_int64, _i32]),
    "pf_frontend": (C.c_int, [_vp, _f, C.c_int64, _f, C.c_int64, _i32]),
    "pf_fbank": (C.c_int, [_vp, _f, C.c_int64, _f, C.c_int64, _i32]),
    "pf_forward_feats": (C.c_int, [_vp, _f, C.c_int32, C.c_int32, _i32, C.c_int32, _P(PfBatchOut)]),
    "pf_model_proj": (C.c_int, [_vp, _P(_f), _i32, C.c_int32, _i32, C.c_int32, _P(PfBatchOut)]),
    "pf_recognize": (C.c_int, [_vp, _P(_f), _i64, C.c_int32, _i32, C.c_int32, _P(PfBatchOut)]),
    "pf_stage_audio": (C.c_int, [_vp, _P(_f), _i64, C.c_int32]),
    "pf_engine_set_hotwords": (C.c_int, [_vp, C.POINTER(C.c_int32), C.c_int32]),
    "pf_run_staged": (C.c_int, [_vp]),
    "pf_host_wav_read": (C.c_int, [C.c_char_p, _f, C.c_int64, _i64, _i32, _i32, C.POINTER(C.c_double)]),
    "pf_host_resample": (C.c_int, [_f, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _f, C.c_int64, _i64]),
    "pf_host_is_audio": (C.c_int, [C.c_char_p, _i32]),
    "pf_group_create": (C.c_int, [_P(PfEngineConfig), _i32, C.c_int32, _P(_vp)]),
    "pf_group_destroy": (None, [_vp]),
    "pf_group_info": (C.c_int, [_vp, _i32, _i32]),
    "pf_group_engine": (_vp, [_vp, C.c_int32]),
    "pf_group_recognize": (C.c_int, [_vp, _P(_f), _i64, C.c_int32, _i32, C.c_int32, _P(PfBatchOut)]),
    "pf_group_fetch": (C.c_int, [_vp, _P(PfBatchOut)]),
    "pf_sync": (C.c_int, [_vp]),
    "pf_fetch": (C.c_int, [_vp, _P(PfBatchOut)]),
    "pf_fetch_ids_device": (C.c_int, [_vp, _vp, C.c_int32, _i32]),
    "pf_host_group_sim": (C.c_int, [C.c_int32, C.c_int32, _i32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _i64,
                                    C.c_int32, _i32, _i32]),
    "pf_profile_enable": (C.c_int, [_vp, C.c_int32]),
    "pf_profile_reset": (C.c_int, [_vp]),
    "pf_profile_select": (C.c_int, [_vp, C.c_char_p]),
    "pf_profile_get": (C.c_int, [_vp, C.c_char_p, _P(C.c_double), _i64, _P(C.c_double)]),
    "pf_profile_kernel": (C.c_int, [_vp, C.c_char_p, C.c_char_p, C.c_int32]),
    "pf_last_flops": (C.c_int, [_vp, _P(C.c_double)]),
    "pf_op_lfr_cmvn_pad": (C.c_int, [_vp, _P(_f), _i32, C.c_int32, C.c_int32, _f, C.c_int64, _i32]),
    "pf_op_qlinear": (C.c_int, [_vp, _f, _f, _f, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f, _P(C.c_uint8), _f,
                                _P(C.c_uint8), _f, _i32]),
    "pf_op_argmax": (C.c_int, [_vp, _f, C.c_int64, C.c_int32, _i64]),
    "pf_op_gemm": (C.c_int, [_vp, _f, _f, _f, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f]),
    "pf_op_gemm_ex": (C.c_int, [_vp, _P(PfGemmDesc), _f, _f, _f]),
    "pf_op_gemm_rc": (C.c_int, [_vp, _P(PfGemmRcDesc), _f, _f, _f, _f, _f]),
    "pf_op_ffn": (C.c_int, [_vp, _f, _f, _f, _f, _f, _f, C.c_int32, C.c_int32, C.c_int32, _f]),
    "pf_op_ffn_fused": (C.c_int, [_vp, _f, _f, _f, _f, _f, _f, _f, _f, C.c_int32, _f, _f]),
    "pf_op_dec_ffn_fused": (C.c_int, [_vp, C.c_void_p, _f, _f, _f]),
    "pf_op_attn_ffn_fused": (C.c_int, [_vp, _vp, _f, _f]),
    "pf_op_fsmn_enc": (C.c_int, [_vp, _f, _f, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f]),
    "pf_op_linear32": (C.c_int, [_vp, _f, _f, _f, _f, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f]),
    "pf_op_ffn32": (C.c_int, [_vp, _f, _f, _f, _f, _f, C.c_int32, C.c_int32, C.c_int32, _f]),
    "pf_op_fsmn_dec": (C.c_int, [_vp, _f, _f, _i32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f]),
    "pf_op_logsoftmax_argmax": (C.c_int, [_vp, _f, C.c_int64, C.c_int32, _f, _i64]),
    "pf_op_layernorm": (C.c_int, [_vp, _f, _f, _f, C.c_int64, C.c_int32, _f]),
    "pf_op_attention": (C.c_int, [_vp, _f, _f, _f, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f]),
    "pf_op_fsmn": (C.c_int, [_vp, _f, _f, _f, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f]),
    "pf_op_qkv_attention": (C.c_int, [_vp, _f, _f, _f, C.c_int32, C.c_int32, C.c_int32, _f, _f, _f, _f]),
    "pf_op_cif": (C.c_int, [_vp, _f, _f, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, _f, _i32, _i32, _i32]),
    "pf_op_encoder": (C.c_int, [_vp, _f, C.c_int32, C.c_int32, _f]),
    "pf_recognizer_create": (C.c_int, [C.c_char_p] * 6 + [C.c_int32, C.c_int32, C.c_int32, _P(_vp)]),
    "pf_recognizer_dispose": (None, [_vp]),
    "pf_recognizer_free": (None, [_vp]),
    "pf_recognizer_engine": (_vp, [_vp]),
    "pf_recognizer_create_stream": (C.c_int, [_vp, _P(_vp)]),
    "pf_stream_add_samples": (C.c_int, [_vp, _f, C.c_int64]),
    "pf_stream_set_hotwords": (C.c_int, [_vp, _i32, _i32, C.c_int32]),
    "pf_stream_get_hotwords": (C.c_int, [_vp, _i32, C.c_int32, _i32, C.c_int32, _i32]),
    "pf_stream_num_feature_floats": (C.c_int, [_vp, _i32]),
    "pf_stream_dispose": (None, [_vp]),
    "pf_stream_free": (None, [_vp]),
    "pf_recognizer_num_engines": (C.c_int, [_vp]),
    "pf_recognizer_get_results": (C.c_int, [_vp, _P(_vp), C.c_int32]),
    "pf_result_text": (C.c_int, [_vp, C.c_int32, _cpp, _i32]),
    "pf_result_num_tokens": (C.c_int, [_vp, C.c_int32, _i32]),
    "pf_result_token": (C.c_int, [_vp, C.c_int32, C.c_int32, _cpp]),
    "pf_result_timestamp": (C.c_int, [_vp, C.c_int32, C.c_int32, _P(_i32), _i32]),
    "pf_result_num_timestamps": (C.c_int, [_vp, C.c_int32, _i32]),
    "pf_stream_tokens": (C.c_int, [_vp, _P(_i64), _i32]),
    "pf_stream_create": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_char_p, _P(_vp)]),
    "pf_stream_set_tokens": (C.c_int, [_vp, _i64, C.c_int32]),
    "pf_stream_num_timestamps": (C.c_int, [_vp, _i32]),
    "pf_stream_timestamp": (C.c_int, [_vp, C.c_int32, _P(_i32), _i32]),
    "pf_stream_set_timestamps": (C.c_int, [_vp, _i32, _i32, C.c_int32]),
    "pf_stream_get_speech": (C.c_int, [_vp, _f, C.c_int64, _i32]),
    "pf_stream_set_speech": (C.c_int, [_vp, _f, C.c_int32, C.c_int32]),
    "pf_online_recognizer_create": (C.c_int, [C.c_char_p] * 5 + [C.c_int32, C.c_int32, _P(_vp)]),
    "pf_online_recognizer_dispose": (None, [_vp]),
    "pf_online_recognizer_free": (None, [_vp]),
    "pf_online_recognizer_engine": (_vp, [_vp]),
    "pf_online_create_stream": (C.c_int, [_vp, _P(_vp)]),
    "pf_online_stream_add_samples": (C.c_int, [_vp, _f, C.c_int64]),
    "pf_online_get_results": (C.c_int, [_vp, _P(_vp), C.c_int32]),
    "pf_online_result_text": (C.c_int, [_vp, C.c_int32, _cpp]),
    "pf_online_stream_tokens": (C.c_int, [_vp, _P(_i64), _i32]),
    "pf_online_stream_dispose": (None, [_vp]),
    "pf_online_stream_free": (None, [_vp]),
    "pf_online_encoder": (C.c_int, [_vp, _f, C.c_int32, C.c_int32, _f, _f]),
    "pf_online_decoder": (C.c_int, [_vp, _f, C.c_int32, C.c_int32, _f, C.c_int32, _i32, _f, C.c_int32, _f, _i64, _f]),
    "pf_host_online_lfr": (C.c_int, [_f, C.c_int32, C.c_int32, C.c_int32, _f, C.c_int64, _i32]),
    "pf_host_online_posenc": (C.c_int, [_f, C.c_int32, C.c_int32, C.c_int32]),
    "pf_host_online_dynamic_mask": (C.c_int, [_f, C.c_int32]),
    "pf_host_online_cif": (C.c_int, [_f, _f, C.c_int32, C.c_int32, C.c_float, _f, C.c_int32, _i32, _f, _f]),
    "pf_host_online_decode": (C.c_int, [_cpp, C.c_int32, _i64, C.c_int32, C.c_char_p, C.c_int32]),
    "pf_host_timestamps": (C.c_int, [_f, C.c_int32, _i64, C.c_int32, _i32, C.c_int32]),
    "pf_host_hotword_ids": (C.c_int, [_cpp, C.c_int32, _cpp, C.c_int32, _i32, C.c_int32, _i32, C.c_int32, _i32]),
    "pf_host_decode": (C.c_int, [_cpp, C.c_int32, _i64, C.c_int32, _i32, _i32, C.c_int32, _P(_vp)]),
    "pf_decoded_text": (C.c_int, [_vp, _cpp, _i32]),
    "pf_decoded_num_tokens": (C.c_int, [_vp, _i32]),
    "pf_decoded_token": (C.c_int, [_vp, C.c_int32, _cpp]),
    "pf_decoded_num_timestamps": (C.c_int, [_vp, _i32]),
    "pf_decoded_timestamp": (C.c_int, [_vp, C.c_int32, _P(_i32), _i32]),
    "pf_decoded_free": (None, [_vp]),
}

_lib = None


class NativeLibraryMissing(ImportError):
    pass


def load() -> C.CDLL:
    """Loads libparaformer_hip.so (built in-tree by __graft_entry__.build() / csrc/Makefile).
    Raises NativeLibraryMissing when it is absent — there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryMissing(
            f"{LIB_PATH} not found: build it with `make -C aliparaformerasr_amd/csrc` "
            "(or __graft_entry__.build()); this package has no CPU fallback")
    try:
        # If torch is (or will be) in the process, its bundled libamdhip64 must be the one
        # HIP runtime instance: import it first so both share one runtime.
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class PfAttnFfnDesc(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("M", C.c_int32), ("T", C.c_int32), ("reserved", C.c_int32)] + \
               [(n, C.POINTER(C.c_float)) for n in ("ctx", "wo", "bo", "v", "fsmn_w", "ln2_gamma", "ln2_beta", "resid",
                                                    "w1", "b1", "w2", "b2", "ln_gamma", "ln_beta",
                                                    "wqkv", "bqkv", "q_out", "k_out", "v_out")]


class PfDecFfnDesc(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("M", C.c_int32), ("splits", C.c_int32), ("reserved", C.c_int32)] + \
               [(n, C.POINTER(C.c_float)) for n in ("x", "w1", "b1", "gamma_f", "beta_f", "w2", "ln_gamma", "ln_beta",
                                                    "ctx", "wo", "bo", "resid", "ln1_gamma", "ln1_beta")]


class PfError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"[{code}] {message}")
        self.code = code
        self.message = message


def check(code: int) -> int:
    if code < 0:
        msg = load().pf_last_error()
        raise PfError(code, msg.decode("utf-8", "replace") if msg else "")
    return code

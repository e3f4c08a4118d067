"""ctypes binding of include/q3asr.h.  There is no fallback: if libq3asr_hip.so is missing the import of
anything that needs it raises, and every engine call fails loudly when no HIP device is present."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# Q3A_LIB: load an A/B build of the same library (qwen3_asr_rs_amd/build.py, Q3A_BUILD_VARIANT) instead
LIB_PATH = os.environ.get("Q3A_LIB") or os.path.join(HERE, "lib", "libq3asr_hip.so")

# every symbol include/q3asr.h declares (tests check the exports against the header text)
SYMBOLS = [
    "q3a_opts_default", "q3a_device_count", "q3a_engine_create", "q3a_arena_bytes", "q3a_arena_pack", "q3a_engine_create_from_arena",
    "q3a_engine_destroy", "q3a_last_error", "q3a_get_dims", "q3a_weights_rounded", "q3a_num_frames", "q3a_num_audio_tokens",
    "q3a_build_prompt", "q3a_mel", "q3a_encode", "q3a_prefill", "q3a_decode_step", "q3a_set_next_tokens",
    "q3a_upload_pcm", "q3a_run_resident", "q3a_fetch_ids", "q3a_transcribe_batch", "q3a_transcribe_batch_ptrs", "q3a_io_timings_last", "q3a_stage_timings",
    "q3a_profile_decode_step", "q3a_profile_weight_stream", "q3a_measure_peaks", "q3a_debug_read", "q3a_debug_set", "q3a_selftest_gemm", "q3a_selftest_gemm16",
    "q3a_load_audio", "q3a_resample", "q3a_resample_rubato", "q3a_free", "q3a_tokenizer_create", "q3a_tokenizer_destroy",
    "q3a_tokenizer_decode", "q3a_tokenizer_encode", "q3a_normalize_nfc", "q3a_parse_asr_output", "q3a_capitalize_first",
    "q3a_group_create", "q3a_group_destroy", "q3a_group_size", "q3a_group_used_rccl", "q3a_group_last_error",
    "q3a_group_engine", "q3a_group_partition", "q3a_group_transcribe", "q3a_group_transcribe_ptrs", "q3a_group_startup_seconds",
]


class Peaks(C.Structure):
    _fields_ = [("hbm_read_gbps", C.c_double), ("hbm_copy_gbps", C.c_double), ("hbm_triad_gbps", C.c_double), ("hbm_read_bytes", C.c_double),
                ("mfma_bf16_tflops", C.c_double), ("gemm_m", C.c_int32), ("gemm_n", C.c_int32), ("gemm_k", C.c_int32), ("n_cu", C.c_int32),
                ("reps", C.c_int32), ("reserved", C.c_int32), ("mfma_bf16_tflops_random", C.c_double)]


class Opts(C.Structure):
    _fields_ = [("precise", C.c_int32), ("max_new_tokens", C.c_int32), ("use_graph", C.c_int32),
                ("debug_taps", C.c_int32), ("valu_attention", C.c_int32), ("reserved", C.c_int32 * 11)]


class DimsC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "enc_d_model", "enc_layers", "enc_heads", "enc_ffn", "num_mel_bins", "n_window", "n_window_infer",
        "conv_channels", "enc_output_dim", "max_source_positions", "vocab_size", "hidden_size",
        "intermediate_size", "dec_layers", "num_q_heads", "num_kv_heads", "head_dim", "tie_word_embeddings",
        "mrope_interleaved")] + [("mrope_section", C.c_int32 * 4), ("rms_norm_eps", C.c_float), ("rope_theta", C.c_double)]


class Timings(C.Structure):
    _fields_ = [("mel_ms", C.c_float), ("encoder_ms", C.c_float), ("prefill_ms", C.c_float), ("decode_ms", C.c_float),
                ("total_ms", C.c_float), ("decode_steps", C.c_int32), ("batch", C.c_int32),
                ("total_audio_tokens", C.c_int32), ("total_prompt_tokens", C.c_int32)]


class IoTimings(C.Structure):
    _fields_ = [("stage_ms", C.c_float), ("h2d_ms", C.c_float), ("wall_ms", C.c_float), ("pieces", C.c_int32), ("threads", C.c_int32),
                ("mode", C.c_int32)]


KC_NAMES = ["gemv_qkv_gateup", "decode_attn", "argmax", "gemm", "norm", "other", "gemv_o_proj", "gemv_down", "gemv_lm_head"]


class KernelProfile(C.Structure):
    _fields_ = [("total_us", C.c_float * 9), ("launches", C.c_int32 * 9), ("weight_bytes", C.c_double * 9)]


_lib = None


def load() -> C.CDLL:
    """Load the shared library (building it is `python -m qwen3_asr_rs_amd.build` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: build it with `python -m qwen3_asr_rs_amd.build` "
                           "(there is no CPU fallback for the HIP engine)")
    lib = C.CDLL(LIB_PATH)
    P, i32, i64, u64, f32p = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.POINTER(C.c_float)
    i32p, i64p, u8p = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_uint8)
    sig = {
        "q3a_opts_default": (None, [C.POINTER(Opts)]),
        "q3a_device_count": (i32, []),
        "q3a_engine_create": (i32, [C.c_char_p, i32, C.POINTER(Opts), C.POINTER(P)]),
        "q3a_arena_bytes": (i32, [C.c_char_p, C.POINTER(u64)]),
        "q3a_arena_pack": (i32, [C.c_char_p, P, u64]),
        "q3a_engine_create_from_arena": (i32, [C.c_char_p, i32, P, u64, C.POINTER(Opts), C.POINTER(P)]),
        "q3a_engine_destroy": (None, [P]),
        "q3a_last_error": (C.c_char_p, [P]),
        "q3a_get_dims": (i32, [P, C.POINTER(DimsC)]),
        "q3a_weights_rounded": (i32, [P]),
        "q3a_num_frames": (i64, [i64]),
        "q3a_num_audio_tokens": (i32, [P, i64]),
        "q3a_build_prompt": (i32, [i32, i32p, i32, i32p, i32p]),
        "q3a_mel": (i32, [P, f32p, i64p, i32, f32p, i32p]),
        "q3a_encode": (i32, [P, f32p, i32p]),
        "q3a_prefill": (i32, [P, i32p, i32p, i32, f32p, i32p]),
        "q3a_decode_step": (i32, [P, i32p, u8p, f32p]),
        "q3a_set_next_tokens": (i32, [P, i32p, i32]),
        "q3a_upload_pcm": (i32, [P, f32p, i64p, i32]),
        "q3a_run_resident": (i32, [P, i32p, i32, i32, i32]),
        "q3a_fetch_ids": (i32, [P, i32p, i32, i32p]),
        "q3a_transcribe_batch": (i32, [P, f32p, i64p, i32, i32p, i32, i32, i32, i32p, i32, i32p]),
        "q3a_transcribe_batch_ptrs": (i32, [P, C.POINTER(P), i64p, i32, i32p, i32, i32, i32, i32p, i32, i32p]),
        "q3a_io_timings_last": (i32, [P, C.POINTER(IoTimings)]),
        "q3a_stage_timings": (i32, [P, C.POINTER(Timings)]),
        "q3a_profile_decode_step": (i32, [P, C.POINTER(KernelProfile)]),
        "q3a_profile_weight_stream": (i32, [P, i32, f32p, C.POINTER(C.c_double), i32p]),
        "q3a_measure_peaks": (i32, [i32, i32, C.POINTER(Peaks)]),
        "q3a_debug_read": (i32, [P, C.c_char_p, P, u64, C.POINTER(u64)]),
        "q3a_debug_set": (i32, [C.c_char_p, i32]),
        "q3a_selftest_gemm": (i32, [i32, i32, i32, i32, i32, f32p, f32p]),
        "q3a_selftest_gemm16": (i32, [i32, i32, i32, i32, i32, f32p, f32p, f32p, f32p]),
        "q3a_load_audio": (i32, [C.c_char_p, i32, C.POINTER(f32p), i64p]),
        "q3a_resample": (i32, [f32p, i64, i32, i32, C.POINTER(f32p), i64p]),
        "q3a_resample_rubato": (i32, [f32p, i64, i32, i32, C.POINTER(f32p), i64p]),
        "q3a_free": (None, [P]),
        "q3a_tokenizer_create": (i32, [C.c_char_p, C.POINTER(P)]),
        "q3a_tokenizer_destroy": (None, [P]),
        "q3a_tokenizer_decode": (i32, [P, i32p, i32, i32, C.c_char_p, i32, i32p]),
        "q3a_tokenizer_encode": (i32, [P, C.c_char_p, i32p, i32, i32p]),
        "q3a_parse_asr_output": (i32, [C.c_char_p, i32, C.c_char_p, i32, C.c_char_p, i32]),
        "q3a_capitalize_first": (i32, [C.c_char_p, C.c_char_p, i32]),
        "q3a_normalize_nfc": (i32, [C.c_char_p, C.c_char_p, i32, i32p]),
        "q3a_group_create": (i32, [C.c_char_p, i32, i32p, C.POINTER(Opts), C.POINTER(P)]),
        "q3a_group_destroy": (None, [P]),
        "q3a_group_size": (i32, [P]),
        "q3a_group_used_rccl": (i32, [P]),
        "q3a_group_startup_seconds": (i32, [P, C.POINTER(C.c_double)]),
        "q3a_group_last_error": (C.c_char_p, [P]),
        "q3a_group_engine": (P, [P, i32]),
        "q3a_group_partition": (None, [i32, i32, i32, i32p, i32p]),
        "q3a_group_transcribe": (i32, [P, f32p, i64p, i32, i32p, i32, i32, i32, i32p, i32, i32p]),
        "q3a_group_transcribe_ptrs": (i32, [P, C.POINTER(P), i64p, i32, i32p, i32, i32, i32, i32p, i32, i32p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib

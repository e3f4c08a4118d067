"""WAV ingest and text glue of the pipeline shell -- thin ctypes wrappers over the C++ host code
(csrc/host_audio.cpp, csrc/host_text.cpp), so the Python tests, the `asr` CLI and a Rust shim all share ONE
implementation.  Reference: src/audio.rs:7 (load_audio), src/tokenizer.rs (AsrTokenizer), src/inference.rs:276-313."""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Sequence, Tuple

import numpy as np

from . import _lib


def _err() -> str:
    return (_lib.load().q3a_last_error(None) or b"").decode()


def _take(ptr, n) -> np.ndarray:
    lib = _lib.load()
    out = np.ctypeslib.as_array(ptr, shape=(max(int(n), 1),))[:int(n)].copy()
    lib.q3a_free(C.cast(ptr, C.c_void_p))
    return out


def load_audio(path: str, target_sr: int = 16000) -> np.ndarray:
    """src/audio.rs:7 load_audio(path, target_sample_rate) -> mono f32 at the target rate (WAV input)."""
    lib = _lib.load()
    p, n = C.POINTER(C.c_float)(), C.c_int64()
    if lib.q3a_load_audio(os.fsencode(path), target_sr, C.byref(p), C.byref(n)) != 0:
        raise RuntimeError(_err())
    return _take(p, n.value)


def resample(x: np.ndarray, sr_in: int, sr_out: int, method: str = "polyphase") -> np.ndarray:
    """method "polyphase": this backend's Kaiser polyphase filter; "rubato": the reference fallback's resampler
    (src/audio.rs:220-245) restated (see include/q3asr.h q3a_resample_rubato)."""
    lib = _lib.load()
    x = np.ascontiguousarray(x, dtype=np.float32)
    p, n = C.POINTER(C.c_float)(), C.c_int64()
    fn = lib.q3a_resample_rubato if method == "rubato" else lib.q3a_resample
    if fn(x.ctypes.data_as(C.POINTER(C.c_float)), len(x), sr_in, sr_out, C.byref(p), C.byref(n)) != 0:
        raise RuntimeError(_err())
    return _take(p, n.value)


class AsrTokenizer:
    """src/tokenizer.rs:4-50 over tokenizer.json (byte-level BPE): decode(ids) with special tokens skipped,
    encode(text) for the ASCII forced-language prompt."""

    def __init__(self, tokenizer_json: str):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        if self._lib.q3a_tokenizer_create(os.fsencode(tokenizer_json), C.byref(self._h)) != 0:
            raise RuntimeError(_err())

    @classmethod
    def from_dir(cls, model_dir: str) -> "AsrTokenizer":
        return cls(os.path.join(model_dir, "tokenizer.json"))

    def decode(self, ids: Sequence[int], skip_special_tokens: bool = True) -> str:
        a = np.asarray(list(ids), dtype=np.int32)
        n = C.c_int32()
        ap = a.ctypes.data_as(C.POINTER(C.c_int32))
        if self._lib.q3a_tokenizer_decode(self._h, ap, len(a), int(skip_special_tokens), None, 0, C.byref(n)) != 0:
            raise RuntimeError(_err())
        buf = C.create_string_buffer(n.value + 1)
        self._lib.q3a_tokenizer_decode(self._h, ap, len(a), int(skip_special_tokens), buf, n.value + 1, C.byref(n))
        return buf.raw[:n.value].decode("utf-8")

    def encode(self, text: str) -> List[int]:
        ids = np.zeros(max(4 * len(text), 16), dtype=np.int32)
        n = C.c_int32()
        if self._lib.q3a_tokenizer_encode(self._h, text.encode("utf-8"), ids.ctypes.data_as(C.POINTER(C.c_int32)), len(ids), C.byref(n)) != 0:
            raise RuntimeError(_err())
        return ids[:n.value].tolist()

    def __del__(self):
        try:
            if self._h.value:
                self._lib.q3a_tokenizer_destroy(self._h)
        except Exception:
            pass


def parse_asr_output(raw: str, language_forced: bool) -> Tuple[str, str]:
    """src/inference.rs:276-305"""
    lib = _lib.load()
    lang = C.create_string_buffer(1024)
    text = C.create_string_buffer(4 * len(raw.encode("utf-8")) + 64)
    if lib.q3a_parse_asr_output(raw.encode("utf-8"), int(language_forced), lang, len(lang), text, len(text)) != 0:
        raise RuntimeError(_err())
    return lang.value.decode("utf-8"), text.value.decode("utf-8")


def capitalize_first(s: str) -> str:
    """src/inference.rs:307-313"""
    lib = _lib.load()
    out = C.create_string_buffer(len(s.encode("utf-8")) + 8)
    lib.q3a_capitalize_first(s.encode("utf-8"), out, len(out))
    return out.value.decode("utf-8")


def normalize_nfc(s: str) -> str:
    """The normaliser step of the tokenizer (Unicode NFC, csrc/host_text.cpp normalize_nfc)."""
    lib = _lib.load()
    raw = s.encode("utf-8")
    n = C.c_int32()
    out = C.create_string_buffer(3 * len(raw) + 16)
    if lib.q3a_normalize_nfc(raw, out, len(out), C.byref(n)) != 0:
        raise RuntimeError(_err())
    return out.raw[:n.value].decode("utf-8")

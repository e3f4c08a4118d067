"""Host-side mirror of the reference's pipeline interface over the C ABI (include/q3asr.h).

Names, argument meaning and error behaviour follow the reference (second-state/qwen3_asr_rs):
    AsrInference.load(model_dir, device)        src/inference.rs:30-86
    AsrInference.transcribe(audio, language)    src/inference.rs:89-213
    WhisperFeatureExtractor.extract(samples)    src/mel.rs:49-96
    AudioEncoder.forward(...)                   src/audio_encoder.rs:79-169
    TextDecoder prefill / greedy step           src/text_decoder.rs:94-113, src/inference.rs:140-200
    parse_asr_output / capitalize_first         src/inference.rs:276-313
This module holds no arithmetic: every number is produced by the HIP kernels behind libq3asr_hip.so.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib

EOS_TOKEN_IDS = (151643, 151645)  # src/inference.rs:154, src/tokenizer.rs:54-55


class Q3aError(RuntimeError):
    pass


def _f32p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i32p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _i64p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


class HipEngine:
    """Thin RAII wrapper of a q3a_engine handle (one per GPU, one host thread per handle)."""

    def __init__(self, model_dir: str, device: int = 0, precise: bool = False, max_new_tokens: int = 4096,
                 use_graph: bool = True, debug_taps: bool = False, device_arena: Optional[Tuple[int, int]] = None,
                 valu_attention: bool = False):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        opts = _lib.Opts()
        self._lib.q3a_opts_default(C.byref(opts))
        opts.precise = int(precise)
        opts.max_new_tokens = int(max_new_tokens)
        opts.use_graph = int(use_graph)
        opts.debug_taps = int(debug_taps)
        opts.valu_attention = int(valu_attention)
        md = os.fsencode(model_dir)
        if device_arena is None:
            rc = self._lib.q3a_engine_create(md, device, C.byref(opts), C.byref(self._h))
        else:
            ptr, nbytes = device_arena
            rc = self._lib.q3a_engine_create_from_arena(md, device, C.c_void_p(ptr), nbytes, C.byref(opts), C.byref(self._h))
        if rc != 0:
            raise Q3aError((self._lib.q3a_last_error(None) or b"").decode())
        d = _lib.DimsC()
        self._lib.q3a_get_dims(self._h, C.byref(d))
        self.dims = d
        self.model_dir = model_dir
        self.batch = 0
        self._n_frames: List[int] = []
        self._T: List[int] = []

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.q3a_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc: int):
        if rc != 0:
            raise Q3aError((self._lib.q3a_last_error(self._h) or b"").decode())

    # ---- helpers ------------------------------------------------------------------------------------
    @staticmethod
    def _concat(clips: Sequence[np.ndarray]):
        ns = np.array([len(c) for c in clips], dtype=np.int64)
        pcm = np.ascontiguousarray(np.concatenate([np.asarray(c, dtype=np.float32) for c in clips]))
        return pcm, ns

    def num_audio_tokens(self, n_samples: int) -> int:
        return int(self._lib.q3a_num_audio_tokens(self._h, self._lib.q3a_num_frames(int(n_samples))))

    @staticmethod
    def build_prompt(num_audio_tokens: int, lang_prefix_ids: Optional[Sequence[int]] = None) -> np.ndarray:
        lib = _lib.load()
        n = C.c_int32()
        pre = np.asarray(lang_prefix_ids if lang_prefix_ids is not None else [], dtype=np.int32)
        prep = _i32p(pre) if len(pre) else None
        lib.q3a_build_prompt(num_audio_tokens, prep, len(pre), None, C.byref(n))
        ids = np.zeros(n.value, dtype=np.int32)
        lib.q3a_build_prompt(num_audio_tokens, prep, len(pre), _i32p(ids), C.byref(n))
        return ids

    # ---- stage API ----------------------------------------------------------------------------------
    def mel(self, clips: Sequence[np.ndarray]) -> List[np.ndarray]:
        pcm, ns = self._concat(clips)
        B = len(clips)
        nf = np.zeros(B, dtype=np.int32)
        frames = [int(self._lib.q3a_num_frames(int(n))) for n in ns]
        out = np.zeros(int(sum(frames)) * self.dims.num_mel_bins, dtype=np.float32)
        self._chk(self._lib.q3a_mel(self._h, _f32p(pcm), _i64p(ns), B, _f32p(out), _i32p(nf)))
        self.batch, self._n_frames = B, [int(x) for x in nf]
        res, off = [], 0
        for f in self._n_frames:
            res.append(out[off:off + f * self.dims.num_mel_bins].reshape(self.dims.num_mel_bins, f).copy())
            off += f * self.dims.num_mel_bins
        return res

    def encode(self) -> List[np.ndarray]:
        B = self.batch
        T = np.zeros(B, dtype=np.int32)
        total = sum(int(self._lib.q3a_num_audio_tokens(self._h, f)) for f in self._n_frames)
        out = np.zeros((total, self.dims.enc_output_dim), dtype=np.float32)
        self._chk(self._lib.q3a_encode(self._h, _f32p(out), _i32p(T)))
        self._T = [int(t) for t in T]
        res, off = [], 0
        for t in self._T:
            res.append(out[off:off + t].copy())
            off += t
        return res

    def prefill(self, prompts: Sequence[Sequence[int]], want_logits: bool = True):
        B = len(prompts)
        lens = np.array([len(p) for p in prompts], dtype=np.int32)
        ids = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.int32) for p in prompts]))
        logits = np.zeros((B, self.dims.vocab_size), dtype=np.float32) if want_logits else None
        nxt = np.zeros(B, dtype=np.int32)
        self._chk(self._lib.q3a_prefill(self._h, _i32p(ids), _i32p(lens), B, _f32p(logits) if want_logits else None, _i32p(nxt)))
        return logits, nxt

    def decode_step(self, want_logits: bool = True):
        B = self.batch
        logits = np.zeros((B, self.dims.vocab_size), dtype=np.float32) if want_logits else None
        nxt = np.zeros(B, dtype=np.int32)
        done = np.zeros(B, dtype=np.uint8)
        self._chk(self._lib.q3a_decode_step(self._h, _i32p(nxt), done.ctypes.data_as(C.POINTER(C.c_uint8)),
                                            _f32p(logits) if want_logits else None))
        return logits, nxt, done

    def set_next_tokens(self, ids: Sequence[int]):
        a = np.asarray(ids, dtype=np.int32)
        self._chk(self._lib.q3a_set_next_tokens(self._h, _i32p(a), len(a)))

    # ---- whole path ---------------------------------------------------------------------------------
    def upload_pcm(self, clips: Sequence[np.ndarray]):
        pcm, ns = self._concat(clips)
        self._chk(self._lib.q3a_upload_pcm(self._h, _f32p(pcm), _i64p(ns), len(clips)))
        self.batch = len(clips)

    def run_resident(self, lang_prefix_ids: Optional[Sequence[int]] = None, max_new: int = 0, fixed_new_tokens: int = 0):
        pre = np.asarray(lang_prefix_ids if lang_prefix_ids is not None else [], dtype=np.int32)
        self._chk(self._lib.q3a_run_resident(self._h, _i32p(pre) if len(pre) else None, len(pre), max_new, fixed_new_tokens))

    def fetch_ids(self, stride: int) -> List[List[int]]:
        out = np.zeros((self.batch, stride), dtype=np.int32)
        lens = np.zeros(self.batch, dtype=np.int32)
        self._chk(self._lib.q3a_fetch_ids(self._h, _i32p(out), stride, _i32p(lens)))
        return [out[b, :min(int(lens[b]), stride)].tolist() for b in range(self.batch)]

    @staticmethod
    def _ptrs(clips: Sequence[np.ndarray]):
        """One pointer per utterance (q3a_transcribe_batch_ptrs): float32 C-contiguous arrays are passed as they are -- no
        concatenation, no copy on the Python side.  Returns (kept-alive arrays, void* array, int64 lengths)."""
        arrs = [np.ascontiguousarray(c, dtype=np.float32) for c in clips]
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        ns = np.array([a.size for a in arrs], dtype=np.int64)
        return arrs, ptrs, ns

    def transcribe_batch(self, clips: Sequence[np.ndarray], lang_prefix_ids: Optional[Sequence[int]] = None,
                         max_new: int = 4096, fixed_new_tokens: int = 0) -> List[List[int]]:
        """AsrInference::transcribe steps 2-8 for a batch, host PCM -> ids on the host, ONE call through the C ABI."""
        arrs, ptrs, ns = self._ptrs(clips)
        B = len(arrs)
        stride = fixed_new_tokens if fixed_new_tokens > 0 else max_new
        out = np.zeros((B, stride), dtype=np.int32)
        lens = np.zeros(B, dtype=np.int32)
        pre = np.asarray(lang_prefix_ids if lang_prefix_ids is not None else [], dtype=np.int32)
        self._chk(self._lib.q3a_transcribe_batch_ptrs(self._h, ptrs, _i64p(ns), B, _i32p(pre) if len(pre) else None, len(pre), max_new,
                                                      fixed_new_tokens, _i32p(out), stride, _i32p(lens)))
        self.batch = B
        return [out[b, :min(int(lens[b]), stride)].tolist() for b in range(B)]

    def io_timings(self) -> dict:
        """Input side of the last transcribe_batch (q3a_io_timings_last)."""
        t = _lib.IoTimings()
        self._chk(self._lib.q3a_io_timings_last(self._h, C.byref(t)))
        return {n: getattr(t, n) for n, _ in _lib.IoTimings._fields_}

    # ---- measurement / debug --------------------------------------------------------------------------
    def timings(self) -> dict:
        t = _lib.Timings()
        self._chk(self._lib.q3a_stage_timings(self._h, C.byref(t)))
        return {n: getattr(t, n) for n, _ in _lib.Timings._fields_}

    def profile_decode_step(self) -> dict:
        p = _lib.KernelProfile()
        self._chk(self._lib.q3a_profile_decode_step(self._h, C.byref(p)))
        return {name: {"total_us": float(p.total_us[i]), "launches": int(p.launches[i]), "weight_bytes": float(p.weight_bytes[i])}
                for i, name in enumerate(_lib.KC_NAMES)}

    def profile_weight_stream(self, reps: int = 4) -> dict:
        """Back-to-back qkv + gate/up GEMVs over all layers between one HIP event pair (see q3asr.h)."""
        avg, nbytes, n = C.c_float(), C.c_double(), C.c_int32()
        self._chk(self._lib.q3a_profile_weight_stream(self._h, reps, C.byref(avg), C.byref(nbytes), C.byref(n)))
        return {"avg_us": float(avg.value), "bytes_per_launch": float(nbytes.value), "launches": int(n.value)}

    def debug_read(self, name: str) -> np.ndarray:
        n = C.c_uint64()
        self._chk(self._lib.q3a_debug_read(self._h, name.encode(), None, 0, C.byref(n)))
        out = np.zeros(n.value // 4, dtype=np.float32)
        self._chk(self._lib.q3a_debug_read(self._h, name.encode(), out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
        return out

    def debug_read_raw(self, name: str) -> np.ndarray:
        """The same as bytes (records of mixed types: tools/soak_engines.py)."""
        n = C.c_uint64()
        self._chk(self._lib.q3a_debug_read(self._h, name.encode(), None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=np.uint8)
        if n.value:
            self._chk(self._lib.q3a_debug_read(self._h, name.encode(), out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
        return out


class HipGroup:
    """q3a_group: one process, one host thread per GPU; weights loaded once and replicated with one RCCL broadcast;
    utterances partitioned contiguously (include/q3asr.h, csrc/group.cpp)."""

    def __init__(self, model_dir: str, n_gpus: int = 1, devices: Optional[Sequence[int]] = None, precise: bool = False,
                 max_new_tokens: int = 4096):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        opts = _lib.Opts()
        self._lib.q3a_opts_default(C.byref(opts))
        opts.precise = int(precise)
        opts.max_new_tokens = int(max_new_tokens)
        dv = np.asarray(devices, dtype=np.int32) if devices is not None else None
        rc = self._lib.q3a_group_create(os.fsencode(model_dir), n_gpus, _i32p(dv) if dv is not None else None, C.byref(opts), C.byref(self._h))
        if rc != 0:
            raise Q3aError((self._lib.q3a_last_error(None) or b"").decode())

    @property
    def size(self) -> int:
        return int(self._lib.q3a_group_size(self._h))

    @property
    def used_rccl(self) -> bool:
        return bool(self._lib.q3a_group_used_rccl(self._h))

    def engine_timings(self, rank: int) -> dict:
        """Stage timings of rank's engine for its slice of the last batch (q3a_group_engine + q3a_stage_timings)."""
        h = self._lib.q3a_group_engine(self._h, rank)
        t = _lib.Timings()
        if not h or self._lib.q3a_stage_timings(C.c_void_p(h), C.byref(t)) != 0:
            return {}
        return {n: getattr(t, n) for n, _ in _lib.Timings._fields_}

    @property
    def startup_seconds(self) -> dict:
        """Stage times of q3a_group_create: checkpoint read + pack, H2D upload to the first GPU, RCCL broadcast, engine creation."""
        v = (C.c_double * 4)()
        self._lib.q3a_group_startup_seconds(self._h, v)
        return {"pack_s": v[0], "upload_s": v[1], "broadcast_s": v[2], "engines_s": v[3]}

    def transcribe_batch(self, clips: Sequence[np.ndarray], lang_prefix_ids: Optional[Sequence[int]] = None,
                         max_new: int = 4096, fixed_new_tokens: int = 0) -> List[List[int]]:
        arrs, ptrs, ns = HipEngine._ptrs(clips)
        B = len(arrs)
        stride = fixed_new_tokens if fixed_new_tokens > 0 else max_new
        out = np.zeros((B, stride), dtype=np.int32)
        lens = np.zeros(B, dtype=np.int32)
        pre = np.asarray(lang_prefix_ids if lang_prefix_ids is not None else [], dtype=np.int32)
        rc = self._lib.q3a_group_transcribe_ptrs(self._h, ptrs, _i64p(ns), B, _i32p(pre) if len(pre) else None, len(pre), max_new,
                                                 fixed_new_tokens, _i32p(out), stride, _i32p(lens))
        if rc != 0:
            raise Q3aError((self._lib.q3a_group_last_error(self._h) or b"").decode())
        return [out[b, :min(int(lens[b]), stride)].tolist() for b in range(B)]

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.q3a_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def measure_peaks(device: int = 0, reps: int = 5) -> dict:
    """q3a_measure_peaks (include/q3asr.h): HBM read / copy / triad GB/s and the 8192^3 bf16 GEMM TFLOP/s measured on this device."""
    lib = _lib.load()
    pk = _lib.Peaks()
    if lib.q3a_measure_peaks(device, reps, C.byref(pk)) != 0:
        raise RuntimeError("q3a_measure_peaks failed (no HIP device, or less than 2 GiB of free device memory)")
    return {"hbm_read_GBps": round(pk.hbm_read_gbps, 1), "hbm_copy_GBps": round(pk.hbm_copy_gbps, 1), "hbm_triad_GBps": round(pk.hbm_triad_gbps, 1),
            "mfma_bf16_TFLOPs": round(pk.mfma_bf16_tflops, 1), "mfma_bf16_TFLOPs_random_data": round(pk.mfma_bf16_tflops_random, 1), "gemm": [pk.gemm_m, pk.gemm_n, pk.gemm_k], "read_sweep_bytes": int(pk.hbm_read_bytes),
            "n_cu": pk.n_cu, "best_of": pk.reps}


def selftest_gemm16(M: int, N: int, K: int, reps: int = 0, device: int = 0) -> dict:
    """bf16-activation GEMM (global_load_lds path) vs the naive device reference; optional timing of both GEMMs."""
    lib = _lib.load()
    err, ref, t16, t32 = C.c_float(), C.c_float(), C.c_float(), C.c_float()
    rc = lib.q3a_selftest_gemm16(device, M, N, K, reps, C.byref(err), C.byref(ref), C.byref(t16), C.byref(t32))
    if rc != 0:
        raise Q3aError((lib.q3a_last_error(None) or b"").decode())
    flops = 2.0 * M * N * K
    out = {"err": err.value, "ref_max": ref.value}
    if reps > 0:
        out.update(us_bf16=t16.value, us_f32=t32.value, tflops_bf16=flops / t16.value / 1e6, tflops_f32=flops / t32.value / 1e6)
    return out


def selftest_gemm(M: int, N: int, K: int, split: bool = False, device: int = 0) -> Tuple[float, float]:
    lib = _lib.load()
    err, ref = C.c_float(), C.c_float()
    rc = lib.q3a_selftest_gemm(device, M, N, K, int(split), C.byref(err), C.byref(ref))
    if rc != 0:
        raise Q3aError((lib.q3a_last_error(None) or b"").decode())
    return err.value, ref.value


# ------------------------------------------------------------------------------------------------------
# Reference-shaped front door
# ------------------------------------------------------------------------------------------------------
from .audio import AsrTokenizer, capitalize_first, load_audio, parse_asr_output  # noqa: E402  (C++ host code behind the C ABI)


@dataclass
class TranscribeResult:
    """src/inference.rs:269-274 (+ the raw ids, which is where parity is pinned)."""
    text: str
    language: str
    raw_output: str
    ids: List[int]


class AsrInference:
    """Drop-in for the reference's AsrInference (src/inference.rs:19-27) backed by the HIP engine."""

    def __init__(self, engine: HipEngine, tokenizer=None):
        self.engine = engine
        self.tokenizer = tokenizer

    @classmethod
    def load(cls, model_dir: str, device: int = 0, **engine_kwargs) -> "AsrInference":
        """src/inference.rs:30-86.  The tokenizer (tokenizer.json, src/tokenizer.rs:11-30) is optional here:
        synthetic checkpoints have none and parity is defined on token ids."""
        eng = HipEngine(model_dir, device, **engine_kwargs)
        tok = None
        tj = os.path.join(model_dir, "tokenizer.json")
        if os.path.exists(tj):
            tok = AsrTokenizer(tj)
        return cls(eng, tok)

    def transcribe(self, audio, language: Optional[str] = None, max_new_tokens: int = 4096) -> TranscribeResult:
        """src/inference.rs:89-213.  `audio`: path to a WAV file or a 16 kHz float32 array."""
        if isinstance(audio, (str, os.PathLike)):
            samples = load_audio(os.fspath(audio), 16000)
        else:
            samples = np.asarray(audio, dtype=np.float32)
        prefix = None
        if language is not None:
            if self.tokenizer is None:
                raise Q3aError("forcing a language needs tokenizer.json (src/inference.rs:246-251)")
            prefix = self.tokenizer.encode("language " + capitalize_first(language))
        ids = self.engine.transcribe_batch([samples], prefix, max_new_tokens)[0]
        raw = self.tokenizer.decode(ids, True) if self.tokenizer is not None else ""
        lang, text = parse_asr_output(raw, language is not None)
        return TranscribeResult(text, lang, raw, ids)

"""Minimal WAV ingest (host glue; the reference's FFmpeg path src/audio.rs:17-160 is out of scope, its
hound/rubato fallback src/audio.rs:162-245 is what this mirrors): PCM s16/s32/f32 WAV -> mono by channel
mean (audio.rs:192-200) -> float32 in [-1,1) with scale 1/2^(bits-1) (audio.rs:177) -> 16 kHz by a
deterministic polyphase windowed-sinc resampler (scipy.signal.resample_poly; the reference's rubato
sinc resampler is not bit-reproducible outside Rust, so parity is pinned at the 16 kHz vector)."""
from __future__ import annotations

import wave
from math import gcd

import numpy as np


def read_wav(path: str):
    with wave.open(path, "rb") as w:
        nch, sw, sr, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    if sw == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif sw == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif sw == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError(f"unsupported sample width {sw}")
    if nch > 1:
        x = x.reshape(-1, nch).mean(axis=1)
    return x.astype(np.float32), sr


def resample(x: np.ndarray, sr_in: int, sr_out: int) -> np.ndarray:
    if sr_in == sr_out:
        return x.astype(np.float32)
    from scipy.signal import resample_poly
    g = gcd(sr_in, sr_out)
    return resample_poly(x.astype(np.float64), sr_out // g, sr_in // g).astype(np.float32)


def load_audio(path: str, target_sr: int = 16000) -> np.ndarray:
    """src/audio.rs:7 load_audio(path, target_sample_rate) -> mono f32 at target rate."""
    x, sr = read_wav(path)
    return resample(x, sr, target_sr)

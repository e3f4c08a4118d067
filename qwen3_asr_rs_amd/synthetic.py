"""Synthetic checkpoints and clips (there are no real Qwen3-ASR weights or datasets on the
build/GPU boxes).  The checkpoint is a *genuine* safetensors directory in the reference's
key layout (reference loaders: src/weights.rs:10-120,
src/audio_encoder.rs:31-76, src/text_decoder.rs:49-88, src/layers.rs) so that both the HIP
engine's loader and the fp32 oracle read the very same bytes.
"""
from __future__ import annotations

import hashlib
import json
import os
import struct
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

# ---- model dimension presets ---------------------------------------------------------------
# 0.6B = serde defaults of the reference (src/config.rs:52-62,90-99).
CONFIG_0P6B = {
    "audio_config": dict(d_model=896, encoder_layers=18, encoder_attention_heads=14, encoder_ffn_dim=3584,
                         num_mel_bins=128, max_source_positions=1500, n_window=50, n_window_infer=800,
                         conv_chunksize=500, downsample_hidden_size=480, output_dim=1024),
    "text_config": dict(vocab_size=151936, hidden_size=1024, intermediate_size=3072, num_hidden_layers=28,
                        num_attention_heads=16, num_key_value_heads=8, head_dim=128, rms_norm_eps=1e-6,
                        rope_theta=1000000.0, tie_word_embeddings=True,
                        rope_scaling=dict(rope_type="default", mrope_section=[24, 20, 20], mrope_interleaved=True)),
}
# 1.7B: expected dims (verified against config.json at load when a real checkpoint exists).
CONFIG_1P7B = {
    "audio_config": dict(d_model=1024, encoder_layers=24, encoder_attention_heads=16, encoder_ffn_dim=4096,
                         num_mel_bins=128, max_source_positions=1500, n_window=50, n_window_infer=800,
                         conv_chunksize=500, downsample_hidden_size=480, output_dim=2048),
    "text_config": dict(vocab_size=151936, hidden_size=2048, intermediate_size=6144, num_hidden_layers=28,
                        num_attention_heads=16, num_key_value_heads=8, head_dim=128, rms_norm_eps=1e-6,
                        rope_theta=1000000.0, tie_word_embeddings=True,
                        rope_scaling=dict(rope_type="default", mrope_section=[24, 20, 20], mrope_interleaved=True)),
}
# tiny: same topology, small dims -- the oracle finishes in well under a second.
CONFIG_TINY = {
    "audio_config": dict(d_model=128, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=256,
                         num_mel_bins=128, max_source_positions=1500, n_window=50, n_window_infer=800,
                         conv_chunksize=500, downsample_hidden_size=32, output_dim=256),
    "text_config": dict(vocab_size=151936, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                        num_attention_heads=4, num_key_value_heads=2, head_dim=128, rms_norm_eps=1e-6,
                        rope_theta=1000000.0, tie_word_embeddings=True),
}
# tiny2: untied lm_head, GQA ratio 4, contiguous mrope map -- exercises the optional branches.
CONFIG_TINY_UNTIED = {
    "audio_config": dict(CONFIG_TINY["audio_config"], encoder_layers=1),
    "text_config": dict(CONFIG_TINY["text_config"], tie_word_embeddings=False, num_attention_heads=8,
                        num_key_value_heads=2, num_hidden_layers=3,
                        rope_scaling=dict(rope_type="default", mrope_section=[24, 20, 20], mrope_interleaved=False)),
}
# Token-embedding std of the checkpoints bench.py times and tests/test_gpu_configs.py checks.  With the 0.02 of the golden
# fixtures the (tied) logits are nearly flat -- top-1/top-2 gaps of a few hundredths next to a bf16 engine's logit error of
# ~0.015, so 7-15 % of the greedy steps were undecidable; at 0.04 the gaps are 0.2-0.55 at |logit| <= 6 (measured with the oracle
# at the 0.6B dimensions), i.e. 10x the rounding noise -- closer to a trained checkpoint's peaked distributions.
PEAKED_EMBED_SCALE = 0.04
PRESETS = {"0.6b": CONFIG_0P6B, "1.7b": CONFIG_1P7B, "tiny": CONFIG_TINY, "tiny_untied": CONFIG_TINY_UNTIED}


def tensor_specs(cfg: dict, embed_scale: float = 0.02) -> List[Tuple[str, Tuple[int, ...], str, float]]:
    """(key, shape, kind, scale) for every tensor of the reference key map (W2)."""
    a, t = cfg["audio_config"], cfg["text_config"]
    d, ffn, ch, mel = a["d_model"], a["encoder_ffn_dim"], a["downsample_hidden_size"], a["num_mel_bins"]
    f3 = ((((mel - 1) // 2 + 1) - 1) // 2 + 1 - 1) // 2 + 1   # mel bins after three stride-2 convs (16)
    at = "thinker.audio_tower"
    s: List[Tuple[str, Tuple[int, ...], str, float]] = []
    s += [(f"{at}.conv2d1.weight", (ch, 1, 3, 3), "w", 1.0 / 3.0), (f"{at}.conv2d1.bias", (ch,), "b", 0.02)]
    for n in ("conv2d2", "conv2d3"):
        s += [(f"{at}.{n}.weight", (ch, ch, 3, 3), "w", 1.4 / (9 * ch) ** 0.5), (f"{at}.{n}.bias", (ch,), "b", 0.02)]
    s += [(f"{at}.conv_out.weight", (d, ch * f3), "w", 1.0 / (ch * f3) ** 0.5)]
    for i in range(a["encoder_layers"]):
        p = f"{at}.layers.{i}"
        for ln in ("self_attn_layer_norm", "final_layer_norm"):
            s += [(f"{p}.{ln}.weight", (d,), "g", 0.1), (f"{p}.{ln}.bias", (d,), "b", 0.02)]
        for pr in ("q_proj", "k_proj", "v_proj"):
            s += [(f"{p}.self_attn.{pr}.weight", (d, d), "w", 1.0 / d ** 0.5), (f"{p}.self_attn.{pr}.bias", (d,), "b", 0.02)]
        s += [(f"{p}.self_attn.out_proj.weight", (d, d), "w", 0.5 / d ** 0.5), (f"{p}.self_attn.out_proj.bias", (d,), "b", 0.02)]
        s += [(f"{p}.fc1.weight", (ffn, d), "w", 1.0 / d ** 0.5), (f"{p}.fc1.bias", (ffn,), "b", 0.02)]
        s += [(f"{p}.fc2.weight", (d, ffn), "w", 0.5 / ffn ** 0.5), (f"{p}.fc2.bias", (d,), "b", 0.02)]
    s += [(f"{at}.ln_post.weight", (d,), "g", 0.1), (f"{at}.ln_post.bias", (d,), "b", 0.02)]
    s += [(f"{at}.proj1.weight", (d, d), "w", 1.0 / d ** 0.5), (f"{at}.proj1.bias", (d,), "b", 0.02)]
    s += [(f"{at}.proj2.weight", (a["output_dim"], d), "w", 0.05 / d ** 0.5), (f"{at}.proj2.bias", (a["output_dim"],), "b", 0.005)]
    h, inter, hd = t["hidden_size"], t["intermediate_size"], t["head_dim"]
    nq, nkv = t["num_attention_heads"], t["num_key_value_heads"]
    assert a["output_dim"] == h, "audio output_dim must equal decoder hidden_size"
    tm = "thinker.model"
    s += [(f"{tm}.embed_tokens.weight", (t["vocab_size"], h), "w", embed_scale)]
    for i in range(t["num_hidden_layers"]):
        p = f"{tm}.layers.{i}"
        s += [(f"{p}.input_layernorm.weight", (h,), "g", 0.1), (f"{p}.post_attention_layernorm.weight", (h,), "g", 0.1)]
        s += [(f"{p}.self_attn.q_proj.weight", (nq * hd, h), "w", 1.0 / h ** 0.5),
              (f"{p}.self_attn.k_proj.weight", (nkv * hd, h), "w", 1.0 / h ** 0.5),
              (f"{p}.self_attn.v_proj.weight", (nkv * hd, h), "w", 1.0 / h ** 0.5),
              (f"{p}.self_attn.o_proj.weight", (h, nq * hd), "w", 0.5 / (nq * hd) ** 0.5),
              (f"{p}.self_attn.q_norm.weight", (hd,), "g", 0.1), (f"{p}.self_attn.k_norm.weight", (hd,), "g", 0.1)]
        s += [(f"{p}.mlp.gate_proj.weight", (inter, h), "w", 1.0 / h ** 0.5),
              (f"{p}.mlp.up_proj.weight", (inter, h), "w", 1.0 / h ** 0.5),
              (f"{p}.mlp.down_proj.weight", (h, inter), "w", 0.5 / inter ** 0.5)]
    s += [(f"{tm}.norm.weight", (h,), "g", 0.1)]
    if not t.get("tie_word_embeddings", True):
        s += [("thinker.lm_head.weight", (t["vocab_size"], h), "w", 0.02)]
    return s


def _gen_tensor(shape, kind, scale, gen: torch.Generator) -> torch.Tensor:
    n = 1
    for x in shape:
        n *= x
    v = torch.randn(n, generator=gen, dtype=torch.float32) * scale
    if kind == "g":
        v = v + 1.0
    return v.reshape(shape).to(torch.bfloat16)


_ST_DTYPES = {"BF16": (torch.bfloat16, torch.int16, 2), "F16": (torch.float16, torch.int16, 2), "F32": (torch.float32, torch.int32, 4)}


def _write_safetensors(path: str, tensors: List[Tuple[str, torch.Tensor]], dtype: str = "BF16"):
    """dtype: storage type of every tensor (the published checkpoints are BF16; F16 / F32 exercise the widening paths of
    src/weights.rs:74-89,134-181)."""
    tdt, idt, esz = _ST_DTYPES[dtype]
    header: Dict[str, dict] = {}
    off = 0
    for name, t in tensors:
        nbytes = t.numel() * esz
        header[name] = {"dtype": dtype, "shape": list(t.shape), "data_offsets": [off, off + nbytes]}
        off += nbytes
    hj = json.dumps(header, separators=(",", ":")).encode("utf-8")
    hj += b" " * ((8 - len(hj) % 8) % 8)
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(hj)))
        f.write(hj)
        for _, t in tensors:
            f.write(t.to(tdt).contiguous().view(idt).numpy().tobytes())


def write_checkpoint(model_dir: str, preset: str = "tiny", seed: int = 0, shards: int = 1,
                     cfg: Optional[dict] = None, eos_trap: bool = False, dtype: str = "BF16", embed_scale: float = 0.02) -> str:
    """Write config.json + model.safetensors (or `shards` shard files + index json, exercising
    the sharded path of src/weights.rs:29-58).  Idempotent: a finished directory is reused.
    eos_trap (untied lm_head only): column 0 of the lm_head is zeroed except +/-64 on the two EOS rows,
    so every argmax is an EOS token -- exercises the stop condition of src/inference.rs:163-165.
    embed_scale: std of the token-embedding rows.  With the default 0.02 the decoder state is dominated by the audio context
    and greedy decoding repeats one token; a large value (with an untied lm_head) makes the state follow the token just fed,
    so the greedy trajectory wanders through the vocabulary -- what plant_eos needs to place stops at chosen steps."""
    cfg = cfg or PRESETS[preset]
    tag = hashlib.sha1(json.dumps([cfg, seed, shards, eos_trap] + ([dtype] if dtype != "BF16" else [])
                                  + ([embed_scale] if embed_scale != 0.02 else []), sort_keys=True).encode()).hexdigest()[:12]
    done = os.path.join(model_dir, f".complete.{tag}")
    if os.path.exists(done):
        return model_dir
    os.makedirs(model_dir, exist_ok=True)
    for stale in os.listdir(model_dir):  # another variant (seed / scale / dtype ...) was written here before: its marker must not outlive its weights
        if stale.startswith(".complete."):
            os.remove(os.path.join(model_dir, stale))
    with open(os.path.join(model_dir, "config.json"), "w") as f:
        json.dump({"thinker_config": cfg}, f, indent=1)
    gen = torch.Generator().manual_seed(seed)
    specs = tensor_specs(cfg, embed_scale)
    tensors = [(k, _gen_tensor(shape, kind, scale, gen)) for k, shape, kind, scale in specs]
    if eos_trap:
        assert not cfg["text_config"].get("tie_word_embeddings", True), "eos_trap needs an untied lm_head"
        for k, t in tensors:
            if k == "thinker.lm_head.weight":
                t[:, 0] = 0
                t[151643, 0] = 64.0
                t[151645, 0] = -64.0
    if shards <= 1:
        _write_safetensors(os.path.join(model_dir, "model.safetensors"), tensors, dtype)
    else:
        per = (len(tensors) + shards - 1) // shards
        wm = {}
        for si in range(shards):
            name = f"model-{si + 1:05d}-of-{shards:05d}.safetensors"
            part = tensors[si * per:(si + 1) * per]
            _write_safetensors(os.path.join(model_dir, name), part, dtype)
            for k, _ in part:
                wm[k] = name
        with open(os.path.join(model_dir, "model.safetensors.index.json"), "w") as f:
            json.dump({"metadata": {}, "weight_map": wm}, f)
    open(done, "w").close()
    return model_dir


def synthetic_clip(index: int, seconds: float = 30.0, sample_rate: int = 16000) -> np.ndarray:
    """Seeded speech-band noise: numpy default_rng(1234+index) white noise ->
    2-pole Butterworth band-pass [100 Hz, 4 kHz] -> 4 Hz raised-cosine envelope -> peak 0.5."""
    from scipy.signal import butter, lfilter
    n = int(round(seconds * sample_rate))
    rng = np.random.default_rng(1234 + index)
    x = rng.standard_normal(n)
    b, a = butter(2, [100.0, 4000.0], btype="bandpass", fs=sample_rate)
    x = lfilter(b, a, x)
    t = np.arange(n) / sample_rate
    env = 0.5 * (1.0 - np.cos(2.0 * np.pi * 4.0 * t + 0.37 * index))
    x = x * (0.15 + 0.85 * env)
    x = 0.5 * x / np.max(np.abs(x))
    return x.astype(np.float32)


# ---- checkpoints that emit EOS on purpose ----------------------------------------------------------------------------
# Random-init weights never produce an EOS token, so the reference's stop condition (src/inference.rs:160-167: each
# utterance breaks at ITS first EOS) would go untested and untimed.  The helpers below rewrite ONE row of the output
# embedding -- the row of <|endoftext|> (151643), an EOS id that does not occur in the prompt, so nothing upstream of the
# lm_head changes -- such that its logit wins at chosen decoder states and loses at all others.  Pure numpy: the states
# come from whoever calls (the fp32 oracle in tests, the engine's own debug taps in bench.py).
ENDOFTEXT_ID = 151643


def _st_header(path: str):
    with open(path, "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        return json.loads(f.read(n).decode("utf-8")), 8 + n


def _locate_tensor(model_dir: str, key: str):
    """(file, header entry, offset of the tensor's first byte in the file) -- single-file or sharded checkpoints."""
    single = os.path.join(model_dir, "model.safetensors")
    if os.path.exists(single):
        path = single
    else:
        with open(os.path.join(model_dir, "model.safetensors.index.json")) as f:
            path = os.path.join(model_dir, json.load(f)["weight_map"][key])
    hdr, data0 = _st_header(path)
    e = hdr[key]
    return path, e, data0 + e["data_offsets"][0]


def read_tensor(model_dir: str, key: str) -> np.ndarray:
    """One tensor of a safetensors checkpoint as fp32 (BF16 / F16 / F32 storage)."""
    path, e, off = _locate_tensor(model_dir, key)
    tdt, idt, esz = _ST_DTYPES[e["dtype"]]
    n = int(np.prod(e["shape"])) if e["shape"] else 1
    with open(path, "rb") as f:
        f.seek(off)
        raw = np.frombuffer(f.read(n * esz), dtype=np.int16 if esz == 2 else np.int32)
    return torch.from_numpy(raw.copy()).view(tdt).to(torch.float32).reshape(e["shape"]).numpy()


def overwrite_row(model_dir: str, key: str, row: int, values: np.ndarray) -> np.ndarray:
    """Overwrite one row of a 2-D tensor in place (values rounded to the tensor's storage type); returns the stored row
    widened back to fp32."""
    path, e, off = _locate_tensor(model_dir, key)
    tdt, idt, esz = _ST_DTYPES[e["dtype"]]
    rows, cols = e["shape"]
    assert 0 <= row < rows and values.shape == (cols,)
    stored = torch.from_numpy(np.asarray(values, dtype=np.float32)).to(tdt)
    with open(path, "r+b") as f:
        f.seek(off + row * cols * esz)
        f.write(stored.contiguous().view(idt).numpy().tobytes())
    return stored.to(torch.float32).numpy()


def overwrite_tensor(model_dir: str, key: str, values: np.ndarray) -> None:
    """Overwrite a whole tensor in place (values rounded to the tensor's storage type)."""
    path, e, off = _locate_tensor(model_dir, key)
    tdt, idt, esz = _ST_DTYPES[e["dtype"]]
    assert tuple(values.shape) == tuple(e["shape"])
    with open(path, "r+b") as f:
        f.seek(off)
        f.write(torch.from_numpy(np.ascontiguousarray(values, dtype=np.float32)).to(tdt).contiguous().view(idt).numpy().tobytes())


def output_embedding_key(model_dir: str) -> str:
    """Tensor the lm_head multiplies by (src/text_decoder.rs:71-79: tied -> the token embedding)."""
    with open(os.path.join(model_dir, "config.json")) as f:
        cfg = json.load(f)
    t = cfg.get("thinker_config", cfg)["text_config"]
    return "thinker.model.embed_tokens.weight" if t.get("tie_word_embeddings", True) else "thinker.lm_head.weight"


def final_rms_norm(model_dir: str, x: np.ndarray) -> np.ndarray:
    """The decoder's final RMSNorm (src/layers.rs:48-54 with thinker.model.norm.weight) on rows of `x` -- turns the
    engine's "head_in" debug tap (last-layer residual rows) into the rows the lm_head multiplies."""
    with open(os.path.join(model_dir, "config.json")) as f:
        cfg = json.load(f)
    eps = float(cfg.get("thinker_config", cfg)["text_config"].get("rms_norm_eps", 1e-6))
    w = read_tensor(model_dir, "thinker.model.norm.weight").astype(np.float64)
    x = np.asarray(x, dtype=np.float64)
    return (x / np.sqrt((x * x).mean(-1, keepdims=True) + eps) * w).astype(np.float32)


def plant_eos(model_dir: str, states: np.ndarray, fire: Sequence[bool], hi: float = 12.0, lo: float = -12.0,
              eos_id: int = ENDOFTEXT_ID) -> dict:
    """Rewrite row `eos_id` of the output embedding so that its logit is ~`hi` at the lm_head input rows states[i] with
    fire[i] and ~`lo` at all others (minimum-norm solution of the linear system; logits of random-init checkpoints stay
    within a few units, so `hi` wins the argmax and `lo` never does).  Returns the achieved logits with the ROUNDED row."""
    H = np.asarray(states, dtype=np.float64)
    fire = np.asarray(fire, dtype=bool)
    assert H.ndim == 2 and fire.shape == (H.shape[0],)
    t = np.where(fire, hi, lo)
    w, *_ = np.linalg.lstsq(H, t, rcond=None)
    stored = overwrite_row(model_dir, output_embedding_key(model_dir), eos_id, w.astype(np.float32))
    got = H @ stored.astype(np.float64)
    return {"row_norm": float(np.linalg.norm(stored)), "logits": got,
            "worst_fire": float(got[fire].min()) if fire.any() else None,
            "worst_quiet": float(got[~fire].max()) if (~fire).any() else None}

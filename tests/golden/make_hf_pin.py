"""Independent pin of the encoder / decoder restatement in oracle/q3asr_oracle.py.

HuggingFace `transformers` (5.15 in the build container) ships `models/qwen3_asr/`: an implementation of the same
network written by other people from the original Python model, with a different checkpoint naming and a different
op sequence (packed windows via cu_seqlens instead of a dense -inf mask, masked_scatter injection, Qwen3 decoder from
the generic library).  This script loads the SEEDED synthetic checkpoints (reference key layout)
into that model through a key remap, runs the three clips below and stores a few outputs; tests/test_oracle.py holds the
oracle to them.  It does not lift "parity unpinned" (only vectors produced by the reference could) but it is the guard
against the builder and the oracle sharing one misreading of src/layers.rs / src/audio_encoder.rs.

Nothing of `transformers` travels: only the numbers in hf_pin.npz.  Run in the build container:
    python tests/golden/make_hf_pin.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import q3asr_oracle as O  # noqa: E402
from qwen3_asr_rs_amd import synthetic  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def remap_key(k: str) -> str:
    """reference checkpoint key -> HF Qwen3ASRForConditionalGeneration state_dict key"""
    if k.startswith("thinker.audio_tower.proj1."):
        return "model.multi_modal_projector.linear_1." + k.split(".")[-1]
    if k.startswith("thinker.audio_tower.proj2."):
        return "model.multi_modal_projector.linear_2." + k.split(".")[-1]
    if k.startswith("thinker.audio_tower."):
        return "model.audio_tower." + k[len("thinker.audio_tower."):]
    if k.startswith("thinker.model."):
        return "model.language_model." + k[len("thinker.model."):]
    if k == "thinker.lm_head.weight":
        return "lm_head.weight"
    raise KeyError(k)


def load_hf(model_dir: str):
    from transformers import Qwen3ASRConfig, Qwen3ASRForConditionalGeneration
    cfg = O.AsrConfig.from_file(os.path.join(model_dir, "config.json"))
    a, t = cfg.audio, cfg.text
    hf_cfg = Qwen3ASRConfig(
        audio_config=dict(num_mel_bins=a.num_mel_bins, encoder_layers=a.encoder_layers,
                          encoder_attention_heads=a.encoder_attention_heads, encoder_ffn_dim=a.encoder_ffn_dim,
                          d_model=a.d_model, n_window=a.n_window, n_window_infer=a.n_window_infer,
                          output_dim=a.output_dim, downsample_hidden_size=a.downsample_hidden_size),
        text_config=dict(model_type="qwen3", vocab_size=t.vocab_size, hidden_size=t.hidden_size,
                         intermediate_size=t.intermediate_size, num_hidden_layers=t.num_hidden_layers,
                         num_attention_heads=t.num_attention_heads, num_key_value_heads=t.num_key_value_heads,
                         head_dim=t.head_dim, rms_norm_eps=t.rms_norm_eps, max_position_embeddings=65536,
                         rope_parameters={"rope_type": "default", "rope_theta": t.rope_theta},
                         tie_word_embeddings=t.tie_word_embeddings, attention_bias=False),
        tie_word_embeddings=t.tie_word_embeddings)
    hf_cfg._attn_implementation = "eager"
    model = Qwen3ASRForConditionalGeneration(hf_cfg).to(torch.float32).eval()
    weights = O.load_model_weights(model_dir)  # fp32, reference key names
    sd = {remap_key(k): v for k, v in weights.items()}
    if t.tie_word_embeddings:
        sd["lm_head.weight"] = sd["model.language_model.embed_tokens.weight"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    missing = [m for m in missing if "positional_embedding" not in m]
    assert not missing and not unexpected, (missing, unexpected)
    return model, cfg


@torch.no_grad()
def hf_run(model, cfg, clip: np.ndarray, steps: int):
    mel = O.WhisperFeatureExtractor(400, 160, cfg.audio.num_mel_bins, 16000).extract(clip)  # (128, F); M2 has its own HF pin (hf_mel.npz)
    F_ = mel.shape[1]
    chunk = cfg.audio.n_window * 2
    Fp = (F_ + chunk - 1) // chunk * chunk
    feats = torch.zeros(1, mel.shape[0], Fp)
    feats[0, :, :F_] = mel
    mask = torch.zeros(1, Fp, dtype=torch.long)
    mask[0, :F_] = 1
    audio = model.get_audio_features(feats, mask, return_dict=True).pooler_output  # (T, output_dim)
    T = audio.shape[0]
    ids, _ = O.build_prompt(T, None)
    inp = torch.tensor([ids], dtype=torch.long)
    out = model(input_ids=inp, input_features=feats, input_features_mask=mask, use_cache=True)
    logits0 = out.logits[0, -1].clone()
    past = out.past_key_values
    gen, nxt = [], int(logits0.argmax())
    step_top = [logits0.topk(4)]
    for _ in range(steps):
        gen.append(nxt)
        o = model(input_ids=torch.tensor([[nxt]]), past_key_values=past, use_cache=True)
        past = o.past_key_values
        lg = o.logits[0, -1]
        step_top.append(lg.topk(4))
        nxt = int(lg.argmax())
    return audio, logits0, gen, step_top


def main():
    out = {}
    cases = [("tiny", "/tmp/q3a_ckpt_tiny", dict(preset="tiny", seed=1), synthetic.synthetic_clip(0, 9.3)),       # 10 chunks: 2 windows
             ("tiny_short", "/tmp/q3a_ckpt_tiny", dict(preset="tiny", seed=1), synthetic.synthetic_clip(1, 2.17)),  # ragged last chunk
             ("untied", "/tmp/q3a_ckpt_tiny_untied", dict(preset="tiny_untied", seed=2, shards=3), synthetic.synthetic_clip(2, 4.0)),
             # round 6: ONE case at the real 0.6B dimensions (18 + 28 layers, 14 / 16 heads, 4 attention windows, P = 405) on the
             # checkpoint and the clip bench.py and tests/test_gpu_configs.py use -- the dimensions the GPU configs are judged at
             ("0p6b", "/tmp/q3a_ckpt_0p6b_peaked", dict(preset="0.6b", seed=0, embed_scale=synthetic.PEAKED_EMBED_SCALE), synthetic.synthetic_clip(0, 30.0))]
    only = sys.argv[1:]
    if only:  # regenerate the named cases only, keep the others as committed
        out.update({k: v for k, v in np.load(os.path.join(HERE, "hf_pin.npz")).items()})
    for name, d, kw, clip in cases:
        if only and name not in only:
            continue
        synthetic.write_checkpoint(d, **kw)
        model, cfg = load_hf(d)
        audio, logits0, gen, step_top = hf_run(model, cfg, clip, steps=8)
        del model
        out[f"{name}_T"] = np.int64(audio.shape[0])
        out[f"{name}_audio_embeds_q"] = audio.numpy()[::7]          # every 7th token row
        out[f"{name}_audio_embeds_sum"] = np.float64(audio.double().sum().item())
        out[f"{name}_logits0_q"] = logits0.numpy()[::97]            # every 97th vocabulary entry
        out[f"{name}_ids"] = np.array(gen, dtype=np.int64)
        out[f"{name}_top_idx"] = np.stack([t.indices.numpy() for t in step_top])
        out[f"{name}_top_val"] = np.stack([t.values.numpy() for t in step_top])
        print(name, "T", audio.shape[0], "ids", gen)
    np.savez_compressed(os.path.join(HERE, "hf_pin.npz"), **out)


if __name__ == "__main__":
    main()

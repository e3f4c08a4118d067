"""Generates the committed golden vectors (run in the build container, where `transformers` and the
reference tree are available).  Outputs:
  hf_mel.npz       log-mel of the three reference test clips (24 kHz WAV -> 16 kHz by qwen3_asr_rs_amd.audio)
                   computed by HuggingFace's WhisperFeatureExtractor -- an implementation independent of both
                   the oracle and the HIP kernel (the reference states it matches it, src/mel.rs:42-46).
                   Every 4th frame is stored to keep the fixture small.
  oracle_tiny.npz  frozen outputs of the fp32 oracle on the seeded tiny checkpoint (regression pin of the
                   oracle itself: catches accidental edits and libtorch kernel drift).
  oracle_stages.npz  the per-stage goldens of SURVEY.md section 8c(3) on the same checkpoint and a 9.3 s clip (10 chunks,
                   two attention windows): conv-stem output, encoder input/output, window segments, RoPE cos/sin
                   tables, prefill last-row logits (subsampled), decoder taps, 16 greedy ids.  The HIP engine is held
                   to the same file on the GPU (tests/test_gpu_parity.py::test_engine_against_committed_stage_goldens).
(hf_pin.npz, the independent HuggingFace-transformers pin of the encoder/decoder, is made by make_hf_pin.py.)
The test clips under tests/golden/test_audio/ are data files of the reference (test_audio/sample{1,2,3}.{wav,txt}).
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from qwen3_asr_rs_amd import synthetic
from qwen3_asr_rs_amd.audio import load_audio
from oracle import q3asr_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


def hf_mel():
    from transformers import WhisperFeatureExtractor
    fe = WhisperFeatureExtractor(feature_size=128, sampling_rate=16000, hop_length=160, n_fft=400)
    out = {}
    for i in (1, 2, 3):
        x = load_audio(os.path.join(HERE, "test_audio", f"sample{i}.wav"), 16000)
        m = fe._np_extract_fbank_features(x[None, :].astype(np.float64), device="cpu")[0]
        out[f"sample{i}_n"] = np.int64(len(x))
        out[f"sample{i}_mel_q"] = m[:, ::4].astype(np.float32)
        out[f"sample{i}_frames"] = np.int64(m.shape[1])
    np.savez_compressed(os.path.join(HERE, "hf_mel.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})


def oracle_tiny():
    d = synthetic.write_checkpoint("/tmp/q3a_ckpt_tiny", "tiny", seed=1)
    orc = O.AsrOracle(d)
    clip = synthetic.synthetic_clip(0, 9.3)
    r = orc.transcribe_ids(clip, fixed_new_tokens=4, want_taps=True)
    top = r.step_logits[0].topk(8)
    np.savez_compressed(os.path.join(HERE, "oracle_tiny.npz"),
                        mel_q=r.taps["mel"].numpy()[:, ::8], audio_embeds_head=r.taps["audio_embeds"].numpy()[:4],
                        audio_embeds_sum=np.float64(r.taps["audio_embeds"].double().sum().item()),
                        logits0_top_idx=top.indices.numpy(), logits0_top_val=top.values.numpy(),
                        ids=np.array(r.all_step_ids), T=np.int64(r.num_audio_tokens), P=np.int64(r.prompt_len))
    print("ids", r.all_step_ids, "T", r.num_audio_tokens, "P", r.prompt_len)


def oracle_stages():
    d = synthetic.write_checkpoint("/tmp/q3a_ckpt_tiny", "tiny", seed=1)
    orc = O.AsrOracle(d)
    clip = synthetic.synthetic_clip(0, 9.3)
    r = orc.transcribe_ids(clip, fixed_new_tokens=16, want_taps=True)
    t = r.taps
    tc = orc.cfg.text
    cos, sin = O.compute_mrope_cos_sin([list(range(r.prompt_len + 16))] * 3, tc.head_dim, tc.rope_theta, tc.mrope_section, tc.mrope_interleaved)
    chunk_tokens = [13] * 9 + [O.feat_extract_output_length(930 - 900)]
    out = dict(
        n_samples=np.int64(len(clip)), T=np.int64(r.num_audio_tokens), P=np.int64(r.prompt_len),
        window_segments=np.array(O.window_segments(chunk_tokens, orc.cfg.audio.n_window, orc.cfg.audio.n_window_infer), dtype=np.int64),
        conv3_q=t["conv3"].numpy()[::3, ::5, :, ::2],            # (chunks, C, F, T) subsampled
        enc_in_q=t["enc_in"].numpy()[::5], enc_last_q=t["enc_last"].numpy()[::5],
        audio_embeds_q=t["audio_embeds"].numpy()[::5],
        rope_cos_q=cos.numpy()[::9, :64], rope_sin_q=sin.numpy()[::9, :64],
        dec_layer0_q=t["dec_layer0"].numpy()[::11], dec_last_hidden=t["dec_last_hidden"].numpy(),
        logits0_q=r.step_logits[0].numpy()[::53], logits8_q=r.step_logits[8].numpy()[::53],
        ids16=np.array(r.all_step_ids, dtype=np.int64))
    np.savez_compressed(os.path.join(HERE, "oracle_stages.npz"), **out)
    print({k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    hf_mel()
    oracle_tiny()
    oracle_stages()

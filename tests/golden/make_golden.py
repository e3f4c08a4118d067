"""Generates the committed golden vectors (run in the build container, where `transformers` and the
reference tree are available).  Outputs:
  hf_mel.npz       log-mel of the three reference test clips (24 kHz WAV -> 16 kHz by qwen3_asr_rs_amd.audio)
                   computed by HuggingFace's WhisperFeatureExtractor -- an implementation independent of both
                   the oracle and the HIP kernel (the reference states it matches it, src/mel.rs:42-46).
                   Every 4th frame is stored to keep the fixture small.
  oracle_tiny.npz  frozen outputs of the fp32 oracle on the seeded tiny checkpoint (regression pin of the
                   oracle itself: catches accidental edits and libtorch kernel drift).
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


if __name__ == "__main__":
    hf_mel()
    oracle_tiny()

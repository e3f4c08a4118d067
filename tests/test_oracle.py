"""CPU tests that pin the fp32 oracle (oracle/q3asr_oracle.py) -- the reference has no numeric fixtures
(SURVEY.md section 4), so the pins are: an independent front end (HuggingFace), integer known answers read off
the reference source, and frozen oracle outputs."""
import os

import numpy as np
import torch

from oracle import q3asr_oracle as O
from qwen3_asr_rs_amd import synthetic
from qwen3_asr_rs_amd.audio import load_audio

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_mel_matches_hf_whisper_on_reference_clips():
    """src/mel.rs:42-46 says extract() matches HF WhisperFeatureExtractor; pin the oracle to it on the
    reference's own three test clips (24 kHz WAV -> 16 kHz by the package's stated resampler)."""
    g = np.load(os.path.join(GOLDEN, "hf_mel.npz"))
    fe = O.WhisperFeatureExtractor()
    for i in (1, 2, 3):
        x = load_audio(os.path.join(GOLDEN, "test_audio", f"sample{i}.wav"), 16000)
        assert len(x) == int(g[f"sample{i}_n"])
        m = fe.extract(x).numpy()
        assert m.shape == (128, int(g[f"sample{i}_frames"]))
        ref = g[f"sample{i}_mel_q"]
        d = np.abs(m[:, ::4] - ref)
        # fp32 FFT (oracle) vs fp64 numpy (HF): a handful of near-floor bins differ by ~1e-3, the bulk by <1e-5
        assert d.max() < 5e-3, d.max()
        assert d.mean() < 2e-5, d.mean()


def test_reference_clip_geometry():
    """sample1 128000 smp/800 fr/8 chunks/T=104/P=119; sample2 66560/416/5/54/69; sample3 89600/560/6/73/88."""
    expect = {1: (128000, 800, 104, 119), 2: (66560, 416, 54, 69), 3: (89600, 560, 73, 88)}
    for i, (n, fr, T, P) in expect.items():
        x = load_audio(os.path.join(GOLDEN, "test_audio", f"sample{i}.wav"), 16000)
        assert len(x) == n
        assert (n + 159) // 160 == fr
        assert O.get_output_length(fr) == T
        ids, pos = O.build_prompt(T)
        assert len(ids) == P and pos[0] == 9 and pos[-1] == 9 + T - 1
        # none of the three exercises the window mask (<= 8 chunks), audio_encoder.rs:181-183
        assert O.window_segments([13] * ((fr + 99) // 100), 50, 800) is None


def test_integer_known_answers():
    assert [O.feat_extract_output_length(n) for n in (100, 99, 50, 17, 8, 1)] == [13, 13, 7, 3, 1, 1]
    assert O.get_output_length(3000) == 390 and O.get_output_length(3001) == 391
    ids, pos = O.build_prompt(2, [11, 12])
    assert ids == [151644, 8948, 198, 151645, 198, 151644, 872, 198, 151669, 151676, 151676, 151670, 151645, 198,
                   151644, 77091, 198, 11, 12]
    assert pos == [9, 10]
    cm = O.build_contiguous_dim_map([24, 20, 20], 64)
    assert cm == [0] * 24 + [1] * 20 + [2] * 20
    im = O.build_interleaved_dim_map([24, 20, 20], 64)
    assert im[:6] == [0, 1, 2, 0, 1, 2] and im.count(0) == 24 and im.count(1) == 20 and im.count(2) == 20
    assert im[60:] == [0, 0, 0, 0]
    assert O.window_segments([13] * 30, 50, 800) == [104, 104, 104, 78]
    assert O.window_segments([13] * 9 + [5], 50, 800) == [104, 18]


def test_mrope_degenerates_to_plain_rope():
    """All three position rows are equal (inference.rs:259-266) so both dim maps give plain RoPE."""
    pos = [list(range(7))] * 3
    c1, s1 = O.compute_mrope_cos_sin(pos, 128, 1e6, [24, 20, 20], False)
    c2, s2 = O.compute_mrope_cos_sin(pos, 128, 1e6, [24, 20, 20], True)
    assert torch.equal(c1, c2) and torch.equal(s1, s2)
    assert torch.equal(c1[:, :64], c1[:, 64:])


def test_causal_and_window_masks():
    m = O.create_causal_mask(3, 2)
    assert m.shape == (1, 1, 3, 5)
    assert torch.isinf(m[0, 0, 0, 3]) and m[0, 0, 0, 2] == 0 and m[0, 0, 2, 4] == 0
    assert (O.create_causal_mask(1, 9) == 0).all()
    w = O.build_window_mask(122, [13] * 9 + [5], 50, 800)
    assert w.shape == (1, 1, 122, 122)
    assert w[0, 0, 0, 103] == 0 and torch.isinf(w[0, 0, 0, 104]) and w[0, 0, 121, 104] == 0 and torch.isinf(w[0, 0, 104, 103])


def test_oracle_frozen_outputs(tiny_oracle):
    g = np.load(os.path.join(GOLDEN, "oracle_tiny.npz"))
    clip = synthetic.synthetic_clip(0, 9.3)
    r = tiny_oracle.transcribe_ids(clip, fixed_new_tokens=4, want_taps=True)
    assert r.num_audio_tokens == int(g["T"]) and r.prompt_len == int(g["P"])
    np.testing.assert_allclose(r.taps["mel"].numpy()[:, ::8], g["mel_q"], atol=2e-4)
    np.testing.assert_allclose(r.taps["audio_embeds"].numpy()[:4], g["audio_embeds_head"], atol=1e-5)
    top = r.step_logits[0].topk(8)
    np.testing.assert_allclose(top.values.numpy(), g["logits0_top_val"], atol=1e-4)
    assert r.all_step_ids == g["ids"].tolist()


def test_oracle_matches_independent_hf_transformers_implementation(tiny_oracle, tiny_untied_dir):
    """tests/golden/hf_pin.npz: outputs of HuggingFace transformers' own `qwen3_asr` model (written by other people
    from the original Python model; different op sequence and checkpoint naming) on the seeded synthetic checkpoints,
    loaded through a key remap (tests/golden/make_hf_pin.py).  Three cases: two attention windows, a ragged last chunk,
    untied lm_head + GQA ratio 4 + contiguous mrope map.  The oracle must agree to fp32 rounding: audio embeddings,
    last-row logits, the top-4 vocabulary indices of 9 consecutive steps and 8 greedy ids."""
    g = np.load(os.path.join(GOLDEN, "hf_pin.npz"))
    cases = [("tiny", tiny_oracle, synthetic.synthetic_clip(0, 9.3)), ("tiny_short", tiny_oracle, synthetic.synthetic_clip(1, 2.17)),
             ("untied", O.AsrOracle(tiny_untied_dir), synthetic.synthetic_clip(2, 4.0))]
    for name, orc, clip in cases:
        r = orc.transcribe_ids(clip, fixed_new_tokens=9, want_taps=True)
        assert r.num_audio_tokens == int(g[f"{name}_T"])
        ae = r.taps["audio_embeds"].numpy()
        np.testing.assert_allclose(ae[::7], g[f"{name}_audio_embeds_q"], atol=2e-6)          # |embeds| <= 0.15
        assert abs(float(ae.astype(np.float64).sum()) - float(g[f"{name}_audio_embeds_sum"])) < 1e-4
        np.testing.assert_allclose(r.step_logits[0].numpy()[::97], g[f"{name}_logits0_q"], atol=2e-5)   # |logits| <= 1.6
        assert r.all_step_ids[:8] == g[f"{name}_ids"].tolist()
        for s in range(9):
            top = r.step_logits[s].topk(4)
            assert top.indices.tolist() == g[f"{name}_top_idx"][s].tolist(), (name, s)
            np.testing.assert_allclose(top.values.numpy(), g[f"{name}_top_val"][s], atol=2e-5)


def test_oracle_matches_hf_transformers_pin_at_0p6b_dims():
    """The same pin at the REAL 0.6B dimensions (round 6): 18 encoder layers x 14 heads, 28 decoder layers,
    GQA 16 / 8, vocabulary 151 936, a 30 s clip = 4 attention windows (104, 104, 104, 78 tokens), P = 405 -- the checkpoint and the
    clip bench.py and tests/test_gpu_configs.py use.  HuggingFace's independent implementation and the oracle must agree to fp32
    rounding through 46 layers: audio embeddings, last-row logits of the prefill, top-4 of 5 steps."""
    g = np.load(os.path.join(GOLDEN, "hf_pin.npz"))
    d = synthetic.write_checkpoint("/tmp/q3a_ckpt_0p6b_peaked", "0.6b", seed=0, embed_scale=synthetic.PEAKED_EMBED_SCALE)
    r = O.AsrOracle(d).transcribe_ids(synthetic.synthetic_clip(0, 30.0), fixed_new_tokens=5, want_taps=True)
    assert (r.num_audio_tokens, r.prompt_len) == (int(g["0p6b_T"]), 405) == (390, 405)
    ae = r.taps["audio_embeds"].numpy()
    scale = float(np.abs(g["0p6b_audio_embeds_q"]).max())
    np.testing.assert_allclose(ae[::7], g["0p6b_audio_embeds_q"], atol=2e-5 * max(scale, 1.0))
    lscale = float(np.abs(g["0p6b_logits0_q"]).max())
    np.testing.assert_allclose(r.step_logits[0].numpy()[::97], g["0p6b_logits0_q"], atol=1e-4 * max(lscale, 1.0))
    assert r.all_step_ids[:5] == g["0p6b_ids"].tolist()[:5]
    for s in range(5):
        top = r.step_logits[s].topk(4)
        assert top.indices.tolist() == g["0p6b_top_idx"][s].tolist(), s
        np.testing.assert_allclose(top.values.numpy(), g["0p6b_top_val"][s], atol=1e-4 * max(lscale, 1.0))


def test_oracle_stage_goldens(tiny_oracle):
    """per-stage goldens (conv stem, window segments of a 10-chunk input, cos/sin tables, encoder
    and decoder taps, 16 greedy ids) -- tests/golden/oracle_stages.npz, which the HIP engine is held to as well."""
    g = np.load(os.path.join(GOLDEN, "oracle_stages.npz"))
    clip = synthetic.synthetic_clip(0, 9.3)
    assert len(clip) == int(g["n_samples"])
    r = tiny_oracle.transcribe_ids(clip, fixed_new_tokens=16, want_taps=True)
    t = r.taps
    assert (r.num_audio_tokens, r.prompt_len) == (int(g["T"]), int(g["P"]))
    assert O.window_segments([13] * 9 + [O.feat_extract_output_length(30)], 50, 800) == g["window_segments"].tolist() == [104, 17]
    np.testing.assert_allclose(t["conv3"].numpy()[::3, ::5, :, ::2], g["conv3_q"], atol=1e-5)
    for k in ("enc_in", "enc_last", "audio_embeds"):
        np.testing.assert_allclose(t[k].numpy()[::5], g[k + "_q"], atol=1e-5, err_msg=k)
    tc = tiny_oracle.cfg.text
    cos, sin = O.compute_mrope_cos_sin([list(range(r.prompt_len + 16))] * 3, tc.head_dim, tc.rope_theta, tc.mrope_section, tc.mrope_interleaved)
    assert np.array_equal(cos.numpy()[::9, :64], g["rope_cos_q"]) and np.array_equal(sin.numpy()[::9, :64], g["rope_sin_q"])
    np.testing.assert_allclose(t["dec_layer0"].numpy()[::11], g["dec_layer0_q"], atol=1e-5)
    np.testing.assert_allclose(t["dec_last_hidden"].numpy(), g["dec_last_hidden"], atol=1e-5)
    np.testing.assert_allclose(r.step_logits[0].numpy()[::53], g["logits0_q"], atol=1e-4)
    np.testing.assert_allclose(r.step_logits[8].numpy()[::53], g["logits8_q"], atol=1e-4)
    assert r.all_step_ids == g["ids16"].tolist()


def test_oracle_teacher_forcing_equals_free_running(tiny_oracle):
    clip = synthetic.synthetic_clip(3, 2.5)
    a = tiny_oracle.transcribe_ids(clip, fixed_new_tokens=3)
    b = tiny_oracle.transcribe_ids(clip, forced_ids=a.ids)
    assert b.all_step_ids[:3] == a.all_step_ids
    for x, y in zip(a.step_logits, b.step_logits):
        assert torch.equal(x, y)
    c = tiny_oracle.transcribe_ids(clip, fixed_new_tokens=3, last_only=True)
    assert torch.allclose(c.step_logits[0], a.step_logits[0], atol=1e-5)


def test_sharded_checkpoint_equals_single(tmp_path):
    """weights.rs:29-58: the sharded index path yields the same tensors as the single file."""
    d1 = synthetic.write_checkpoint(str(tmp_path / "one"), "tiny", seed=5, shards=1)
    d3 = synthetic.write_checkpoint(str(tmp_path / "three"), "tiny", seed=5, shards=3)
    w1, w3 = O.load_model_weights(d1), O.load_model_weights(d3)
    assert w1.keys() == w3.keys()
    for k in w1:
        assert torch.equal(w1[k], w3[k])


def test_parse_asr_output():
    assert O.parse_asr_output("language English<asr_text>Hello there.", False) == ("English", "Hello there.")
    assert O.parse_asr_output("  language Chinese 你好", False) == ("Chinese", "你好")
    assert O.parse_asr_output("no prefix", False) == ("unknown", "no prefix")
    assert O.parse_asr_output(" raw text ", True) == ("forced", "raw text")
    assert O.capitalize_first("english") == "English" and O.capitalize_first("") == ""

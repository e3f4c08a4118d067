"""GPU parity (pytest -m gpu) of LONG-FORM audio at the real (0.6B) dimensions in the benchmarked mode.

The reference takes any clip length: the chunk loop of src/audio_encoder.rs:96-121 cuts the mel into 100-frame chunks,
src/audio_encoder.rs:172-260 masks the encoder attention to windows of 8 chunks, and the greedy loop runs until EOS or
4096 new tokens (src/inference.rs:153).  BASELINE's configs are all 30 s clips (4 windows, P = 405), so without these
tests nothing at the real dimensions exercised
  * more than 8 encoder windows at head dim 64 on fattn_dma_kernel (a 150 s clip: T = 1950 tokens in 19 windows),
  * a causal prefill of ~2000 rows on the 3-stage LDS ring (P = 1965),
  * the one-sequence decode attention beyond 8 key splits, across the 1024- and 2048-key marks (17 splits, merged inside
    the o_proj GEMV), the two-sequence path beyond the GEMV's merge table (two 330 s clips: 34 key splits, separate merge launch),
    and the batched decode attention walking 16 key tiles for one utterance and 1-4 for its neighbours.
Same acceptance rule as tests/test_gpu_configs.py (oracle teacher-forced on the engine's own history, exact ids over the
margin, every flip inside the pair condition), logit error bound LOGIT_TOL.

Reference path: src/inference.rs:89-200, src/audio_encoder.rs:79-260, src/layers.rs:284-342.
"""
import numpy as np
import pytest

from oracle import q3asr_oracle as O
from qwen3_asr_rs_amd import synthetic
from qwen3_asr_rs_amd.engine import HipEngine
from test_gpu_configs import EMBED_TOL, margin_report, rel_l2, stepwise_logits

pytestmark = pytest.mark.gpu


def _ckpt():
    return synthetic.write_checkpoint("/tmp/q3a_ckpt_0p6b_peaked", "0.6b", seed=0, embed_scale=synthetic.PEAKED_EMBED_SCALE)


def test_long_form_150s_clip_0p6b_dims_default_and_precise():
    """One 150 s clip alone (GEMV decode path): T = 1950 audio tokens in 19 attention windows (18 x 104 + 78), P = 1965.
    24 teacher-forced steps in both modes + the encoder output against the oracle."""
    d = _ckpt()
    clip = synthetic.synthetic_clip(40, 150.0)
    N = 24
    orc = O.AsrOracle(d)
    for precise in (True, False):
        eng = HipEngine(d, 0, precise=precise, max_new_tokens=N)
        ids = eng.transcribe_batch([clip], None, max_new=N, fixed_new_tokens=N)[0]
        assert len(ids) == N
        ref = orc.transcribe_ids(clip, forced_ids=ids[:N - 1], last_only=True, want_taps=True)
        assert ref.num_audio_tokens == 1950 and ref.prompt_len == 1965
        eng.mel([clip])
        emb = eng.encode()[0]
        e = rel_l2(emb, ref.taps["audio_embeds"].numpy())
        print(f"[longform] 150 s clip, {'precise' if precise else 'default'} mode: audio embeds rel-L2 {e:.2e}")
        assert e <= (1e-4 if precise else EMBED_TOL)
        L, T = stepwise_logits(eng, [HipEngine.build_prompt(1950)], [ids], N)
        assert [int(t[0]) for t in T] == ids, "stage-API decode differs from the hipGraph-replayed decode"
        margin_report(f"longform 150 s clip 0.6B dims {'precise' if precise else 'default'} mode", ids, [l[0] for l in L], ref,
                      tol=2e-4 if precise else None)
        eng.close()


def test_long_form_ragged_batch_150s_7s_30s_0p6b_dims():
    """The 150 s clip in a ragged batch with a 7 s and a 30 s clip (skinny MFMA decode path, key-split attention + merge:
    3 x 8 workgroups are below the batched kernel's threshold), default mode, every utterance against its own oracle run."""
    d = _ckpt()
    clips = [synthetic.synthetic_clip(40, 150.0), synthetic.synthetic_clip(41, 7.0), synthetic.synthetic_clip(42, 30.0)]
    N = 24
    orc = O.AsrOracle(d)
    eng = HipEngine(d, 0, max_new_tokens=N)
    ids = eng.transcribe_batch(clips, None, max_new=N, fixed_new_tokens=N)
    refs = [orc.transcribe_ids(c, forced_ids=i[:N - 1], last_only=True) for c, i in zip(clips, ids)]
    assert [r.num_audio_tokens for r in refs] == [1950, 91, 390]
    eng.mel(clips); eng.encode()
    prompts = [HipEngine.build_prompt(r.num_audio_tokens) for r in refs]
    L, T = stepwise_logits(eng, prompts, ids, N)
    for b in range(3):
        assert [int(t[b]) for t in T] == ids[b], f"utterance {b}: stage-API decode differs from the hipGraph-replayed decode"
        margin_report(f"longform ragged batch utterance {b} (T = {refs[b].num_audio_tokens})", ids[b], [l[b] for l in L], refs[b])
    eng.close()


def test_one_sequence_decode_across_1024_and_2048_keys_0p6b_dims():
    """Free-running greedy decode of ONE sequence whose context crosses the 1024-key mark (a 76 s clip: P = 1003, 40 tokens) and
    the 2048-key mark (the 150 s clip: P = 1965, 100 tokens: 16 -> 17 live key splits, re-captured graphs by split count),
    default mode.  The oracle is teacher-forced on the engine's history over ALL steps."""
    d = _ckpt()
    orc = O.AsrOracle(d)
    for idx, seconds, N, cross in ((43, 76.0, 40, 1024), (40, 150.0, 100, 2048)):
        clip = synthetic.synthetic_clip(idx, seconds)
        eng = HipEngine(d, 0, max_new_tokens=N)
        ids = eng.transcribe_batch([clip], None, max_new=N, fixed_new_tokens=N)[0]
        ref = orc.transcribe_ids(clip, forced_ids=ids[:N - 1], last_only=True)
        P = ref.prompt_len
        assert P < cross < P + N - 1, (P, cross)
        eng.mel([clip]); eng.encode()
        L, T = stepwise_logits(eng, [HipEngine.build_prompt(ref.num_audio_tokens)], [ids], N)
        assert [int(t[0]) for t in T] == ids, "stage-API decode differs from the hipGraph-replayed decode"
        margin_report(f"one sequence across {cross} keys (P = {P}, {N} tokens)", ids, [l[0] for l in L], ref)
        eng.close()


def test_two_five_minute_clips_separate_split_merge_launch_0p6b_dims():
    """Two 330 s clips as one batch (T = 4290 audio tokens in 42 windows, P = 4305): the two-sequence GEMV decode path with 34 live key
    splits per head -- 2 x 16 x 34 = 1088 partial entries exceed the o_proj GEMV's merge table (GEMV_ATTN_MAX_TABLE = 1024,
    csrc/k_gemv.hip), so the splits are merged by the separate attn_combine launch, a path no 30 s workload reaches (the reference
    accepts any clip length: src/audio_encoder.rs:96-121, src/inference.rs:153).  Prefill of 8610 rows on the batch-sized kernels.
    Default mode, 8 teacher-forced steps.  The two utterances are the SAME clip (the oracle's dense-mask encoder and 4305-row prefill
    take about a minute of CPU per run: one run serves both; the engine must also give both the same ids)."""
    d = _ckpt()
    clip = synthetic.synthetic_clip(50, 330.0)
    clips = [clip, clip.copy()]
    N = 8
    orc = O.AsrOracle(d)
    eng = HipEngine(d, 0, max_new_tokens=N)
    ids = eng.transcribe_batch(clips, None, max_new=N, fixed_new_tokens=N)
    assert ids[0] == ids[1], "identical utterances of one batch must decode identically"
    ref = orc.transcribe_ids(clip, forced_ids=ids[0][:N - 1], last_only=True)
    refs = [ref, ref]
    assert ref.num_audio_tokens == 4290 and ref.prompt_len == 4305
    eng.mel(clips); eng.encode()
    L, T = stepwise_logits(eng, [HipEngine.build_prompt(4290)] * 2, ids, N)
    for b in range(2):
        assert [int(t[b]) for t in T] == ids[b], f"utterance {b}: stage-API decode differs from the hipGraph-replayed decode"
        margin_report(f"two 330 s clips, utterance {b} (P = 4305, 34 key splits, separate merge launch)", ids[b], [l[b] for l in L], refs[b])
    eng.close()


def test_batched_decode_attention_long_and_short_contexts_0p6b_dims():
    """The batched decode attention kernel (one workgroup per (sequence, kv head), 128-key tiles) with very unequal contexts
    in one group: 16 utterances = one 150 s clip (16 key tiles) + fifteen 7-30 s clips (1-4 tiles); the long utterance and two short
    ones are compared with the oracle over 20 steps, default mode."""
    d = _ckpt()
    secs = [150.0] + [7.0 + 1.6 * i for i in range(15)]
    clips = [synthetic.synthetic_clip(60 + i, s) for i, s in enumerate(secs)]
    N = 20
    eng = HipEngine(d, 0, max_new_tokens=N)
    ids = eng.transcribe_batch(clips, None, max_new=N, fixed_new_tokens=N)
    orc = O.AsrOracle(d)
    keep = [0, 1, 15]
    refs = {b: orc.transcribe_ids(clips[b], forced_ids=ids[b][:N - 1], last_only=True) for b in keep}
    eng.mel(clips); eng.encode()
    prompts = [HipEngine.build_prompt(eng.num_audio_tokens(len(c))) for c in clips]
    L, T = stepwise_logits(eng, prompts, ids, N, keep=keep)
    for b in range(len(clips)):
        assert [int(t[b]) for t in T] == ids[b], f"utterance {b}: stage-API decode differs from the hipGraph-replayed decode"
    for b in keep:
        margin_report(f"batched decode attention, 16 utterances, utterance {b} (P = {refs[b].prompt_len})", ids[b], [l[b] for l in L], refs[b])
    eng.close()


def test_transcribe_ptrs_overlapped_upload_equals_resident_path(tiny_dir):
    """q3a_transcribe_batch_ptrs (pinned staging by host threads, H2D in pieces on a copy stream, log-mel per piece) against the
    PCM-resident path (q3a_upload_pcm + q3a_run_resident): identical mel bits and ids on a ragged batch; the geometry cache
    (same lengths, other samples) must not serve stale PCM; a differently shaped batch afterwards rebuilds the tables."""
    N = 6
    rng = np.random.default_rng(5)
    lens = [16000 * 3 + 17, 161, 16000 * 12, 4000, 16000 * 31 + 5, 16000 * 7, 8001, 16000 * 2, 16000 * 9 + 3, 320]
    clips_a = [(0.1 * rng.standard_normal(n)).astype(np.float32) for n in lens]
    clips_b = [(0.1 * rng.standard_normal(n)).astype(np.float32) for n in lens]   # same geometry, other content
    clips_c = [synthetic.synthetic_clip(i, 2.0 + i) for i in range(4)]
    eng = HipEngine(tiny_dir, 0, precise=True, max_new_tokens=N, debug_taps=True)
    ref_eng = HipEngine(tiny_dir, 0, precise=True, max_new_tokens=N, debug_taps=True)
    for clips in (clips_a, clips_b, clips_a, clips_c, clips_b):
        got = eng.transcribe_batch(clips, None, max_new=N, fixed_new_tokens=N)
        io = eng.io_timings()
        assert io["mode"] == 1 and 1 <= io["pieces"] <= 8
        mel_got = eng.debug_read("mel")
        ref_eng.upload_pcm(clips)
        ref_eng.run_resident(None, 0, N)
        want = ref_eng.fetch_ids(N)
        mel_want = ref_eng.debug_read("mel")
        assert got == want
        assert mel_got.shape == mel_want.shape and np.array_equal(mel_got, mel_want)
    # the concatenated form of the ABI goes through the same path
    import ctypes as C
    from qwen3_asr_rs_amd.engine import _f32p, _i32p, _i64p
    pcm, ns = HipEngine._concat(clips_c)
    out = np.zeros((len(clips_c), N), dtype=np.int32)
    ol = np.zeros(len(clips_c), dtype=np.int32)
    eng._chk(eng._lib.q3a_transcribe_batch(eng._h, _f32p(pcm), _i64p(ns), len(clips_c), None, 0, N, N, _i32p(out), N, _i32p(ol)))
    ref_eng.upload_pcm(clips_c); ref_eng.run_resident(None, 0, N)
    assert [out[b, :ol[b]].tolist() for b in range(len(clips_c))] == ref_eng.fetch_ids(N)
    eng.close(); ref_eng.close()

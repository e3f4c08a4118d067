"""GPU parity tests (pytest -m gpu): every stage of the HIP path, called through the C ABI, against the
fp32 oracle on the same seeded inputs.

Tolerances (floating point; measured headroom ~10x):
  precise mode (bf16 hi+lo split activations, fp32 KV)  rel-L2 <= 1e-4 per stage, logits max-abs <= 2e-4,
                                                        greedy ids exact
  default mode (bf16 activations into the MFMA, bf16 KV) rel-L2 <= 2e-2 per stage, logits max-abs <= 6e-2,
                                                        greedy ids exact wherever the oracle's top-1/top-2
                                                        margin exceeds twice the observed logit error
  log-mel (fp32, weight-free, both modes)               max-abs <= 1e-4 vs the oracle
"""
import os

import numpy as np
import pytest
import torch

from oracle import q3asr_oracle as O
from qwen3_asr_rs_amd import synthetic
from qwen3_asr_rs_amd.audio import load_audio
from qwen3_asr_rs_amd.engine import HipEngine, selftest_gemm

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = {True: dict(rel=1e-4, logit=2e-4), False: dict(rel=2e-2, logit=6e-2)}


def rel_l2(got, ref):
    got, ref = np.asarray(got, np.float64).ravel(), np.asarray(ref, np.float64).ravel()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert np.isfinite(got).all()
    return float(np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-30))


@pytest.mark.parametrize("shape", [(64, 64, 32), (70, 96, 64), (390, 2688, 896), (1000, 480, 4320), (3000, 1024, 3072)])
@pytest.mark.parametrize("split", [False, True])
def test_gemm_against_device_reference(shape, split):
    """MFMA GEMM vs a naive fp64-accumulating kernel on asymmetric random data (transpose-detecting)."""
    err, ref = selftest_gemm(*shape, split=split)
    assert err <= (2e-5 if split else 4e-3) * ref, (err, ref)


@pytest.mark.parametrize("shape", [
    (128, 128, 64), (130, 200, 96),                      # 32x64 tiles: one K tile in the 4-stage ring; BK = 32 path (K % 64 != 0)
    (70, 98, 64), (1100, 1602, 128),                     # N % 4 != 0: element-wise epilogue (32x64 ring; 64x64 tiles, 3 stages)
    (96, 128, 128), (70, 96, 256),                       # 32x64 tiles, ring shorter than / as long as its stage count
    (33, 36, 512), (390, 896, 3584), (390, 1024, 896),   # 32x32 tiles, 4 waves split K (steps of 128, ring of 4 stages)
    (520, 3072, 128), (405, 4096, 1024),                 # 64x64 tiles, 2 and 16 K tiles through the ring
    (1000, 480, 4320), (3000, 3584, 896),                # 128x128 / 256x256 tiles
])
def test_gemm16_against_device_reference(shape):
    """bf16-activation LDS-DMA GEMM (default mode) vs the fp64-accumulating device reference on the same bf16 inputs:
    products are exact, only the fp32 summation order differs.  The shapes cover every staging the dispatcher picks: the 3-4-stage
    rings with counted vmcnt (dense operands, up to 768 tiles), two LDS buffers (more tiles; BK = 32), the K-split 32x32 tiles."""
    from qwen3_asr_rs_amd.engine import selftest_gemm16
    r = selftest_gemm16(*shape)
    assert r["err"] <= 2e-5 * max(r["ref_max"], 1.0), r


def test_measured_peaks_are_plausible():
    """q3a_measure_peaks (csrc/k_peaks.hip; SURVEY.md section 8d "measure on the box"): the numbers bench.py prints as the measured
    denominators must be what an MI355X can do -- a read stream between a third of and the whole nominal 8 TB/s, copy and triad
    below the read rate x 1.2, the library's own 8192^3 bf16 GEMM between 0.3 and 1.0 of the nominal 2.5 PFLOP/s."""
    from qwen3_asr_rs_amd.engine import measure_peaks
    pk = measure_peaks(0, 3)
    assert 2600 < pk["hbm_read_GBps"] <= 8000, pk
    assert 1500 < pk["hbm_copy_GBps"] <= 1.2 * pk["hbm_read_GBps"] and 1500 < pk["hbm_triad_GBps"] <= 1.2 * pk["hbm_read_GBps"], pk
    assert 750 < pk["mfma_bf16_TFLOPs"] <= 2500 and pk["gemm"] == [8192, 8192, 8192] and pk["n_cu"] >= 64, pk
    # the same GEMM on random operands: slower (the chip clocks down under the toggling bits), never faster than the constant-operand run by more than noise
    assert 600 < pk["mfma_bf16_TFLOPs_random_data"] <= 1.05 * pk["mfma_bf16_TFLOPs"], pk


def test_mel_reference_clips_and_hf_golden(tiny_dir):
    """Weight-free stage: real parity on the reference's own clips, also against the HF fixture."""
    eng = HipEngine(tiny_dir, 0)
    fe = O.WhisperFeatureExtractor()
    g = np.load(os.path.join(GOLDEN, "hf_mel.npz"))
    clips = [load_audio(os.path.join(GOLDEN, "test_audio", f"sample{i}.wav"), 16000) for i in (1, 2, 3)]
    clips.append(synthetic.synthetic_clip(7, 30.0))
    clips.append(synthetic.synthetic_clip(8, 1.003))          # not a multiple of the hop, shorter than one chunk
    clips.append(np.zeros(4000, dtype=np.float32))            # digital silence: log floor / max-8 clamp path
    mels = eng.mel(clips)
    for i, (c, m) in enumerate(zip(clips, mels)):
        ref = fe.extract(c).numpy()
        assert m.shape == ref.shape
        assert np.abs(m - ref).max() <= 1e-4, (i, np.abs(m - ref).max())
    for i in (1, 2, 3):
        assert np.abs(mels[i - 1][:, ::4] - g[f"sample{i}_mel_q"]).max() < 5e-3
    eng.close()


def _stage_check(model_dir, clips, precise, steps=4):
    tol = TOL[precise]
    orc = O.AsrOracle(model_dir)
    res = [orc.transcribe_ids(c, fixed_new_tokens=steps, want_taps=True) for c in clips]
    eng = HipEngine(model_dir, 0, precise=precise, debug_taps=True, max_new_tokens=32)
    eng.mel(clips)
    embeds = eng.encode()
    cat = lambda key, f=(lambda t: t): np.concatenate([f(r.taps[key]).contiguous().numpy().ravel() for r in res])
    # fp32 VALU stage; the default mode stores the map rounded to bf16 (what conv2's MFMA consumes): |err| <= 2^-9 |x|
    assert rel_l2(eng.debug_read("conv1"), cat("conv1", lambda t: t.permute(0, 2, 3, 1))) <= (1e-5 if precise else 2.0 ** -9)
    assert rel_l2(eng.debug_read("conv2"), cat("conv2", lambda t: t.permute(0, 2, 3, 1))) <= tol["rel"]
    assert rel_l2(eng.debug_read("conv3"), cat("conv3", lambda t: t.permute(0, 3, 2, 1))) <= tol["rel"]
    for k in ("enc_in", "enc_layer0", "enc_last", "audio_embeds"):
        assert rel_l2(eng.debug_read(k), cat(k)) <= tol["rel"], k
    for b, e in enumerate(embeds):
        assert e.shape == tuple(res[b].taps["audio_embeds"].shape)
    prompts = [HipEngine.build_prompt(r.num_audio_tokens) for r in res]
    logits, nxt = eng.prefill(prompts)
    for k in ("dec_embed", "dec_layer0", "dec_last_hidden"):
        assert rel_l2(eng.debug_read(k), cat(k)) <= tol["rel"], k
    worst = 0.0
    for b in range(len(clips)):
        worst = max(worst, float(np.abs(logits[b] - res[b].step_logits[0].numpy()).max()))
    # teacher-forced decode keeps engine and oracle on the same token history
    step_tok = [nxt.copy()]
    for s in range(steps - 1):
        eng.set_next_tokens([r.all_step_ids[s] for r in res])
        lg, nx, _ = eng.decode_step()
        step_tok.append(nx.copy())
        for b in range(len(clips)):
            worst = max(worst, float(np.abs(lg[b] - res[b].step_logits[s + 1].numpy()).max()))
    assert worst <= tol["logit"], worst
    for b in range(len(clips)):
        for s in range(steps):
            top = res[b].step_logits[s].topk(2).values
            if precise or float(top[0] - top[1]) > 2 * worst:
                assert int(step_tok[s][b]) == res[b].all_step_ids[s], (b, s)
    # free-running whole path (graph-replayed decode)
    ids = eng.transcribe_batch(clips, None, max_new=steps, fixed_new_tokens=steps)
    for b in range(len(clips)):
        if precise:
            assert ids[b] == res[b].all_step_ids[:steps]
        else:  # default mode: the free-running ids equal the oracle's up to its first step inside the rounding noise
            for s in range(steps):
                top = res[b].step_logits[s].topk(2).values
                if float(top[0] - top[1]) <= 2 * worst:
                    break
                assert ids[b][s] == res[b].all_step_ids[s], (b, s)
    t = eng.timings()
    assert t["batch"] == len(clips) and t["decode_steps"] == steps - 1 and t["total_ms"] > 0
    eng.close()
    return worst


@pytest.mark.parametrize("precise", [True, False])
def test_stage_parity_tiny_ragged_batch(tiny_dir, precise):
    """B=3 ragged utterances (10 chunks -> two attention windows; 2.17 s; a length that is not a multiple of
    the hop): batched results must equal running the reference once per utterance."""
    clips = [synthetic.synthetic_clip(0, 9.3), synthetic.synthetic_clip(1, 2.17), synthetic.synthetic_clip(2, 4.0)[:63999]]
    _stage_check(tiny_dir, clips, precise)


def test_stage_parity_untied_gqa4_sharded(tiny_untied_dir):
    _stage_check(tiny_untied_dir, [synthetic.synthetic_clip(2, 4.0)], True)


def test_batch_above_gemv_path(tiny_dir):
    """B=6 > 4 switches the decode step from the GEMV path to the GEMM path."""
    clips = [synthetic.synthetic_clip(10 + i, 1.5 + 0.37 * i) for i in range(6)]
    _stage_check(tiny_dir, clips, True, steps=3)


def test_key_splits_follow_the_context_not_the_capacity(tiny_dir):
    """The one-sequence decode attention launches as many 128-key splits as the caches HOLD keys for (longest prompt + steps
    so far), whatever max_new_tokens reserves: a 40-token prompt generating 230 tokens crosses the 128- and 256-key marks, so
    the loop replays three different graphs (split count 1, 2, 3) out of the engine's graph cache.  Precise mode: exact ids
    against the oracle, graph replay and eager, 1 / 2 sequences (GEMV path) and 5 (skinny path + split attention + merge
    launch); default mode: graph == eager; a second batch of the same shape re-uses the cached graphs."""
    steps = 230
    clips = [synthetic.synthetic_clip(500 + i, 1.6 + 0.3 * i) for i in range(5)]
    orc = O.AsrOracle(tiny_dir)
    want = [orc.transcribe_ids(c, fixed_new_tokens=steps).all_step_ids[:steps] for c in clips]
    for B in (1, 2, 5):
        for graph in (True, False):
            eng = HipEngine(tiny_dir, 0, precise=True, use_graph=graph, max_new_tokens=512)   # capacity 768 keys: 6 splits
            got = eng.transcribe_batch(clips[:B], None, max_new=steps, fixed_new_tokens=steps)
            assert got == want[:B], (B, graph)
            if graph:
                assert eng.transcribe_batch(clips[:B], None, max_new=steps, fixed_new_tokens=steps) == want[:B]
            eng.close()
    out = []
    for graph in (True, False):
        eng = HipEngine(tiny_dir, 0, use_graph=graph, max_new_tokens=512)
        out.append(eng.transcribe_batch(clips[:2], None, max_new=steps, fixed_new_tokens=steps))
        eng.close()
    assert out[0] == out[1]


def test_batched_head_argmax_in_the_gemm_epilogue(tiny_dir):
    """Default mode, 3..32 sequences: the lm_head GEMM writes one argmax partial per 64-column tile and row, and inside
    transcribe_batch the logits are not stored at all.  The ids of transcribe_batch, the ids of the step API and the argmax
    of the step API's logits (first index on ties, tensor.rs:370-372) are the same numbers."""
    for B in (5, 32):
        clips = [synthetic.synthetic_clip(300 + i, 1.0 + 0.21 * (i % 6)) for i in range(B)]
        eng = HipEngine(tiny_dir, 0, max_new_tokens=16)  # no debug taps: the fused loop runs without a logits buffer write
        ids = eng.transcribe_batch(clips, None, max_new=6, fixed_new_tokens=6)
        eng.mel(clips)
        prompts = [HipEngine.build_prompt(e.shape[0]) for e in eng.encode()]
        logits, nxt = eng.prefill(prompts)
        got = [[int(t)] for t in nxt]
        assert [int(np.argmax(logits[b])) for b in range(B)] == [g[0] for g in got]
        for _ in range(5):
            lg, nx, _done = eng.decode_step()
            for b in range(B):
                assert int(np.argmax(lg[b])) == int(nx[b]), b
                got[b].append(int(nx[b]))
        assert got == ids
        eng.close()


def test_batched_decode_attention_and_sequence_groups(tiny_dir):
    """Batched decode step: (i) the one-workgroup-per-(sequence, kv head) attention kernel (online softmax over 128-key
    tiles, context written directly, no merge launch) forced on at tiny dims, both modes, long context (7+ tiles);
    (ii) 40 sequences = a group of 32 + a group of 8 on the skinny MFMA path, each checked against per-utterance oracle
    runs (teacher-forced logits, exact ids in the precise mode)."""
    from qwen3_asr_rs_amd import _lib
    lib = _lib.load()
    try:
        assert lib.q3a_debug_set(b"dattn_batched_min_wgs", 1) == 0
        clips = [synthetic.synthetic_clip(30 + i, 1.2 + 0.41 * i) for i in range(5)] + [synthetic.synthetic_clip(29, 61.3)]
        _stage_check(tiny_dir, clips, True, steps=3)
        _stage_check(tiny_dir, clips, False, steps=3)
        clips = [synthetic.synthetic_clip(40 + i, 1.0 + 0.13 * (i % 9)) for i in range(40)]
        _stage_check(tiny_dir, clips, True, steps=3)
        assert lib.q3a_debug_set(b"dattn_batched_min_wgs", 1 << 30) == 0   # key splits + merge on the same groups
        _stage_check(tiny_dir, clips[:35], False, steps=3)
        # groups of 8 sequences as parallel stream / hipGraph branches: same ids as one group after the other, and oracle parity
        assert lib.q3a_debug_set(b"dattn_batched_min_wgs", 1) == 0
        assert lib.q3a_debug_set(b"decode_group_size", 8) == 0
        _stage_check(tiny_dir, clips[:29], True, steps=3)
        ids = {}
        for par in (1, 0):
            assert lib.q3a_debug_set(b"decode_parallel_groups", par) == 0
            eng = HipEngine(tiny_dir, 0, max_new_tokens=8)
            ids[par] = eng.transcribe_batch(clips[:29], None, max_new=8, fixed_new_tokens=8)
            eng.close()
        assert ids[0] == ids[1]
    finally:
        lib.q3a_debug_set(b"dattn_batched_min_wgs", 128)
        lib.q3a_debug_set(b"decode_group_size", 0)
        lib.q3a_debug_set(b"decode_parallel_groups", 1)


def test_fused_qknorm_rope_epilogue_of_the_qkv_gemm(tiny_dir, tiny_untied_dir):
    """Batch-sized prefills run QK-norm + RoPE + the KV-cache append as the epilogue of the 256x256 qkv GEMM.  Forced on
    at tiny dims (ragged row count: the last row tile is partial; GQA ratios 2 and 4, qkv bias in the untied preset),
    checked against the oracle (prefill logits, then teacher-forced decode steps that read the cache the epilogue
    wrote) and against the separate-kernel path on the same inputs."""
    from qwen3_asr_rs_amd import _lib
    lib = _lib.load()
    clips = [synthetic.synthetic_clip(60 + i, 2.0 + 0.53 * i) for i in range(7)]
    try:
        assert lib.q3a_debug_set(b"gemm256_min_tiles", 0) == 0
        for d in (tiny_dir, tiny_untied_dir):
            got = {}
            for fuse in (1, 0):
                assert lib.q3a_debug_set(b"fuse_qkrope", fuse) == 0
                if fuse:
                    _stage_check(d, clips, False, steps=3)
                eng = HipEngine(d, 0, debug_taps=True, max_new_tokens=8)
                eng.mel(clips)
                eng.encode()
                prompts = [HipEngine.build_prompt(eng.num_audio_tokens(len(c))) for c in clips]
                assert sum(len(p) for p in prompts) >= 128   # the 256-row-tile kernel takes the qkv projection
                logits, _ = eng.prefill(prompts)
                eng.set_next_tokens([11] * len(clips))
                lg, _, _ = eng.decode_step()
                got[fuse] = (logits.copy(), lg.copy())
                eng.close()
            for a, b in zip(got[1], got[0]):
                assert rel_l2(a, b) <= 2e-3
    finally:
        lib.q3a_debug_set(b"gemm256_min_tiles", 128)
        lib.q3a_debug_set(b"fuse_qkrope", 1)


def test_quarter_workgroup_skinny_gemm_matches_full_tiles(tiny_dir):
    """Batched decode step: the o / down projections as 8-row x 16-sequence workgroups (default) against the 16-row x
    32-sequence shape on the same inputs -- same K slices and reduction order per output, only the RMSNorm partial sums
    are grouped differently -- for a full group (32 = two sequence halves), a short group (8) and 16 < S < 32."""
    from qwen3_asr_rs_amd import _lib
    lib = _lib.load()
    clips = [synthetic.synthetic_clip(80 + i, 1.0 + 0.11 * (i % 7)) for i in range(40)]
    try:
        for n in (40, 21):
            got = {}
            for q in (1, 0):
                assert lib.q3a_debug_set(b"skinny_q", q) == 0
                eng = HipEngine(tiny_dir, 0, max_new_tokens=8)
                eng.mel(clips[:n])
                eng.encode()
                prompts = [HipEngine.build_prompt(eng.num_audio_tokens(len(c))) for c in clips[:n]]
                eng.prefill(prompts)
                eng.set_next_tokens([7 + i for i in range(n)])
                lg1, _, _ = eng.decode_step()
                lg2, _, _ = eng.decode_step()
                got[q] = np.concatenate([lg1.ravel(), lg2.ravel()])
                eng.close()
            # (fp32-level differences in 1/rms flip a bf16 rounding here and there downstream: bf16-noise level, not 1e-7)
            assert rel_l2(got[1], got[0]) <= 2e-3
        assert lib.q3a_debug_set(b"skinny_q", 1) == 0
        _stage_check(tiny_dir, clips[:9], False, steps=3)
    finally:
        lib.q3a_debug_set(b"skinny_q", 1)


def test_gate_up_skinny_gemm_forms_are_bit_identical():
    """Batched decode step, gate/up projection (k_skinny.hip): the pair form (16 gate + 16 up rows per workgroup; one pass at the
    0.6B dimensions, two passes with the partial tile aliased into the weight region where the pair form has more workgroups than
    the GPU has CUs: the 1.7B dimensions) and the half-pair form (3 tiles of 8 gate + 8 up rows per workgroup, knob skinny_glu_hp3:
    the default at the 1.7B dimensions, forced with 2 at the 0.6B dimensions) share K slices and reduction order per output
    element: logits of two teacher-forced steps must agree BIT FOR BIT -- 32 sequences (two sequence halves), 16 and 5 at the 0.6B
    dimensions, 16 and 32 (BASELINE configs[3] / configs[4] per GPU) at the 1.7B dimensions.  (Round 5 also held the single-pass
    pair form at the 1.7B dimensions to the same bits; its knob went with the round-6 pruning.)"""
    from qwen3_asr_rs_amd import _lib
    from qwen3_asr_rs_amd.distributed import pack_arena_host
    lib = _lib.load()
    clips = [synthetic.synthetic_clip(200 + i, 1.2 + 0.09 * (i % 9)) for i in range(32)]
    cases = [(synthetic.write_checkpoint("/tmp/q3a_ckpt_0p6b", "0.6b", seed=0), (32, 16, 5), (("pair", 0), ("half pair", 2))),
             (synthetic.write_checkpoint("/tmp/q3a_ckpt_1p7b", "1.7b", seed=0, shards=2), (16, 32), (("pair", 0), ("half pair", 1)))]
    try:
        for d, sizes, forms in cases:
            arena = pack_arena_host(d).to("cuda:0")  # one upload per model, an engine per form on top of it
            torch.cuda.synchronize()
            for n in sizes:
                got = {}
                for name, hp3 in forms:
                    assert lib.q3a_debug_set(b"skinny_glu_hp3", hp3) == 0
                    eng = HipEngine(d, 0, max_new_tokens=8, device_arena=(arena.data_ptr(), arena.numel()))
                    eng.mel(clips[:n])
                    eng.encode()
                    prompts = [HipEngine.build_prompt(eng.num_audio_tokens(len(c))) for c in clips[:n]]
                    eng.prefill(prompts, want_logits=False)
                    eng.set_next_tokens([11 + 3 * i for i in range(n)])
                    lg1, _, _ = eng.decode_step()
                    lg2, nx, _ = eng.decode_step()
                    got[name] = (lg1.copy(), lg2.copy(), nx.copy())
                    eng.close()
                for name, _ in forms[1:]:
                    for a, b in zip(got["pair"], got[name]):
                        assert np.isfinite(a).all() and np.array_equal(a, b), (d, n, name, float(np.abs(a.astype(np.float64) - b).max()))
            del arena
    finally:
        lib.q3a_debug_set(b"skinny_glu_hp3", 1)


@pytest.mark.parametrize("wgs", [2, 5, 13, 250])
def test_gemm256_walk_with_odd_workgroup_counts(wgs):
    """The walk's XCD chunk arithmetic (k_gemm256.hip: workgroup b walks tiles loc, loc + w, ... of the chunk of XCD b % 8) on grids
    that are not what an MI355X gives it: two workgroups walking everything, fewer workgroups than XCDs, counts that are not multiples
    of 8 -- against the fp64-accumulating device reference, on shapes with ragged edges, two K tiles (the seam variants are the whole
    K loop) and an odd K-tile count (the staging half flips per tile)."""
    from qwen3_asr_rs_amd import _lib
    from qwen3_asr_rs_amd.engine import selftest_gemm16
    lib = _lib.load()
    try:
        assert lib.q3a_debug_set(b"gemm256_min_tiles", 0) == 0
        assert lib.q3a_debug_set(b"gemm256_persist", wgs) == 0
        for shape in [(1300, 1030, 128), (2049, 1024, 448), (3000, 2688, 896)] if wgs > 5 else [(1300, 1030, 128), (1000, 770, 448)]:
            for _ in range(2):
                r = selftest_gemm16(*shape)
                assert r["err"] <= 2e-5 * max(r["ref_max"], 1.0), (wgs, shape, r)
    finally:
        lib.q3a_debug_set(b"gemm256_persist", 1)
        lib.q3a_debug_set(b"gemm256_min_tiles", 128)


def test_gemm256_persistent_walk_is_bit_identical_to_one_workgroup_per_tile():
    """k_gemm256.hip, round 6: a launch of more tiles than CUs is min(tiles, CUs) workgroups that WALK the tiles, the next tile's first
    K tile arriving under the epilogue (staging halved to 8 KiB per wave, four 32-row passes); knob gemm256_persist = 0 launches one
    workgroup per tile.  Same K order and the same epilogue arithmetic, so everything downstream must agree BIT FOR BIT: 32 x 30 s
    clips at the 0.6B dimensions put every epilogue kind through tile seams -- conv2 / conv3 (implicit-GEMM loader, 6000 / 1560 tiles),
    encoder qkv / fc1 (bf16, GELU; 506 / 686), decoder qkv with the QK-norm + RoPE + cache-append epilogue (768: the KV cache the
    decode steps read), gate / up (SwiGLU, 1224); the 196 / 204-tile residual shapes run one round either way.  The tile ORDER (knob
    gemm256_group_m: groups of 8 tile rows on the wide matrices by default) only changes which workgroup computes which tile.  Compared: audio
    embeddings, prefill logits, two decode steps' logits (they read the cache rows the fused epilogue wrote)."""
    from qwen3_asr_rs_amd import _lib
    from qwen3_asr_rs_amd.distributed import pack_arena_host
    lib = _lib.load()
    d = synthetic.write_checkpoint("/tmp/q3a_ckpt_0p6b", "0.6b", seed=0)
    clips = [synthetic.synthetic_clip(300 + i, 30.0) for i in range(32)]
    arena = pack_arena_host(d).to("cuda:0")
    torch.cuda.synchronize()
    got = {}
    try:
        for persist, group_m in ((0, 0), (1, 0), (1, 1), (0, 8)):  # (group_m: the tile ORDER -- 0 = by rule, 1 = N fastest, 8 = groups of 8 tile rows everywhere)
            assert lib.q3a_debug_set(b"gemm256_persist", persist) == 0
            assert lib.q3a_debug_set(b"gemm256_group_m", group_m) == 0
            eng = HipEngine(d, 0, max_new_tokens=8, device_arena=(arena.data_ptr(), arena.numel()))
            eng.mel(clips)
            emb = np.concatenate([e.ravel() for e in eng.encode()])
            prompts = [HipEngine.build_prompt(eng.num_audio_tokens(len(c))) for c in clips]
            logits, _ = eng.prefill(prompts)
            eng.set_next_tokens([11 + 3 * i for i in range(32)])
            lg1, _, _ = eng.decode_step()
            lg2, nx, _ = eng.decode_step()
            got[(persist, group_m)] = (emb, logits.copy(), lg1.copy(), lg2.copy(), nx.copy())
            eng.close()
    finally:
        lib.q3a_debug_set(b"gemm256_persist", 1)
        lib.q3a_debug_set(b"gemm256_group_m", 0)
    for key in ((1, 0), (1, 1), (0, 8)):
        for a, b in zip(got[(0, 0)], got[key]):
            assert np.isfinite(a).all() and np.array_equal(a, b), (key, float(np.abs(a.astype(np.float64) - b).max()))


def test_mfma_attention_matches_valu_attention(tiny_dir):
    """Default mode: the MFMA flash-attention kernels against the fp32 VALU kernels on the same inputs
    (two windows in the encoder, ragged causal prefill)."""
    clips = [synthetic.synthetic_clip(0, 9.3), synthetic.synthetic_clip(1, 2.17)]
    outs = []
    for valu in (False, True):
        eng = HipEngine(tiny_dir, 0, valu_attention=valu, debug_taps=True, max_new_tokens=8)
        eng.mel(clips)
        emb = np.concatenate([e.ravel() for e in eng.encode()])
        prompts = [HipEngine.build_prompt(eng.num_audio_tokens(len(c))) for c in clips]
        logits, _ = eng.prefill(prompts)
        outs.append((emb, eng.debug_read("dec_layer0"), logits.ravel()))
        eng.close()
    for a, b in zip(outs[0], outs[1]):
        assert rel_l2(a, b) <= 1e-2


def test_graph_replay_equals_eager(tiny_dir):
    clip = synthetic.synthetic_clip(4, 3.0)
    out = []
    for g in (True, False):
        eng = HipEngine(tiny_dir, 0, use_graph=g, max_new_tokens=16)
        out.append(eng.transcribe_batch([clip], None, max_new=8, fixed_new_tokens=8)[0])
        if g:  # second batch of a different shape re-captures the graph
            out.append(eng.transcribe_batch([clip, clip[:20000]], None, max_new=8, fixed_new_tokens=8)[0])
        eng.close()
    assert out[0] == out[1] == out[2] and len(out[0]) == 8


def test_eos_stops_generation(tmp_path):
    """A checkpoint whose lm_head always prefers an EOS id: generated_ids is empty (inference.rs:163-167)."""
    d = synthetic.write_checkpoint(str(tmp_path / "eos"), "tiny_untied", seed=4, eos_trap=True)
    clip = synthetic.synthetic_clip(5, 2.0)
    ref = O.AsrOracle(d).transcribe_ids(clip, max_new_tokens=16)
    assert ref.ids == [] and ref.all_step_ids[0] in (151643, 151645)
    eng = HipEngine(d, 0, precise=True, max_new_tokens=16)
    assert eng.transcribe_batch([clip], None, max_new=16) == [[]]
    # the cap also bounds a run that never sees EOS
    eng.close()


def test_max_new_cap_without_eos(tiny_dir):
    eng = HipEngine(tiny_dir, 0, max_new_tokens=64)
    ids = eng.transcribe_batch([synthetic.synthetic_clip(6, 2.0)], None, max_new=11)[0]
    assert len(ids) == 11
    eng.close()


def test_forced_language_prefix(tiny_dir):
    """inference.rs:246-251: the forced-language prompt appends encode("language X") ids."""
    clip = synthetic.synthetic_clip(9, 2.0)
    prefix = [11528, 6364]
    ref = O.AsrOracle(tiny_dir).transcribe_ids(clip, language_prefix_ids=prefix, fixed_new_tokens=3)
    eng = HipEngine(tiny_dir, 0, precise=True, max_new_tokens=8)
    assert eng.transcribe_batch([clip], prefix, max_new=3, fixed_new_tokens=3)[0] == ref.all_step_ids[:3]
    eng.close()


def test_replica_from_broadcast_arena(tiny_dir):
    """The multi-GPU start-up path on one GPU: pack the arena on the host, move the bytes into a torch device tensor
    (what dist.broadcast delivers on every rank) and create the engine from that pointer."""
    from qwen3_asr_rs_amd.distributed import pack_arena_host
    clip = synthetic.synthetic_clip(12, 2.5)
    ref_eng = HipEngine(tiny_dir, 0, max_new_tokens=8)
    ref = ref_eng.transcribe_batch([clip], None, max_new=6, fixed_new_tokens=6)[0]
    ref_eng.close()
    arena = pack_arena_host(tiny_dir).to("cuda:0")
    torch.cuda.synchronize()
    eng = HipEngine(tiny_dir, 0, max_new_tokens=8, device_arena=(arena.data_ptr(), arena.numel()))
    assert eng.transcribe_batch([clip], None, max_new=6, fixed_new_tokens=6)[0] == ref
    eng.close()
    from qwen3_asr_rs_amd.engine import Q3aError
    with pytest.raises(Q3aError, match="arena"):
        HipEngine(tiny_dir, 0, device_arena=(arena.data_ptr(), arena.numel() - 256))


def test_two_engines_on_one_gpu_share_an_arena_from_two_host_threads(tiny_dir):
    """Several engines per GPU on one weight arena (q3a_engine_create_from_arena), each driven by its own host thread -- the
    serving shape that overlaps one request's decode with the next one's encoder / prefill: per-engine state only, the ids are
    those of one engine run alone, batch after batch, for a GEMV-path batch and a skinny-path batch running CONCURRENTLY."""
    import threading
    from qwen3_asr_rs_amd.distributed import pack_arena_host
    arena = pack_arena_host(tiny_dir).to("cuda:0")
    torch.cuda.synchronize()
    work = [[synthetic.synthetic_clip(70, 2.0)], [synthetic.synthetic_clip(71 + i, 1.0 + 0.4 * i) for i in range(5)]]
    ref = []
    for clips in work:
        eng = HipEngine(tiny_dir, 0, max_new_tokens=32)
        ref.append(eng.transcribe_batch(clips, None, max_new=24, fixed_new_tokens=24))
        eng.close()
    engs = [HipEngine(tiny_dir, 0, max_new_tokens=32, device_arena=(arena.data_ptr(), arena.numel())) for _ in range(2)]
    got, errs = [[], []], []

    def run(i):
        try:
            for _ in range(6):
                got[i].append(engs[i].transcribe_batch(work[i], None, max_new=24, fixed_new_tokens=24))
        except Exception as ex:  # noqa: BLE001
            errs.append(ex)

    th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    assert not errs, errs
    # every batch complete and token-for-token equal to the engine that ran alone (round 3 tolerated one differing utterance
    # here: the packed-fp32 op_sel hazard of csrc/dev.h, gone since the round-4 build; tests/test_gpu_soak.py is the long version)
    for i in range(2):
        assert len(got[i]) == 6 and all(len(g) == len(ref[i]) and all(len(u) == 24 for u in g) for g in got[i]), i
        differing = sum(u != r for g in got[i] for u, r in zip(g, ref[i]))
        assert differing == 0, (i, differing)
    # the same two engines one after the other (nothing else on the GPU): exact
    for i in range(2):
        assert engs[i].transcribe_batch(work[i], None, max_new=24, fixed_new_tokens=24) == ref[i]
    for e in engs: e.close()


def test_native_group_one_gpu_through_rccl(tiny_dir, monkeypatch):
    """q3a_group_create / q3a_group_transcribe (csrc/group.cpp) on the one GPU of this box: Q3A_GROUP_FORCE_RCCL=1 makes the
    start-up go through librccl (dlopen, ncclCommInitAll with one rank, ncclBroadcast of the arena, destroy) exactly as the
    n-GPU path does; the group's ids must equal the single-engine ids, utterance by utterance."""
    from qwen3_asr_rs_amd.engine import HipGroup, Q3aError
    clips = [synthetic.synthetic_clip(50 + i, 1.0 + 0.3 * i) for i in range(3)]
    eng = HipEngine(tiny_dir, 0, max_new_tokens=8)
    ref = eng.transcribe_batch(clips, None, max_new=5, fixed_new_tokens=5)
    eng.close()
    monkeypatch.setenv("Q3A_GROUP_FORCE_RCCL", "1")
    grp = HipGroup(tiny_dir, 1, max_new_tokens=8)
    assert grp.size == 1 and grp.used_rccl
    st = grp.startup_seconds   # pinned pack -> async H2D -> ncclBroadcast -> engines, each timed (q3a_group_startup_seconds)
    assert st["pack_s"] > 0 and st["upload_s"] > 0 and st["broadcast_s"] > 0 and st["engines_s"] > 0, st
    assert grp.transcribe_batch(clips, None, max_new=5, fixed_new_tokens=5) == ref
    tm = grp.engine_timings(0)
    assert tm["batch"] == 3 and tm["total_ms"] > 0, tm
    grp.close()
    monkeypatch.delenv("Q3A_GROUP_FORCE_RCCL")
    grp = HipGroup(tiny_dir, 1, max_new_tokens=8)
    assert not grp.used_rccl and grp.transcribe_batch(clips[:1], None, max_new=5, fixed_new_tokens=5) == ref[:1]
    grp.close()
    with pytest.raises(Q3aError, match="out of range"):
        HipGroup(tiny_dir, 64)


def test_native_group_on_every_visible_gpu(tiny_dir):
    """The >1-rank start-up of q3a_group_create (ncclCommInitAll over N devices, ncclGroupStart / one ncclBroadcast per rank
    / ncclGroupEnd, csrc/group.cpp) and the one-thread-per-GPU q3a_group_transcribe: runs whenever this box shows at
    least two HIP devices (the 1-GPU test boxes skip it; an 8-GPU node exercises it without anybody editing a flag).
    Contiguous partition, ids equal to one engine's on the same utterances, fewer utterances than GPUs tolerated."""
    from qwen3_asr_rs_amd import _lib
    from qwen3_asr_rs_amd.engine import HipGroup
    n = int(_lib.load().q3a_device_count())
    if n < 2:
        pytest.skip(f"{n} HIP device(s) visible: the multi-rank RCCL path needs at least 2")
    n = min(n, 8)
    clips = [synthetic.synthetic_clip(60 + i, 1.0 + 0.21 * (i % 5)) for i in range(2 * n + 1)]
    eng = HipEngine(tiny_dir, 0, max_new_tokens=8)
    ref = eng.transcribe_batch(clips, None, max_new=5, fixed_new_tokens=5)
    eng.close()
    grp = HipGroup(tiny_dir, n, max_new_tokens=8)
    assert grp.size == n and grp.used_rccl
    assert grp.transcribe_batch(clips, None, max_new=5, fixed_new_tokens=5) == ref
    assert grp.transcribe_batch(clips[:1], None, max_new=5, fixed_new_tokens=5) == ref[:1]   # ranks 1.. get nothing
    grp.close()
    grp = HipGroup(tiny_dir, 2, devices=[n - 1, 0], max_new_tokens=8)   # explicit device list, root on the last GPU
    assert grp.size == 2 and grp.transcribe_batch(clips[:5], None, max_new=5, fixed_new_tokens=5) == ref[:5]
    grp.close()


def test_long_audio_many_windows_and_long_context(tiny_dir):
    """61.3 s clip: 62 chunks -> 8 attention windows in the encoder, prompt of ~800 tokens -> the decode step
    runs over 7+ key splits (flash-decoding merge in the o_proj GEMV); plus an exact multiple of the chunk size."""
    clips = [synthetic.synthetic_clip(20, 61.3)]
    _stage_check(tiny_dir, clips, True, steps=3)
    _stage_check(tiny_dir, [synthetic.synthetic_clip(21, 8.0)], False, steps=2)   # 800 frames = 8 full chunks: no window mask in the reference


def test_minimum_length_audio(tiny_dir):
    """161 samples is the shortest input reflection padding accepts (mel.rs:63-65): two frames, one chunk, one token."""
    clip = synthetic.synthetic_clip(22, 1.0)[:161]
    _stage_check(tiny_dir, [clip], True, steps=2)


def test_errors_are_reported(tiny_dir):
    from qwen3_asr_rs_amd.engine import Q3aError
    eng = HipEngine(tiny_dir, 0)
    with pytest.raises(Q3aError, match="too short"):
        eng.mel([np.zeros(100, dtype=np.float32)])
    eng.mel([synthetic.synthetic_clip(0, 1.0)])
    eng.encode()
    with pytest.raises(Q3aError, match="audio_pad"):
        eng.prefill([[151644, 8948, 198]])
    eng.close()


def test_parity_1p7b_dims_sharded_batch2():
    """BASELINE configs[3] shape class: 1.7B dims (expected dims), sharded safetensors
    (weights.rs:29-58), two ragged clips (the second one crosses the 8-chunk window boundary)."""
    d = synthetic.write_checkpoint("/tmp/q3a_ckpt_1p7b", "1.7b", seed=0, shards=2)
    clips = [synthetic.synthetic_clip(0, 6.0), synthetic.synthetic_clip(1, 11.0)]
    orc = O.AsrOracle(d)
    refs = [orc.transcribe_ids(c, fixed_new_tokens=2, last_only=True, want_taps=True) for c in clips]
    eng = HipEngine(d, 0, precise=True, max_new_tokens=8)
    eng.mel(clips)
    emb = eng.encode()
    for b in range(2):
        assert rel_l2(emb[b], refs[b].taps["audio_embeds"].numpy()) <= TOL[True]["rel"]
    logits, nxt = eng.prefill([HipEngine.build_prompt(r.num_audio_tokens) for r in refs])
    for b in range(2):
        assert float(np.abs(logits[b] - refs[b].step_logits[0].numpy()).max()) <= TOL[True]["logit"]
        assert int(nxt[b]) == refs[b].all_step_ids[0]
    eng.set_next_tokens([r.all_step_ids[0] for r in refs])
    lg, nx, _ = eng.decode_step()
    for b in range(2):
        assert float(np.abs(lg[b] - refs[b].step_logits[1].numpy()).max()) <= TOL[True]["logit"]
        assert int(nx[b]) == refs[b].all_step_ids[1]
    eng.close()
    del orc, refs


def test_parity_0p6b_dims_30s_clip():
    """BASELINE configs[1] shape: 0.6B dims (synthetic weights), one 30 s clip -> 4 attention windows
    (104,104,104,78), P=405.  Oracle uses last_only=True (same last-row logits, skips the all-position lm_head)."""
    d = synthetic.write_checkpoint("/tmp/q3a_ckpt_0p6b", "0.6b", seed=0)
    clip = synthetic.synthetic_clip(0, 30.0)
    steps = 3
    ref = O.AsrOracle(d).transcribe_ids(clip, fixed_new_tokens=steps, last_only=True, want_taps=True)
    assert ref.num_audio_tokens == 390 and ref.prompt_len == 405
    for precise in (True, False):
        eng = HipEngine(d, 0, precise=precise, max_new_tokens=16)
        eng.mel([clip])
        emb = eng.encode()[0]
        assert rel_l2(emb, ref.taps["audio_embeds"].numpy()) <= TOL[precise]["rel"]
        logits, nxt = eng.prefill([HipEngine.build_prompt(390)])
        err = float(np.abs(logits[0] - ref.step_logits[0].numpy()).max())
        assert err <= TOL[precise]["logit"] * (1 if precise else 2), err
        if precise:
            assert int(nxt[0]) == ref.all_step_ids[0]
            assert eng.transcribe_batch([clip], None, max_new=steps, fixed_new_tokens=steps)[0] == ref.all_step_ids[:steps]
        prof = eng.profile_decode_step()
        gemv = [prof[k] for k in ("gemv_qkv_gateup", "gemv_o_proj", "gemv_down", "gemv_lm_head")]
        assert [g["launches"] for g in gemv] == [56, 28, 28, 1]
        assert abs(sum(g["weight_bytes"] for g in gemv) - 1.192e9) < 2e6   # 1.192 GB per token
        ws = eng.profile_weight_stream(reps=1)
        assert ws["launches"] == 56 and ws["bytes_per_launch"] == (4096 + 6144) * 1024 * 2 / 2 and ws["avg_us"] > 0
        eng.close()


def test_parity_0p6b_dims_batch3_skinny_decode():
    """0.6B dims, three ragged short clips: the decode step runs on the skinny MFMA GEMM path (RMSNorm fused, bf16
    fragment-order activations into o/down, weights staged through LDS at K = 1024/2048/3072, lm_head on the LDS-DMA
    GEMM) -- shapes the tiny checkpoints do not reach.  Teacher-forced against per-utterance oracle runs."""
    d = synthetic.write_checkpoint("/tmp/q3a_ckpt_0p6b", "0.6b", seed=0)
    clips = [synthetic.synthetic_clip(3, 2.0), synthetic.synthetic_clip(4, 3.1), synthetic.synthetic_clip(5, 1.2)]
    steps = 3
    orc = O.AsrOracle(d)
    refs = [orc.transcribe_ids(c, fixed_new_tokens=steps, last_only=True) for c in clips]
    for precise in (True, False):
        eng = HipEngine(d, 0, precise=precise, max_new_tokens=8)
        eng.mel(clips)
        eng.encode()
        logits, nxt = eng.prefill([HipEngine.build_prompt(r.num_audio_tokens) for r in refs])
        scale = 1 if precise else 2
        for b in range(3):
            assert float(np.abs(logits[b] - refs[b].step_logits[0].numpy()).max()) <= TOL[precise]["logit"] * scale
        for s in range(steps - 1):
            eng.set_next_tokens([r.all_step_ids[s] for r in refs])
            lg, nx, _ = eng.decode_step()
            for b in range(3):
                err = float(np.abs(lg[b] - refs[b].step_logits[s + 1].numpy()).max())
                assert err <= TOL[precise]["logit"] * scale, (precise, s, b, err)
                if precise:
                    assert int(nx[b]) == refs[b].all_step_ids[s + 1]
        eng.close()
    del orc, refs


def test_engine_against_committed_stage_goldens(tiny_dir):
    """The HIP path against tests/golden/oracle_stages.npz directly (no oracle code runs): SURVEY.md section 8c(3) per-stage
    goldens -- conv stem, encoder in/out, audio embeddings, decoder taps, subsampled logits of steps 0 and 8, 16 greedy ids
    (precise mode: exact ids; default mode: stage tolerances of the module docstring)."""
    g = np.load(os.path.join(GOLDEN, "oracle_stages.npz"))
    clip = synthetic.synthetic_clip(0, 9.3)
    T, P = int(g["T"]), int(g["P"])
    for precise in (True, False):
        tol = TOL[precise]
        eng = HipEngine(tiny_dir, 0, precise=precise, debug_taps=True, max_new_tokens=16)
        eng.mel([clip])
        emb = eng.encode()[0]
        assert emb.shape[0] == T
        conv3 = eng.debug_read("conv3").reshape(10, 13, 16, -1)   # engine layout [chunk][t][f][c]; golden (chunk, c, f, t)
        assert rel_l2(conv3.transpose(0, 3, 2, 1)[::3, ::5, :, ::2], g["conv3_q"]) <= tol["rel"]
        D = g["enc_in_q"].shape[1]
        assert rel_l2(eng.debug_read("enc_in").reshape(T, D)[::5], g["enc_in_q"]) <= tol["rel"]
        assert rel_l2(eng.debug_read("enc_last").reshape(T, D)[::5], g["enc_last_q"]) <= tol["rel"]
        assert rel_l2(emb[::5], g["audio_embeds_q"]) <= tol["rel"]
        logits, nxt = eng.prefill([HipEngine.build_prompt(T)])
        H = g["dec_layer0_q"].shape[1]
        assert rel_l2(eng.debug_read("dec_layer0").reshape(P, H)[::11], g["dec_layer0_q"]) <= tol["rel"]
        assert rel_l2(eng.debug_read("dec_last_hidden"), g["dec_last_hidden"]) <= tol["rel"]
        assert float(np.abs(logits[0][::53] - g["logits0_q"]).max()) <= tol["logit"]
        ids16 = g["ids16"].tolist()
        toks = [int(nxt[0])]
        for s in range(8):
            eng.set_next_tokens([ids16[s]])
            lg, nx, _ = eng.decode_step()
            toks.append(int(nx[0]))
        assert float(np.abs(lg[0][::53] - g["logits8_q"]).max()) <= tol["logit"]
        if precise:
            assert toks == ids16[:9]
            assert eng.transcribe_batch([clip], None, max_new=16, fixed_new_tokens=16)[0] == ids16
        eng.close()


def test_cli_end_to_end(tiny_dir, tmp_path):
    """`asr <model_dir> <wav> [language]` (src/main.rs:7-81): stdout contract and agreement with the library path.
    The model directory gets a synthetic tokenizer.json (id i <-> token "t{i}", two special ids)."""
    import json
    import shutil
    import subprocess
    import wave
    from qwen3_asr_rs_amd.build import CLI_PATH
    from qwen3_asr_rs_amd.engine import AsrInference
    mdir = tmp_path / "model"
    mdir.mkdir()
    for f in os.listdir(tiny_dir):
        if f.endswith((".json", ".safetensors")):
            os.symlink(os.path.join(tiny_dir, f), mdir / f)
    vocab = {f"t{i}": i for i in range(151936) if i not in (151643, 151645)}
    tok = {"version": "1.0", "added_tokens": [{"id": 151643, "content": "<|endoftext|>", "special": True},
                                              {"id": 151645, "content": "<|im_end|>", "special": True}],
           "model": {"type": "BPE", "vocab": vocab, "merges": []}}
    (mdir / "tokenizer.json").write_text(json.dumps(tok))
    clip = synthetic.synthetic_clip(11, 2.0)
    wav = tmp_path / "clip.wav"
    with wave.open(str(wav), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(24000)
        x24 = np.interp(np.arange(int(len(clip) * 1.5)) / 1.5, np.arange(len(clip)), clip)
        w.writeframes((np.clip(x24, -1, 1) * 32767).astype("<i2").tobytes())
    r = subprocess.run([CLI_PATH, str(mdir), str(wav)], capture_output=True, text=True, timeout=300, env=dict(os.environ, RUST_LOG="warn"))
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().split("\n")
    assert len(lines) == 2 and lines[0].startswith("Language: ") and lines[1].startswith("Text: ")
    res = AsrInference.load(str(mdir), 0).transcribe(str(wav))
    assert len(res.ids) == 4096                              # random weights never emit EOS: the reference's cap (inference.rs:153)
    assert lines[0] == f"Language: {res.language}" and lines[1] == f"Text: {res.text}"
    assert res.raw_output.startswith(f"t{res.ids[0]}t{res.ids[1]}")

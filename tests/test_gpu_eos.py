"""GPU parity (pytest -m gpu) of the greedy loop's stop condition in the mode a user runs it: natural EOS, several
utterances of ONE batch finishing at different steps while the captured decode step keeps replaying.

Reference: src/inference.rs:160-167 -- every utterance breaks at ITS first EOS token and the EOS is not part of
generated_ids; "run the reference once per utterance" is the batch semantics.  The engine evaluates the condition on the
device (argmax_finalize: done[], n_done, pinned progress words) and the host loop never synchronises the stream, so
what is checked here is q3a_transcribe_batch(fixed_new_tokens = 0): out_lens / ids per utterance == the oracle's
natural-EOS run on the same (planted) checkpoint, for hipGraph replay and eager launches, both decode paths (GEMV: <= 2
sequences, skinny MFMA: groups of <= 32), several run-ahead depths, and the number of decode steps actually executed.
"""
import numpy as np
import pytest

from oracle import q3asr_oracle as O
from qwen3_asr_rs_amd import _lib, synthetic
from qwen3_asr_rs_amd.engine import HipEngine

from eos_plan import fresh_eos_checkpoint, leading_decidable, plan_ragged_eos

pytestmark = pytest.mark.gpu


def _expected(model_dir, clips, kmax, stops):
    orc = O.AsrOracle(model_dir)   # loaded AFTER planting
    ref = [orc.transcribe_ids(c, max_new_tokens=kmax, keep_logits=False, last_only=True) for c in clips]
    assert [len(r.ids) for r in ref] == [kmax if k is None else k for k in stops]
    return [r.ids for r in ref]


def _check(tag, got, ref, stops, decidable, exact):
    """exact (precise mode): every utterance equals the oracle.  Default mode: an utterance is compared when every step in
    front of its stop (all kmax steps for a never-EOS one) is decidable for a bf16 engine; a flipped token would lead to
    decoder states the planted row was never constrained on.  Returns the number of utterances compared."""
    assert len(got) == len(ref)
    compared = 0
    for u, (g, r) in enumerate(zip(got, ref)):
        assert synthetic.ENDOFTEXT_ID not in g and 151645 not in g, f"{tag}: utterance {u}: EOS inside generated ids"
        if not exact and decidable[u] < len(r):
            continue
        assert len(g) == len(r), f"{tag}: utterance {u}: {len(g)} ids, oracle {len(r)} (planned stop {stops[u]})"
        assert g == r, f"{tag}: utterance {u}: ids differ from the oracle"
        compared += 1
    return compared


def _run(model_dir, clips, kmax, **kw):
    eng = HipEngine(model_dir, 0, max_new_tokens=kmax, **kw)
    got = eng.transcribe_batch(clips, None, max_new=kmax)   # fixed_new_tokens = 0: natural EOS
    steps = int(eng.timings()["decode_steps"])
    eng.close()
    return got, steps


def test_ragged_eos_inside_a_batch_tiny_dims():
    """40 utterances (skinny MFMA path, groups 32 + 8), stops spread over steps 1..8 and "never"; precise and default
    mode, graph replay and eager; then subsets on the GEMV path (2 sequences) and a batch in which every utterance stops
    (the loop must end within `eos_run_ahead` steps of the last EOS)."""
    lib = _lib.load()
    d = fresh_eos_checkpoint("/tmp/q3a_ckpt_tinyu_eos", "tiny_untied", seed=5)
    B, kmax = 40, 9
    clips = [synthetic.synthetic_clip(200 + i, 1.0 + 0.17 * (i % 7)) for i in range(B)]
    pat = [1, None, 3, 5, 7, None, 2, 4, 8, 6, 3]
    # planned for the precise mode (fp32-level logits: a margin of 1e-3 is decidable); the default bf16 mode is compared on
    # the utterances whose steps in front of the stop all have a margin of 0.03 (measured logit error at these dims: 7e-3)
    stops, decidable, info = plan_ragged_eos(d, clips, kmax, [pat[i % len(pat)] for i in range(B)], margin_min=1e-3)
    dec_bf16 = [leading_decidable(m, 0.03) for m in info["margins"]]
    kinds = {k for k in stops}
    assert None in kinds and len(kinds) >= 5, stops
    print(f"[eos] tiny: planted stops {stops}; EOS row norm {info['row_norm']:.1f}; bf16-decidable leading steps {dec_bf16}")
    ref = _expected(d, clips, kmax, stops)
    runs = {}
    for precise in (True, False):
        for use_graph in (True, False):
            got, steps = _run(d, clips, kmax, precise=precise, use_graph=use_graph)
            runs[precise, use_graph] = got
            n = _check(f"tiny B=40 precise={precise} graph={use_graph}", got, ref, stops, decidable if precise else dec_bf16, exact=precise)
            print(f"[eos] tiny B=40 precise={precise} graph={use_graph}: {n} utterances compared with the oracle")
            assert n == B or not precise
            assert steps == kmax - 1   # a never-EOS utterance runs to the cap
        assert runs[precise, True] == runs[precise, False], "hipGraph replay and eager launches disagree"
    # every utterance of the batch stops: the device-side all-done flag ends the loop
    fin = [u for u in range(B) if stops[u] is not None]
    try:
        for ahead in (1, 2, 4):
            assert lib.q3a_debug_set(b"eos_run_ahead", ahead) == 0
            for sel in (fin[:12], fin[:2], [fin[0]]):   # skinny path, GEMV path with 2 sequences, one sequence
                sub = [clips[u] for u in sel]
                last = max(stops[u] for u in sel)
                got, steps = _run(d, sub, kmax, precise=True)
                _check(f"tiny subset {sel} ahead={ahead}", got, [ref[u] for u in sel], [stops[u] for u in sel],
                       [decidable[u] for u in sel], exact=True)
                assert last <= steps <= min(last + ahead, kmax - 1), (sel, ahead, last, steps)
    finally:
        lib.q3a_debug_set(b"eos_run_ahead", 2)


def test_ragged_eos_inside_a_batch_0p6b_dims_default_mode():
    """The same at the 0.6B dimensions (untied lm_head so that the <|endoftext|> row can be planted without touching the
    input embeddings) in the DEFAULT bf16 mode: 6 utterances on the skinny MFMA path and 2 on the GEMV path."""
    cfg = {"audio_config": dict(synthetic.CONFIG_0P6B["audio_config"]),
           "text_config": dict(synthetic.CONFIG_0P6B["text_config"], tie_word_embeddings=False)}
    d = fresh_eos_checkpoint("/tmp/q3a_ckpt_0p6bu_eos", "0.6b", seed=3, cfg=cfg)
    kmax = 8
    # logits of random-init weights are nearly flat, so at these dimensions only some steps are decidable for a bf16 engine:
    # plan over a pool of short clips and keep the utterances whose stop could be planted behind a decidable prefix
    pool = [synthetic.synthetic_clip(300 + i, 1.5 + 0.2 * i) for i in (2, 3, 4, 5, 8, 9, 10, 11)]
    p_stops, p_dec, info = plan_ragged_eos(d, pool, kmax, [4, 2, 5, 3, None, 4, 2, 6], margin_min=0.04)
    planted = sorted((u for u in range(len(pool)) if p_stops[u] is not None and p_stops[u] <= p_dec[u]), key=lambda u: p_stops[u])
    keep, seen = [], set()
    for u in planted:   # distinct stop steps first
        if p_stops[u] not in seen:
            keep.append(u); seen.add(p_stops[u])
    keep += [u for u in planted if u not in keep][:max(0, 5 - len(keep))]
    never = max((u for u in range(len(pool)) if p_stops[u] is None), key=lambda u: p_dec[u])
    sel = sorted(keep[:5] + [never])
    clips, stops, decidable = [pool[u] for u in sel], [p_stops[u] for u in sel], [p_dec[u] for u in sel]
    assert None in stops and len({k for k in stops if k is not None}) >= 2, (p_stops, stops)
    print(f"[eos] 0.6B dims: pool stops {p_stops}, kept {sel} -> {stops}; EOS row norm {info['row_norm']:.1f}")
    ref = _expected(d, clips, kmax, stops)
    for use_graph in (True, False):
        got, steps = _run(d, clips, kmax, use_graph=use_graph)
        n = _check(f"0.6B-dims B={len(clips)} graph={use_graph}", got, ref, stops, decidable, exact=False)
        assert n >= len(clips) - 1, n   # (the never-EOS utterance may have an undecidable step; the planted ones cannot)
        assert steps == kmax - 1
    fin = [u for u in range(len(clips)) if stops[u] is not None]
    for sub in (fin, fin[:2]):
        last = max(stops[u] for u in sub)
        got, steps = _run(d, [clips[u] for u in sub], kmax)
        _check(f"0.6B-dims subset {sub}", got, [ref[u] for u in sub], [stops[u] for u in sub], [decidable[u] for u in sub], exact=False)
        assert last <= steps <= min(last + 2, kmax - 1), (sub, last, steps)

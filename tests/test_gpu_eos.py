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

from eos_plan import free_run_margins, peaked_checkpoint, plan_class_stops

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


# bf16 engine noise on the logits of the peaked fixtures (eos_plan.py): walk logits of ~10 carry ~0.05; a planted row of norm n
# carries ~0.004 * |state| * n (tiny dims: |state| = 16, router rows of norm <= 130 -> <= 8; the planted margins are >= 20).
# An utterance is compared in the default mode when every step of the oracle's run has at least this top-1/top-2 margin:
BF16_MARGIN = 1.0


def test_ragged_eos_inside_a_batch_tiny_dims():
    """40 utterances (skinny MFMA path, groups 32 + 8) in 7 clip-length classes with stops at steps 1 / 2 / 3 / 5 / 7 and two
    never-EOS classes, on a PEAKED checkpoint (eos_plan.peaked_checkpoint: walk head + router rows + planted EOS row), so that
    the default bf16 mode is decidable too: precise and default mode, graph replay and eager, ALL 40 utterances compared with the
    oracle's natural-EOS run; then subsets on the GEMV path (2 sequences) and batches in which every utterance stops (the loop
    must end within `eos_run_ahead` steps of the last EOS)."""
    lib = _lib.load()
    d = peaked_checkpoint("/tmp/q3a_ckpt_tinyu_peaked", "tiny_untied", seed=5)
    B, kmax = 40, 9
    classes = [i % 7 for i in range(B)]
    clips = [synthetic.synthetic_clip(200 + i, 1.0 + 0.35 * classes[i]) for i in range(B)]
    stops, routers, info = plan_class_stops(d, clips, classes, [1, None, 3, 5, 7, None, 2], kmax)
    ref, margins = free_run_margins(d, clips, kmax)   # the oracle, re-loaded AFTER planting, natural-EOS mode: the expectation
    assert [len(r) for r in ref] == [kmax if k is None else k for k in stops]
    dec_bf16 = [len(r) if min(m) >= BF16_MARGIN else 0 for r, m in zip(ref, margins)]
    print(f"[eos] tiny peaked: stops {stops[:7]} x {B // 7}+; planted row norms {[round(n) for n in info['row_norms']]}; natural peak "
          f"{info['natural_peak']:.1f}; smallest margin of any step {min(min(m) for m in margins):.2f}; bf16-decidable utterances {sum(1 for x in dec_bf16 if x)}")
    assert sum(1 for x in dec_bf16 if x) >= 36
    everything = [len(r) for r in ref]
    runs = {}
    for precise in (True, False):
        for use_graph in (True, False):
            got, steps = _run(d, clips, kmax, precise=precise, use_graph=use_graph)
            runs[precise, use_graph] = got
            n = _check(f"tiny B=40 precise={precise} graph={use_graph}", got, ref, stops, everything if precise else dec_bf16, exact=precise)
            print(f"[eos] tiny B=40 precise={precise} graph={use_graph}: {n} utterances compared with the oracle")
            assert n == B if precise else n >= 36
            assert steps == kmax - 1   # a never-EOS utterance runs to the cap
        assert runs[precise, True] == runs[precise, False], "hipGraph replay and eager launches disagree"
    # every utterance of the batch stops: the device-side all-done flag ends the loop -- default mode and precise mode
    fin = [u for u in range(B) if stops[u] is not None]
    try:
        for ahead in (1, 2, 4):
            assert lib.q3a_debug_set(b"eos_run_ahead", ahead) == 0
            for sel in (fin[:12], fin[:2], [fin[2]]):   # skinny path, GEMV path with 2 sequences, one sequence
                sub = [clips[u] for u in sel]
                last = max(stops[u] for u in sel)
                for precise in (True, False):
                    got, steps = _run(d, sub, kmax, precise=precise)
                    n = _check(f"tiny subset {sel} ahead={ahead} precise={precise}", got, [ref[u] for u in sel], [stops[u] for u in sel],
                               [everything[u] if precise else dec_bf16[u] for u in sel], exact=precise)
                    assert n == len(sel)
                    assert last <= steps <= min(last + ahead, kmax - 1), (sel, ahead, last, steps)
    finally:
        lib.q3a_debug_set(b"eos_run_ahead", 1)


def test_ragged_eos_inside_a_batch_0p6b_dims_default_mode():
    """The same at the 0.6B dimensions (untied lm_head; token-embedding scale 4 so that 28 layers of residual do not drown the
    token just fed) in the DEFAULT bf16 mode: 8 utterances in 4 length classes (stops 3 / never / 4 / 6) on the skinny MFMA path,
    every one compared with the oracle; then all-stop subsets on the skinny and the GEMV path."""
    cfg = {"audio_config": dict(synthetic.CONFIG_0P6B["audio_config"]),
           "text_config": dict(synthetic.CONFIG_0P6B["text_config"], tie_word_embeddings=False)}
    d = peaked_checkpoint("/tmp/q3a_ckpt_0p6bu_peaked", "0.6b", seed=3, cfg=cfg, embed_scale=4.0)
    kmax = 8
    classes = [0, 1, 2, 3, 0, 1, 2, 3]
    clips = [synthetic.synthetic_clip(300 + i, 1.5 + 0.5 * classes[i]) for i in range(8)]
    stops, routers, info = plan_class_stops(d, clips, classes, [3, None, 4, 6], kmax)
    ref, margins = free_run_margins(d, clips, kmax)
    assert [len(r) for r in ref] == [kmax if k is None else k for k in stops]
    print(f"[eos] 0.6B dims peaked: stops {stops}; planted row norms {[round(n) for n in info['row_norms']]}; natural peak {info['natural_peak']:.1f}; "
          f"smallest margin of any step {min(min(m) for m in margins):.2f}")
    assert min(min(m) for m in margins) >= BF16_MARGIN
    everything = [len(r) for r in ref]
    for use_graph in (True, False):
        got, steps = _run(d, clips, kmax, use_graph=use_graph)
        n = _check(f"0.6B-dims B={len(clips)} graph={use_graph}", got, ref, stops, everything, exact=False)
        assert n == len(clips), n
        assert steps == kmax - 1
    fin = [u for u in range(len(clips)) if stops[u] is not None]
    for sub in (fin, fin[:2]):
        last = max(stops[u] for u in sub)
        got, steps = _run(d, [clips[u] for u in sub], kmax)
        n = _check(f"0.6B-dims subset {sub}", got, [ref[u] for u in sub], [stops[u] for u in sub], [everything[u] for u in sub], exact=False)
        assert n == len(sub)
        assert last <= steps <= min(last + 1, kmax - 1), (sub, last, steps)

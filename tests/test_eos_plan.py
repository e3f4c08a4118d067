"""CPU test of the ragged-EOS fixture (tests/eos_plan.py + synthetic.plant_eos): after planting, the fp32 oracle run in
the reference's natural-EOS mode (src/inference.rs:160-167) stops every utterance at the planned step, EOS excluded from
the generated ids, and utterances of one batch stop at different steps."""
import numpy as np

from oracle import q3asr_oracle as O
from qwen3_asr_rs_amd import synthetic

from eos_plan import fresh_eos_checkpoint, plan_ragged_eos


def test_planted_checkpoint_stops_the_oracle_at_the_planned_steps(tmp_path):
    d = fresh_eos_checkpoint(str(tmp_path / "ckpt"), "tiny_untied", seed=5)
    clips = [synthetic.synthetic_clip(200 + i, 1.0 + 0.17 * (i % 7)) for i in range(8)]
    kmax = 7
    stops, decidable, info = plan_ragged_eos(d, clips, kmax, [None, 5, 3, 2, 4, 1, 6, 3], margin_min=1e-3)
    assert None in stops and len({s for s in stops if s is not None}) >= 3, stops
    assert info["row_norm"] < 200.0, info["row_norm"]   # a huge row would amplify a bf16 engine's rounding noise
    orc = O.AsrOracle(d)   # re-loaded from the rewritten checkpoint
    for clip, k in zip(clips, stops):
        r = orc.transcribe_ids(clip, max_new_tokens=kmax)
        assert len(r.ids) == (kmax if k is None else k)
        assert synthetic.ENDOFTEXT_ID not in r.ids                      # generated_ids excludes the EOS (inference.rs:163-167)
        if k is not None:
            assert r.all_step_ids[-1] == synthetic.ENDOFTEXT_ID and len(r.all_step_ids) == k + 1
    # the tensor helpers round-trip
    key = synthetic.output_embedding_key(d)
    assert key == "thinker.lm_head.weight"
    row = synthetic.read_tensor(d, key)[synthetic.ENDOFTEXT_ID]
    assert abs(float(np.linalg.norm(row)) - info["row_norm"]) < 1e-3 * info["row_norm"]


def test_peaked_fixture_walks_routes_and_stops_by_class(tmp_path):
    """The round-4 fixture (peaked_checkpoint + plan_class_stops): the greedy successor of a token is walk_next, every length
    class is routed down its own walk at step 0, the classes stop at their planted steps, and every step of every utterance
    has a top-1/top-2 margin a bf16 engine can decide."""
    from eos_plan import free_run_margins, peaked_checkpoint, plan_class_stops, walk_next
    d = peaked_checkpoint(str(tmp_path / "peaked"), "tiny_untied", seed=5)
    B, kmax = 14, 8
    classes = [i % 7 for i in range(B)]
    clips = [synthetic.synthetic_clip(200 + i, 1.0 + 0.35 * classes[i]) for i in range(B)]
    class_stops = [1, None, 3, 5, 7, None, 2]
    stops, routers, info = plan_class_stops(d, clips, classes, class_stops, kmax)
    ids, margins = free_run_margins(d, clips, kmax)
    V = 151936
    for u in range(B):
        k = classes[u]
        assert len(ids[u]) == (kmax if class_stops[k] is None else class_stops[k])
        assert ids[u][0] == routers[k]
        assert all(ids[u][s] == walk_next(ids[u][s - 1], V) for s in range(1, len(ids[u])))
        assert min(margins[u]) >= 1.0, (u, margins[u])
    assert max(info["row_norms"]) < 200.0, info   # longer rows would amplify a bf16 engine's rounding noise beyond the margins

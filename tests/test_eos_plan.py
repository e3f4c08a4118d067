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

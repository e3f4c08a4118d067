"""Test helper: checkpoints whose utterances stop at chosen, different steps (ragged natural EOS inside a batch).

The reference's greedy loop breaks each utterance at ITS first EOS and excludes the EOS from generated_ids
(src/inference.rs:160-167).  Random-init checkpoints never emit EOS, so `plan_ragged_eos` runs the fp32 oracle once per
utterance (EOS ignored), picks the step at which every utterance shall stop, and has synthetic.plant_eos rewrite the
<|endoftext|> row of the output embedding accordingly.  The oracle is then RE-LOADED from the modified checkpoint by
the caller and run in natural-EOS mode: that run, not the plan, is what the engine is compared with.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from oracle import q3asr_oracle as O
from qwen3_asr_rs_amd import synthetic


def plan_ragged_eos(model_dir: str, clips: Sequence[np.ndarray], kmax: int, targets: Sequence[Optional[int]],
                    margin_min: float):
    """targets[u]: step at which utterance u shall emit EOS, None = never within kmax tokens.

    With the large-embedding checkpoints of fresh_eos_checkpoint the decoder state at step s is (nearly) a function of the
    tokens fed so far, so utterances that share a token history have nearly identical states there and MUST share the
    decision "EOS here or not" -- a row that told them apart would need a huge norm, i.e. amplify a bf16 engine's
    rounding noise.  Stops are therefore planned on the trie of token histories: a wanted step is moved (later first, then
    earlier, finally to whatever the trie already dictates) until
      * its history node has not been passed quietly by another utterance and none of its prefixes already fires, and
      * every step in front of it has a top-1/top-2 logit margin of at least `margin_min` (ids in front of the stop must
        be decidable for a bf16 engine).
    Step 0 (empty history: the same last prompt token for everyone) is shared by the whole batch and never planned.
    Returns (stops, decidable, info): stops[u] = planted step or None, decidable[u] = number of leading steps whose greedy
    id is decidable at `margin_min`, info = synthetic.plant_eos's report (+ info["margins"][u] = the top-1/top-2 margin of
    every step, for callers that compare an engine with a coarser rounding than the one the plan was made for)."""
    orc = O.AsrOracle(model_dir)
    node = {}   # token history -> "quiet" | "fire"
    plan_states, plan_fire = [], []
    stops, decidable, all_margins = [None] * len(clips), [0] * len(clips), [None] * len(clips)
    # late stops and "never" first: they lay down quiet nodes; early stops then take what is left of the shared prefixes
    order = sorted(range(len(clips)), key=lambda u: -(kmax if targets[u] is None else int(targets[u])))
    for u in order:
        clip, want = clips[u], targets[u]
        r = orc.transcribe_ids(clip, fixed_new_tokens=kmax, want_hidden=True, last_only=True)
        assert len(r.step_hidden) >= kmax
        toks = r.all_step_ids[:kmax]
        keys = [tuple(toks[:s]) for s in range(kmax)]   # history behind the decision of step s
        margins = []
        for lg in r.step_logits:
            top = lg.topk(2).values
            margins.append(float(top[0] - top[1]))
        lim = next((s for s in range(kmax) if margins[s] < margin_min), kmax)
        forced = next((s for s in range(kmax) if node.get(keys[s]) == "fire"), None)   # the trie already stops this path
        if forced is not None:
            k = forced
        else:
            k = None
            if want is not None:
                for c in list(range(max(int(want), 1), kmax)) + list(range(int(want) - 1, 0, -1)):
                    if c <= lim and node.get(keys[c]) is None:
                        k = c
                        break
        stops[u], decidable[u], all_margins[u] = k, lim, margins
        for s in range(kmax if k is None else k + 1):
            fire = k is not None and s == k
            node[keys[s]] = "fire" if fire else "quiet"
            plan_states.append(r.step_hidden[s].numpy())
            plan_fire.append(fire)
    info = synthetic.plant_eos(model_dir, np.stack(plan_states), plan_fire)
    assert info["worst_fire"] is None or info["worst_fire"] > 8.0, info
    assert info["worst_quiet"] is None or info["worst_quiet"] < -8.0, info
    info["margins"] = all_margins
    return stops, decidable, info


def leading_decidable(margins: Sequence[float], margin_min: float) -> int:
    """Number of leading steps whose top-1/top-2 margin is at least `margin_min`."""
    return next((s for s, m in enumerate(margins) if m < margin_min), len(margins))


def fresh_eos_checkpoint(model_dir: str, preset: str, seed: int, cfg: Optional[dict] = None, embed_scale: float = 0.5) -> str:
    """A checkpoint directory of its own for planting (the shared /tmp/q3a_ckpt_* directories must stay untouched): the
    EOS row is reset to zero so a re-run of the planning sees the never-EOS trajectories again."""
    synthetic.write_checkpoint(model_dir, preset, seed=seed, cfg=cfg, embed_scale=embed_scale)
    key = synthetic.output_embedding_key(model_dir)
    cols = synthetic._locate_tensor(model_dir, key)[1]["shape"][1]
    synthetic.overwrite_row(model_dir, key, synthetic.ENDOFTEXT_ID, np.zeros(cols, dtype=np.float32))
    return model_dir

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


# ---- round 4: peaked logits + per-class trajectories ---------------------------------------------------------------------
# Random-init logits are nearly flat (top-1/top-2 gaps of a few hundredths next to a bf16 engine's logit error of ~1e-2), so in
# the default mode almost no utterance of the fixtures above is decidable at every step.  Real checkpoints are peaked.  The
# fixture below makes a synthetic checkpoint peaked without touching anything upstream of the lm_head:
#   * the untied lm_head becomes gamma * embed[next^-1(v)] + the random head, with next(t) = (A t + C) mod V a permutation of the
#     vocabulary: with a large token-embedding scale the last decoder state is dominated by the token just fed, so the logit of
#     next(t) towers over the rest (gaps of 1.4-5 at logits of ~6) and the greedy trajectory is the walk t -> next(t) -> ...;
#   * all utterances share the prompt's last token, hence the walk -- so "router" rows are planted too: the step-0 states of
#     clips of different lengths differ (another prompt length, other RoPE phases) by ~20 % of their norm, enough for a row per
#     length class that wins at step 0 for its class only (row norms 40-130) and sends the class down its own walk;
#   * the <|endoftext|> row is planted to win at each class's stop step.
# Every state the free run will visit is known BEFORE planting: hidden states depend on the tokens fed, not on the lm_head, so
# the oracle is teacher-forced along each class's walk to collect them.
WALK_A, WALK_C = 48271, 12345


def walk_next(t: int, vocab: int) -> int:
    return (int(t) * WALK_A + WALK_C) % vocab


def peaked_checkpoint(model_dir: str, preset: str, seed: int, cfg: Optional[dict] = None, embed_scale: float = 0.5,
                      peak: float = 14.0) -> str:
    """Untied-lm_head checkpoint whose greedy successor of token t is walk_next(t) (see above).  embed_scale: std of the token
    embedding (large: the residual stream of all layers must not drown the token just fed -- 0.5 is enough for the 2-3 layers of
    the tiny presets, the 28 layers of the 0.6B dimensions need 4; larger values make the step-0 states of the length classes
    more alike, i.e. the router rows longer and a bf16 engine's noise on their logits larger); peak: logit the successor reaches if the normed last state
    were the token's embedding direction exactly (gamma = peak / (hidden * embed_scale)).  Re-created on every call: the planting
    below rewrites rows of the head."""
    import shutil
    shutil.rmtree(model_dir, ignore_errors=True)
    synthetic.write_checkpoint(model_dir, preset, seed=seed, cfg=cfg, embed_scale=embed_scale)
    assert synthetic.output_embedding_key(model_dir) == "thinker.lm_head.weight", "peaked_checkpoint needs an untied lm_head"
    emb = synthetic.read_tensor(model_dir, "thinker.model.embed_tokens.weight")
    head = synthetic.read_tensor(model_dir, "thinker.lm_head.weight")
    V = emb.shape[0]
    nxt = (np.arange(V, dtype=np.int64) * WALK_A + WALK_C) % V
    inv = np.empty(V, dtype=np.int64)
    inv[nxt] = np.arange(V)
    gamma = peak / (emb.shape[1] * embed_scale)
    new = gamma * emb[inv] + head
    new[synthetic.ENDOFTEXT_ID] = 0.0
    synthetic.overwrite_tensor(model_dir, "thinker.lm_head.weight", new.astype(np.float32))
    return model_dir


def plan_class_stops(model_dir: str, clips: Sequence[np.ndarray], classes: Sequence[int], class_stops: Sequence[Optional[int]],
                     kmax: int, hi: float = 30.0, lo: float = -10.0):
    """classes[u]: class of utterance u (utterances of one class must have near-identical step-0 states: same clip length);
    class_stops[k]: step at which class k emits EOS (None: never within kmax).  Plants one router row per class and the EOS row.
    Returns (stops per utterance, router token per class, info)."""
    orc = O.AsrOracle(model_dir)
    V = orc.cfg.text.vocab_size
    n_cls = max(classes) + 1
    routers = [1000 + 977 * k for k in range(n_cls)]
    walks = []
    for k in range(n_cls):
        w = [routers[k]]
        while len(w) < kmax:
            w.append(walk_next(w[-1], V))
        assert synthetic.ENDOFTEXT_ID not in w and 151645 not in w
        walks.append(w)
    states, owner = [], []   # owner[i] = (utterance, step)
    peak = 0.0
    for u, clip in enumerate(clips):
        k = classes[u]
        last = kmax - 1 if class_stops[k] is None else class_stops[k]
        r = orc.transcribe_ids(clip, forced_ids=walks[k][:last], want_hidden=True, last_only=True)
        assert len(r.step_hidden) >= last + 1
        for s in range(last + 1):
            states.append(r.step_hidden[s].numpy())
            owner.append((u, s))
        peak = max(peak, max(float(l.max()) for l in r.step_logits[:last + 1]))
    H = np.stack(states).astype(np.float64)
    key = synthetic.output_embedding_key(model_dir)
    norms, worst_hi, worst_lo = [], np.inf, -np.inf
    rows = [(routers[k], [classes[u] == k and s == 0 for u, s in owner]) for k in range(n_cls)]
    rows.append((synthetic.ENDOFTEXT_ID, [class_stops[classes[u]] is not None and s == class_stops[classes[u]] for u, s in owner]))
    for tok, fire in rows:
        fire = np.asarray(fire, dtype=bool)
        if not fire.any():
            continue
        t = np.where(fire, hi, lo)
        w, *_ = np.linalg.lstsq(H, t, rcond=1e-4)
        stored = synthetic.overwrite_row(model_dir, key, tok, w.astype(np.float32))
        got = H @ stored.astype(np.float64)
        norms.append(float(np.linalg.norm(stored)))
        worst_hi, worst_lo = min(worst_hi, float(got[fire].min())), max(worst_lo, float(got[~fire].max()) if (~fire).any() else -np.inf)
    info = {"row_norms": norms, "worst_hi": worst_hi, "worst_lo": worst_lo, "natural_peak": peak, "routers": routers}
    assert worst_hi > peak + 10.0 and worst_lo < -2.0, info
    return [class_stops[k] for k in classes], routers, info


def free_run_margins(model_dir: str, clips: Sequence[np.ndarray], kmax: int):
    """Natural-EOS oracle run on the (planted) checkpoint: ids and the top-1/top-2 margin of every step, per utterance."""
    orc = O.AsrOracle(model_dir)
    ids, margins = [], []
    for c in clips:
        r = orc.transcribe_ids(c, max_new_tokens=kmax, last_only=True)
        ids.append(r.ids)
        margins.append([float(l.topk(2).values[0] - l.topk(2).values[1]) for l in r.step_logits])
    return ids, margins

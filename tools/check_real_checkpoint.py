#!/usr/bin/env python
"""Real-checkpoint check of the drop-in claim: the reference's only known answers are the three clips under test_audio/
and their transcripts (test_audio/sample{1,2,3}.txt; .github/workflows/ci.yml:129-163 runs the binary on them,
src/inference.rs:89-213 is the path).  No checkpoint exists on the build / GPU boxes, so this harness is what a user WITH
the weights runs:

    python tools/check_real_checkpoint.py /path/to/Qwen3-ASR-0.6B [--device 0] [--no-oracle] [--max-new 256]

For each clip (24 kHz WAV -> 16 kHz with this backend's resampler, the parity input of SURVEY.md section 0 item 9):
  * HIP engine, default (bf16) mode and precise mode: greedy ids until EOS, decoded text (tokenizer.json);
  * fp32 CPU oracle (the reference's tch-CPU op sequence): greedy ids  [skipped with --no-oracle];
  * checks: ids(default) == ids(precise) == ids(oracle)  ("token-exact greedy", BASELINE.json north_star);
            text == sample*.txt after whitespace normalisation (what the reference prints as `Text:`).
Prints one JSON report; exit status 0 iff every check passed.  tests/test_real_checkpoint.py wraps it for pytest
(-m gpu, skipped unless $Q3A_MODEL_DIR is set).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
AUDIO = os.path.join(ROOT, "tests", "golden", "test_audio")


def norm(s: str) -> str:
    return " ".join(s.split())


def run(model_dir: str, device: int = 0, oracle: bool = True, max_new: int = 256, clips=(1, 2, 3)) -> dict:
    from qwen3_asr_rs_amd.audio import load_audio
    from qwen3_asr_rs_amd.engine import AsrInference
    if not os.path.exists(os.path.join(model_dir, "tokenizer.json")):
        raise SystemExit(f"{model_dir}: tokenizer.json missing (needed to decode text, src/tokenizer.rs:11-30)")
    report = {"model_dir": model_dir, "clips": [], "ok": True}
    engines = {"default": AsrInference.load(model_dir, device, max_new_tokens=max_new),
               "precise": AsrInference.load(model_dir, device, precise=True, max_new_tokens=max_new)}
    orc = None
    if oracle:
        from oracle import q3asr_oracle as O
        orc = O.AsrOracle(model_dir)
    for i in clips:
        wav = os.path.join(AUDIO, f"sample{i}.wav")
        want = norm(open(os.path.join(AUDIO, f"sample{i}.txt"), encoding="utf-8").read())
        samples = load_audio(wav, 16000)
        entry = {"clip": f"sample{i}.wav", "seconds": round(len(samples) / 16000.0, 3), "expected_text": want}
        ids = {}
        for mode, inf in engines.items():
            t0 = time.time()
            r = inf.transcribe(samples, None, max_new_tokens=max_new)
            ids[mode] = r.ids
            entry[mode] = {"language": r.language, "text": r.text, "n_ids": len(r.ids), "seconds": round(time.time() - t0, 3),
                           "text_matches_reference_transcript": norm(r.text) == want}
        if orc is not None:
            t0 = time.time()
            ro = orc.transcribe_ids(samples, max_new_tokens=max_new, keep_logits=False, last_only=True)
            ids["oracle"] = ro.ids
            entry["oracle"] = {"n_ids": len(ro.ids), "seconds": round(time.time() - t0, 1),
                               "text": engines["default"].tokenizer.decode(ro.ids, True)}
        entry["ids_default_eq_precise"] = ids["default"] == ids["precise"]
        if orc is not None:
            entry["ids_default_eq_oracle"] = ids["default"] == ids["oracle"]
            entry["ids_precise_eq_oracle"] = ids["precise"] == ids["oracle"]
            if not entry["ids_default_eq_oracle"]:
                k = next((j for j, (a, b) in enumerate(zip(ids["default"], ids["oracle"])) if a != b), min(len(ids["default"]), len(ids["oracle"])))
                entry["first_divergence_default_vs_oracle"] = k
        checks = [entry["ids_default_eq_precise"], entry["default"]["text_matches_reference_transcript"],
                  entry["precise"]["text_matches_reference_transcript"]]
        if orc is not None:
            checks += [entry["ids_default_eq_oracle"], entry["ids_precise_eq_oracle"]]
        entry["ok"] = all(checks)
        report["ok"] = report["ok"] and entry["ok"]
        report["clips"].append(entry)
    for inf in engines.values():
        inf.engine.close()
    return report


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("model_dir")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--no-oracle", action="store_true", help="skip the fp32 CPU oracle (about a minute per clip)")
    ap.add_argument("--max-new", type=int, default=256)
    a = ap.parse_args()
    rep = run(a.model_dir, a.device, not a.no_oracle, a.max_new)
    print(json.dumps(rep, ensure_ascii=False, indent=1))
    return 0 if rep["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())

"""Opt-in real-weight check (pytest -m gpu with Q3A_MODEL_DIR=/path/to/Qwen3-ASR-0.6B): the three reference clips
through the HIP engine (default + precise) and the fp32 oracle -- ids token-exact across the three, text equal to
test_audio/sample*.txt.  Skipped when no checkpoint directory is given (none exists on the build / GPU boxes:
transcript parity on real weights is otherwise UNMEASURED).  Reference: src/inference.rs:89-213,
.github/workflows/ci.yml:129-163."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

MODEL_DIR = os.environ.get("Q3A_MODEL_DIR") or os.environ.get("Q3A_REAL_CKPT")


@pytest.mark.gpu
@pytest.mark.skipif(not MODEL_DIR, reason="set Q3A_MODEL_DIR to a real Qwen3-ASR checkpoint directory (config.json, *.safetensors, tokenizer.json)")
def test_reference_clips_token_exact_and_transcripts():
    import check_real_checkpoint as H
    rep = H.run(MODEL_DIR, 0, oracle=os.environ.get("Q3A_REAL_NO_ORACLE") is None)
    for c in rep["clips"]:
        assert c["ids_default_eq_precise"], c
        assert c["default"]["text_matches_reference_transcript"], (c["default"]["text"], c["expected_text"])
        if "ids_default_eq_oracle" in c:
            assert c["ids_default_eq_oracle"] and c["ids_precise_eq_oracle"], c


def test_harness_imports_and_normalises():
    """CPU: the harness module loads without a GPU and its transcript normalisation is whitespace-only."""
    import check_real_checkpoint as H
    assert H.norm("  a  b\n c ") == "a b c"
    for i in (1, 2, 3):
        assert os.path.exists(os.path.join(H.AUDIO, f"sample{i}.wav")) and os.path.exists(os.path.join(H.AUDIO, f"sample{i}.txt"))

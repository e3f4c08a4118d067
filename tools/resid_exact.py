"""gemm256's fp32-residual epilogue with and without the residual prefetch (q3a_debug_set gemm256_resid_prefetch) must give the SAME
bits: 32 clips through mel + encoder + prefill + 2 tokens, last hidden rows and ids compared.  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from qwen3_asr_rs_amd import synthetic, _lib
from qwen3_asr_rs_amd.engine import HipEngine
d = synthetic.write_checkpoint("/tmp/q3a_ckpt_0p6b_pipe", "0.6b", seed=0)
clips = [synthetic.synthetic_clip(i, 30.0) for i in range(32)]
eng = HipEngine(d, 0, max_new_tokens=16, debug_taps=2)
lib = _lib.load()
out = {}
for v in (0, 1, 0, 1):
    assert lib.q3a_debug_set(b"gemm256_resid_prefetch", v) == 0
    ids = eng.transcribe_batch(clips, None, max_new=2, fixed_new_tokens=2)
    h = eng.debug_read("dec_last_hidden").view(np.uint32).copy()
    if v in out:
        assert (out[v][0] == h).all() and out[v][1] == ids, "not repeatable"
    out[v] = (h, ids)
same = bool((out[0][0] == out[1][0]).all()) and out[0][1] == out[1][1]
print("RESID_EXACT prefetch 0 vs 1: last hidden rows and ids", "IDENTICAL" if same else "DIFFER", flush=True)
lib.q3a_debug_set(b"gemm256_resid_prefetch", 1)
eng.close()
sys.exit(0 if same else 1)

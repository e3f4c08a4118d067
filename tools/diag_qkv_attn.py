"""Diagnostics of the fused qkv + attention launch (q3a_debug_set "fuse_qkv_attn"): where did the workgroups of each kv
head run (XCC_ID), how many waits ran out, what had arrived by then; decode time and ids against the separate launches.
    python tools/diag_qkv_attn.py      (0.6B dims, one 30 s clip, 50 tokens)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qwen3_asr_rs_amd import _lib, synthetic  # noqa: E402
from qwen3_asr_rs_amd.engine import HipEngine  # noqa: E402

lib = _lib.load()
d = synthetic.write_checkpoint("/tmp/q3a_ckpt_0p6b", "0.6b", seed=0)
clip = synthetic.synthetic_clip(0, 30.0)
out = {}
for fuse in (1, 0, 1):
    assert lib.q3a_debug_set(b"fuse_qkv_attn", fuse) == 0
    eng = HipEngine(d, 0, debug_taps=True, max_new_tokens=50)
    eng.transcribe_batch([clip], None, max_new=50, fixed_new_tokens=50)   # warm (graph capture)
    ids = eng.transcribe_batch([clip], None, max_new=50, fixed_new_tokens=50)[0]
    t = eng.timings()
    out[fuse] = ids
    print(f"fuse_qkv_attn={fuse}: decode {t['decode_ms']:.2f} ms for {t['decode_steps']} steps = {1e3 * t['decode_ms'] / max(1, t['decode_steps']):.1f} us per step")
    if fuse:
        w = np.frombuffer(eng.debug_read("xcd_sync").tobytes(), dtype=np.uint32).reshape(2, 8, 64)
        for g in range(8):
            xcc = sorted(set((w[1, g, :32].astype(int) - 1).tolist()))
            print(f"  kv head {g}: arrivals {w[0, g, 0]} departures {w[0, g, 32]} | waits that ran out {w[1, g, 32]} (seen {w[1, g, 33:37].tolist()}) | XCC ids {xcc}")
    eng.close()
print("ids equal:", out[1] == out[0])

"""Diagnostics of the fused qkv + attention launch (Q3A_FUSE_QKV_ATTN=1): where did the workgroups of each kv head run
(XCC_ID), how many waits ran out, what had arrived by then.  python tools/diag_qkv_attn.py  (0.6B dims, one 30 s clip)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["Q3A_FUSE_QKV_ATTN"] = "1"
from qwen3_asr_rs_amd import synthetic  # noqa: E402
from qwen3_asr_rs_amd.engine import HipEngine  # noqa: E402

d = synthetic.write_checkpoint("/tmp/q3a_ckpt_0p6b", "0.6b", seed=0)
for graph in (False, True):
    eng = HipEngine(d, 0, debug_taps=True, use_graph=graph, max_new_tokens=8)
    clip = synthetic.synthetic_clip(0, 30.0)
    eng.mel([clip]); eng.encode()
    eng.prefill([HipEngine.build_prompt(eng.num_audio_tokens(len(clip)))])
    t0 = time.time()
    eng.decode_step()
    t1 = time.time()
    eng.decode_step()
    t2 = time.time()
    w = np.frombuffer(eng.debug_read("xcd_sync").tobytes(), dtype=np.uint32).reshape(2, 8, 64)
    print(f"graph={graph}: decode steps took {1e3 * (t1 - t0):.1f} / {1e3 * (t2 - t1):.1f} ms")
    for g in range(8):
        xcc = (w[1, g, :32].astype(int) - 1).tolist()
        print(f"  kv head {g}: arrivals {w[0, g, 0]} departures {w[0, g, 32]} | waits that ran out {w[1, g, 32]} seen {w[1, g, 33:37].tolist()}"
              f" | XCC ids of its 32 workgroups: {sorted(set(xcc))}")
    eng.close()

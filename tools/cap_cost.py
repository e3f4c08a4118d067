"""What a generous max_new_tokens costs (one 30 s clip, 100 tokens generated, 0.6B dims): run on the GPU box (the one-sequence decode
attention launches the key splits the caches HOLD keys for; round 5 measured the alternative behind a knob that is gone)."""
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
from qwen3_asr_rs_amd import synthetic
from qwen3_asr_rs_amd.engine import HipEngine
d = synthetic.write_checkpoint("/tmp/q3a_ckpt_0p6b_cap", "0.6b", seed=0)
clip = synthetic.synthetic_clip(0, 30.0)
for cap in (100, 512, 4096):
    eng = HipEngine(d, 0, max_new_tokens=cap)
    eng.upload_pcm([clip])
    for _ in range(2): eng.run_resident(None, cap, 100)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): eng.run_resident(None, cap, 100); ids = eng.fetch_ids(cap)
    dt = (time.perf_counter() - t0) / 5
    print(f"max_new_tokens {cap:5d}: {dt*1e3:7.2f} ms per clip, {30.0/dt:6.1f} audio-s/s, {len(ids[0])} tokens, decode_ms {eng.timings()['decode_ms']:.2f}", flush=True)
    eng.close()

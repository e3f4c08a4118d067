import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from qwen3_asr_rs_amd import _lib
lib = _lib.load()
# fixed-N reference on this box
from qwen3_asr_rs_amd import synthetic
from qwen3_asr_rs_amd.engine import HipEngine
d = synthetic.write_checkpoint("/tmp/q3a_ckpt_0p6b_pipe", "0.6b", seed=0, embed_scale=synthetic.PEAKED_EMBED_SCALE)
clip = synthetic.synthetic_clip(0, 30.0)
eng = HipEngine(d, 0, max_new_tokens=100)
eng.upload_pcm([clip])
for _ in range(3): eng.run_resident(None, 0, 100)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(8): eng.run_resident(None, 0, 100); eng.fetch_ids(100)
fixed_ms = (time.perf_counter() - t0) / 8 * 1e3
print("fixed-N ms per clip", round(fixed_ms, 3), eng.timings(), flush=True)
eng.close()
for ahead in (2, 3, 4, 1):
    lib.q3a_debug_set(b"eos_run_ahead", ahead)
    r = bench.natural_eos_leg("0.6b", 30.0, 100, 8, 3, False, fixed_ms)
    print("ahead", ahead, {k: r[k] for k in ("ms_per_step", "vs_fixed_n_ms", "decode_steps_executed", "generated_tokens")}, flush=True)

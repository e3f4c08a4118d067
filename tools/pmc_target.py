"""Small workload for `rocprofv3 --pmc ...` passes (PMC collection serialises every dispatch, so the full
bench is far too long): one 30 s clip, 4 generated tokens, then two sweeps of the dominant decode kernel
(qkv + gate/up GEMVs of all layers, streamed from HBM) -- the same launches bench.py times for `roofline`.
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d OUT -o fetch -- python tools/pmc_target.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d OUT -o write -- python tools/pmc_target.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qwen3_asr_rs_amd import synthetic  # noqa: E402
from qwen3_asr_rs_amd.engine import HipEngine  # noqa: E402

d = synthetic.write_checkpoint("/tmp/q3a_ckpt_0p6b", "0.6b", seed=0)
eng = HipEngine(d, 0, max_new_tokens=16)
eng.upload_pcm([synthetic.synthetic_clip(0, 30.0)])
eng.run_resident(None, 0, 4)
print(eng.profile_weight_stream(reps=2))
eng.close()

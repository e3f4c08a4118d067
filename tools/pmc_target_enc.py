"""Small workload for `rocprofv3 --pmc ...` passes over the encoder / prefill GEMMs (MFMA utilisation): one batch of
8 x 30 s clips through mel + encoder + prefill, no decode loop (PMC collection serialises every dispatch).
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d OUT -o mfma -- python tools/pmc_target_enc.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qwen3_asr_rs_amd import synthetic  # noqa: E402
from qwen3_asr_rs_amd.engine import HipEngine  # noqa: E402

B = int(os.environ.get("PMC_BATCH", "8"))
d = synthetic.write_checkpoint("/tmp/q3a_ckpt_0p6b", "0.6b", seed=0)
eng = HipEngine(d, 0, max_new_tokens=16)
eng.upload_pcm([synthetic.synthetic_clip(i, 30.0) for i in range(B)])
eng.run_resident(None, 0, 1)
print(eng.timings())
eng.close()

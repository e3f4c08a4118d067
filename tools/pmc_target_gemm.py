"""Target for `rocprofv3 --pmc ...` passes over the 256x256 GEMM alone (k_gemm256.hip forced): a few launches each of a
long-K square problem and of the encoder's short-K qkv shape.
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY ... --kernel-trace -d OUT -o sq -- python tools/pmc_target_gemm.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qwen3_asr_rs_amd import _lib  # noqa: E402
from qwen3_asr_rs_amd.engine import selftest_gemm16  # noqa: E402

lib = _lib.load()
lib.q3a_debug_set(b"gemm256_min_tiles", 0)
for shp in ((4096, 4096, 4096), (12480, 2688, 896), (12480, 896, 896)):
    print(shp, selftest_gemm16(*shp, reps=0))

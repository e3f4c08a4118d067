"""Exactness + speed probe of the bf16-activation GEMM (k_gemm16.hip) on the model's shapes, run on the GPU box:
    python tools/gemm16_probe.py            # batch-1 shapes then batch-32 shapes
err is the max abs difference against the fp64-accumulating device reference, us_f32 the fp32-activation kernel."""
import sys

sys.path.insert(0, ".")
from qwen3_asr_rs_amd.engine import selftest_gemm16  # noqa: E402

B1 = [(390, 2688, 896), (390, 896, 896), (390, 3584, 896), (390, 896, 3584), (390, 896, 7680), (390, 1024, 896),
      (405, 4096, 1024), (405, 1024, 2048), (405, 1024, 3072)]
B32 = [(12480, 3584, 896), (12480, 896, 3584), (12960, 4096, 1024), (12960, 1024, 3072), (8192, 8192, 8192)]
SMALL = [(128, 128, 64), (130, 200, 96), (70, 96, 256), (33, 36, 512)]
for shp in SMALL + B1 + B32:
    try:
        r = selftest_gemm16(*shp, reps=5 if shp[0] > 1000 else 20)
        print(shp, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}, flush=True)
    except Exception as e:  # noqa: BLE001
        print(shp, "EXC", e, flush=True)

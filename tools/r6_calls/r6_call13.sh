#!/bin/bash
# round 6, GPU call 13: one-sequence GEMVs fetch x (and the norm weight) once per workgroup through LDS (bits: 1 norm-fused qkv / gate-up,
# 2 down, 4 lm_head) instead of once per wave
python tools/ab_knobs.py --preset 0.6b --batch 1 --rounds 9 base Q3A_GEMV_X_LDS=1 Q3A_GEMV_X_LDS=2 Q3A_GEMV_X_LDS=3 Q3A_GEMV_X_LDS=7 > gpurun_out/r6_ab_gemv_x_lds.txt 2>&1
python tools/ab_knobs.py --preset 1.7b --batch 1 --rounds 5 base Q3A_GEMV_X_LDS=1 Q3A_GEMV_X_LDS=3 Q3A_GEMV_X_LDS=7 >> gpurun_out/r6_ab_gemv_x_lds.txt 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r6_ab_gemv_x_lds.txt"):
    if l.startswith("{"):
        j = json.loads(l); print(f'{j["setting"]:28s} {j["decode_us_per_step"]:8.2f} us/step  {j["ms_per_batch"]:8.3f} ms  ids equal {j["ids_equal_to_first_setting"]} crc {j["ids_crc32"]}')
    elif "rror" in l: print(l.strip()[:300])
PY

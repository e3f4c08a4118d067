#!/bin/bash
# round 6, GPU call 5: o_proj GEMV with the split partials requested in front of the weight stream (k_gemv.hip MF) vs behind it
python tools/ab_knobs.py --preset 0.6b --batch 1 --rounds 9 base Q3A_GEMV_MERGE_FIRST=2 > gpurun_out/r6_ab_merge_first.txt 2>&1
python tools/ab_knobs.py --preset 1.7b --batch 1 --rounds 7 base Q3A_GEMV_MERGE_FIRST=2 >> gpurun_out/r6_ab_merge_first.txt 2>&1
python tools/ab_knobs.py --preset 0.6b --batch 2 --rounds 7 base Q3A_GEMV_MERGE_FIRST=2 >> gpurun_out/r6_ab_merge_first.txt 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r6_ab_merge_first.txt"):
    if l.startswith("{"):
        j = json.loads(l); print(f'{j["setting"]:40s} {j["decode_us_per_step"]:8.2f} us/step  {j["ms_per_batch"]:8.3f} ms  ids equal {j["ids_equal_to_first_setting"]}')
PY
bash tools/trace_env.sh gpurun_out/r6_merge_first_traces base Q3A_GEMV_MERGE_FIRST=2
timeout 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_longform.py -x -q -k "config1 or 1p7b_one_clip or config0 or long" 2>&1 | tail -3

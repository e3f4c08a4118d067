#!/bin/bash
# round 6, GPU call 9: the fused one-sequence launch with the hand-off through the XCD-local L2 (2) vs through memory (1) vs two launches (0)
python tools/ab_knobs.py --preset 0.6b --batch 1 --rounds 7 Q3A_FUSE_QKV_DATTN=0 Q3A_FUSE_QKV_DATTN=1 Q3A_FUSE_QKV_DATTN=5 Q3A_FUSE_QKV_DATTN=7 Q3A_FUSE_QKV_DATTN=9 Q3A_FUSE_QKV_DATTN=11 > gpurun_out/r6_ab_fused_qkv_dattn_xcd.txt 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r6_ab_fused_qkv_dattn_xcd.txt"):
    if l.startswith("{"):
        j = json.loads(l); print(f'{j["setting"]:40s} {j["decode_us_per_step"]:8.2f} us/step  {j["ms_per_batch"]:8.3f} ms  ids equal {j["ids_equal_to_first_setting"]} crc {j["ids_crc32"]}')
    elif "rror" in l: print(l.strip()[:300])
PY

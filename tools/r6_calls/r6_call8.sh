#!/bin/bash
# round 6, GPU call 8: qkv projection + attention of the one-sequence decode step as ONE dispatch-ordered launch vs two launches
python tools/ab_knobs.py --preset 0.6b --batch 1 --rounds 9 base Q3A_FUSE_QKV_DATTN=0 > gpurun_out/r6_ab_fused_qkv_dattn.txt 2>&1
python tools/ab_knobs.py --preset 1.7b --batch 1 --rounds 7 base Q3A_FUSE_QKV_DATTN=0 >> gpurun_out/r6_ab_fused_qkv_dattn.txt 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r6_ab_fused_qkv_dattn.txt"):
    if l.startswith("{"):
        j = json.loads(l); print(f'{j["setting"]:40s} {j["decode_us_per_step"]:8.2f} us/step  {j["ms_per_batch"]:8.3f} ms  ids equal {j["ids_equal_to_first_setting"]} crc {j["ids_crc32"]}')
    elif "rror" in l: print(l.strip()[:300])
PY
tail -5 gpurun_out/r6_ab_fused_qkv_dattn.txt | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_eos.py -x -q -k "config1 or config0 or 1p7b_one_clip or graph or eos or key_splits" 2>&1 | tail -5

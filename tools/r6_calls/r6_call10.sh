#!/bin/bash
# round 6, GPU call 10: batched decode step, qkv projection + attention as ONE dispatch-ordered launch (k_fused.hip) vs two launches
python tools/ab_knobs.py --preset 0.6b --batch 32 --rounds 5 Q3A_FUSE_QKV_DATTN_BATCHED=0 Q3A_FUSE_QKV_DATTN_BATCHED=1 Q3A_FUSE_QKV_DATTN_BATCHED=2 Q3A_FUSE_QKV_DATTN_BATCHED=3 > gpurun_out/r6_ab_fused_batched.txt 2>&1

python - <<'PY'
import json
for l in open("gpurun_out/r6_ab_fused_batched.txt"):
    if l.startswith("{"):
        j = json.loads(l); print(f'{j["setting"]:40s} {j["decode_us_per_step"]:8.2f} us/step  {j["ms_per_batch"]:8.3f} ms  {j["audio_s_per_s"]} audio-s/s  ids equal {j["ids_equal_to_first_setting"]} differing {j["utterances_differing"]}')
    elif "rror" in l: print(l.strip()[:300])
PY
tail -3 gpurun_out/r6_ab_fused_batched.txt | cut -c1-400

#!/bin/bash
# round 6, GPU call 11: the attention's first key tile pulled into the XCD's L2 by the qkv projection launch in front of it (batched step)
python tools/ab_knobs.py --preset 0.6b --batch 32 --rounds 5 base Q3A_SKINNY_KV_PREFETCH=128 Q3A_SKINNY_KV_PREFETCH=64 > gpurun_out/r6_ab_kv_prefetch.txt 2>&1
python tools/ab_knobs.py --preset 1.7b --batch 16 --rounds 3 base Q3A_SKINNY_KV_PREFETCH=128 >> gpurun_out/r6_ab_kv_prefetch.txt 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r6_ab_kv_prefetch.txt"):
    if l.startswith("{"):
        j = json.loads(l); print(f'{j["setting"]:40s} {j["decode_us_per_step"]:8.2f} us/step  {j["ms_per_batch"]:8.3f} ms  {j["audio_s_per_s"]} audio-s/s  ids equal {j["ids_equal_to_first_setting"]} differing {j["utterances_differing"]}')
    elif "rror" in l: print(l.strip()[:300])
PY
TRACE_ARGS="--preset 0.6b --batch 32 --seconds 30 --new-tokens 100 --steps 2 --warmup 1" bash tools/trace_env.sh gpurun_out/r6_kv_prefetch_traces base Q3A_SKINNY_KV_PREFETCH=128

#!/bin/bash
# round 6, GPU call 20: the new bit-identity test of the gemm256 walk; 32 sequences as one group vs 2 x 16 / 4 x 8 parallel chains at HEAD
# (round 3 measured "no gain / lost" with the kernels of that round)
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "persistent_walk" > gpurun_out/r6_gputest_walk.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6_gputest_walk.log; tail -4 gpurun_out/r6_gputest_walk.log
python tools/ab_knobs.py --preset 0.6b --batch 32 --rounds 3 base decode_group_size=16 decode_group_size=8 > gpurun_out/r6_ab_group_size.txt 2>&1
python tools/ab_knobs.py --preset 1.7b --batch 16 --rounds 3 base decode_group_size=8 >> gpurun_out/r6_ab_group_size.txt 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r6_ab_group_size.txt"):
    if l.startswith("{"):
        j = json.loads(l); print(f'{j["setting"]:28s} {j["decode_us_per_step"]:8.2f} us/step  {j["ms_per_batch"]:8.3f} ms  {j["audio_s_per_s"]} audio-s/s  ids equal {j["ids_equal_to_first_setting"]} differing {j["utterances_differing"]}')
    elif "rror" in l: print(l.strip()[:300])
PY

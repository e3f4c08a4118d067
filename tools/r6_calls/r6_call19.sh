#!/bin/bash
# round 6, GPU call 19: several decode steps per hipGraph in the fixed-length mode (the seam between two graph launches costs ~8 us)
python tools/ab_knobs.py --preset 0.6b --batch 1 --rounds 7 decode_steps_per_graph=1 decode_steps_per_graph=2 decode_steps_per_graph=4 decode_steps_per_graph=8 decode_steps_per_graph=16 decode_steps_per_graph=32 > gpurun_out/r6_ab_steps_per_graph.txt 2>&1
python tools/ab_knobs.py --preset 0.6b --batch 32 --rounds 3 decode_steps_per_graph=1 decode_steps_per_graph=8 >> gpurun_out/r6_ab_steps_per_graph.txt 2>&1
python tools/ab_knobs.py --preset 1.7b --batch 16 --rounds 3 decode_steps_per_graph=1 decode_steps_per_graph=8 >> gpurun_out/r6_ab_steps_per_graph.txt 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r6_ab_steps_per_graph.txt"):
    if l.startswith("{"):
        j = json.loads(l); print(f'{j["setting"]:28s} {j["decode_us_per_step"]:8.2f} us/step  {j["ms_per_batch"]:8.3f} ms  {j["audio_s_per_s"]} audio-s/s  ids equal {j["ids_equal_to_first_setting"]} crc {j["ids_crc32"]}')
    elif "rror" in l: print(l.strip()[:300])
PY
timeout 900 python -m pytest tests -m gpu -x -q -k "eos or graph or config1 or step or cli or soak" > gpurun_out/r6_gputest_steps.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6_gputest_steps.log; tail -4 gpurun_out/r6_gputest_steps.log

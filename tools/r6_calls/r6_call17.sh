#!/bin/bash
# round 6, GPU call 17: persistent tile walk in gemm256 (next tile's first K tile arriving under the epilogue): exactness on shapes with
# more tiles than CUs, timing per shape with the knob off / on, then the 32-clip workloads with the knob off / on and the GPU suite
timeout 900 python tools/gemm256_probe.py quick > gpurun_out/r6_gemm256_persist_probe.txt 2>&1; echo "probe rc=$?" >> gpurun_out/r6_gemm256_persist_probe.txt
tail -32 gpurun_out/r6_gemm256_persist_probe.txt
timeout 600 python tools/ab_knobs.py --preset 0.6b --batch 32 --rounds 5 gemm256_persist=0 gemm256_persist=1 > gpurun_out/r6_ab_gemm256_persist.txt 2>&1
timeout 600 python tools/ab_knobs.py --preset 1.7b --batch 16 --rounds 3 gemm256_persist=0 gemm256_persist=1 >> gpurun_out/r6_ab_gemm256_persist.txt 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r6_ab_gemm256_persist.txt"):
    if l.startswith("{"):
        j = json.loads(l); print(f'{j["setting"]:24s} {j["ms_per_batch"]:8.3f} ms  enc {j["encoder_ms"]:7.3f} prefill {j["prefill_ms"]:7.3f} decode {j["decode_ms"]:8.3f}  {j["audio_s_per_s"]} audio-s/s  ids equal {j["ids_equal_to_first_setting"]} differing {j["utterances_differing"]}')
    elif "rror" in l: print(l.strip()[:300])
PY
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6_gputest_persist.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6_gputest_persist.log; tail -5 gpurun_out/r6_gputest_persist.log

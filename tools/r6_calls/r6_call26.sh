#!/bin/bash
# round 6, GPU call 26: gemm256 tile order by rule (groups of 8 tile rows where the matrix is >= 8 tiles wide, else N fastest) vs N fastest
# everywhere vs groups of 8 everywhere, interleaved, 32-clip workloads
timeout 900 python tools/ab_knobs.py --preset 0.6b --batch 32 --rounds 7 gemm256_group_m=1 gemm256_group_m=0 gemm256_group_m=8 > gpurun_out/r6_ab_gemm256_group_m_rule.txt 2>&1
timeout 600 python tools/ab_knobs.py --preset 1.7b --batch 16 --rounds 5 gemm256_group_m=1 gemm256_group_m=0 >> gpurun_out/r6_ab_gemm256_group_m_rule.txt 2>&1
timeout 600 python tools/ab_knobs.py --preset 1.7b --batch 32 --rounds 3 gemm256_group_m=1 gemm256_group_m=0 >> gpurun_out/r6_ab_gemm256_group_m_rule.txt 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r6_ab_gemm256_group_m_rule.txt"):
    if l.startswith("{"):
        j = json.loads(l); print(f'{j["setting"]:24s} {j["ms_per_batch"]:8.3f} ms  enc {j["encoder_ms"]:7.3f} prefill {j["prefill_ms"]:7.3f} decode {j["decode_ms"]:8.3f}  {j["audio_s_per_s"]} audio-s/s  ids equal {j["ids_equal_to_first_setting"]} differing {j["utterances_differing"]}  all {j["all_ms"]}')
    elif "rror" in l: print(l.strip()[:300])
PY

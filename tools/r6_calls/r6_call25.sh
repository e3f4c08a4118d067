#!/bin/bash
# round 6, GPU call 25: gemm256 tile order in groups of 8 tile rows (M fastest inside a group) vs N fastest: exactness, per-shape timing,
# 32-clip workloads, memory-side traffic per shape
timeout 900 python tools/gemm256_probe.py quick > gpurun_out/r6_gemm256_group_m_probe.txt 2>&1; echo "probe rc=$?" >> gpurun_out/r6_gemm256_group_m_probe.txt
tail -17 gpurun_out/r6_gemm256_group_m_probe.txt | cut -c1-260
timeout 600 python tools/ab_knobs.py --preset 0.6b --batch 32 --rounds 5 gemm256_group_m=1 gemm256_group_m=8 gemm256_group_m=4 gemm256_group_m=16 > gpurun_out/r6_ab_gemm256_group_m.txt 2>&1
timeout 600 python tools/ab_knobs.py --preset 1.7b --batch 16 --rounds 3 gemm256_group_m=1 gemm256_group_m=8 >> gpurun_out/r6_ab_gemm256_group_m.txt 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r6_ab_gemm256_group_m.txt"):
    if l.startswith("{"):
        j = json.loads(l); print(f'{j["setting"]:24s} {j["ms_per_batch"]:8.3f} ms  enc {j["encoder_ms"]:7.3f} prefill {j["prefill_ms"]:7.3f} decode {j["decode_ms"]:8.3f}  {j["audio_s_per_s"]} audio-s/s  ids equal {j["ids_equal_to_first_setting"]} differing {j["utterances_differing"]}')
    elif "rror" in l: print(l.strip()[:300])
PY
R=$PWD; out=gpurun_out/r6_gemm_traffic; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $R/$out/${c}_g -o t -- env PMC_BATCH=32 python $R/tools/pmc_target_enc.py > $R/$out/${c}_g.log 2>&1
done
cd $R
python tools/gemm_traffic.py --fetch $(find $out/FETCH_SIZE_g -name "*_results.db" | head -1) --write $(find $out/WRITE_SIZE_g -name "*_results.db" | head -1) --batch 32 > gpurun_out/r6_gemm256_hbm_traffic_group_m8.txt 2>&1
rm -rf $out/FETCH_SIZE_* $out/WRITE_SIZE_*
cut -c1-130 gpurun_out/r6_gemm256_hbm_traffic_group_m8.txt

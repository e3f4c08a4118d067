#!/bin/bash
# round 6, GPU call 15: rows per wave of the one-sequence GEMVs re-measured behind this round's changes (merge in front of the o_proj's
# weights; x through LDS in the norm-fused projections): one process per setting (read once), two interleaved passes
mkdir -p gpurun_out/r6_pr
for pass in 1 2; do
 for s in base Q3A_GEMV_PR_SMALL=2 Q3A_GEMV_PR_MID=4 "Q3A_GEMV_PR_SMALL=2 Q3A_GEMV_PR_MID=4"; do
  tag=$(echo "$s" | tr ' =' '__')
  if [ "$s" = base ]; then e=""; else e="$s"; fi
  env $e timeout 300 python tools/ab_knobs.py --preset 0.6b --batch 1 --rounds 5 base > gpurun_out/r6_pr/$tag.$pass.txt 2>&1
  python - "$s" "$pass" gpurun_out/r6_pr/$tag.$pass.txt <<'PY'
import json, sys
try:
    j = [json.loads(l) for l in open(sys.argv[3]) if l.startswith("{")][-1]
    print(f'{sys.argv[1]:44s} pass {sys.argv[2]}  {j["decode_us_per_step"]:8.2f} us/step  {j["ms_per_batch"]:8.3f} ms  crc {j["ids_crc32"]}', flush=True)
except Exception as ex:
    print(sys.argv[1], "pass", sys.argv[2], "FAILED", ex, flush=True)
PY
 done
done 2>&1 | tee gpurun_out/r6_gemv_rows_per_wave.txt
# the stamped timelines of both decode layers with the probe rebuilt at HEAD (x through LDS in the norm-fused GEMVs)
timeout 120 tools/bin/phase_probe layer,one > gpurun_out/r6_phase_probe_decode_layers.txt 2>&1; tail -14 gpurun_out/r6_phase_probe_decode_layers.txt | cut -c1-200

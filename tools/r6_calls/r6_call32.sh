#!/bin/bash
# round 6, GPU call 32: causal flash attention -- query tiles of a head handed out longest first vs ascending (-DQ3A_FATTN_HEAVY_FIRST=0)
OLD="Q3A_LIB=$PWD/qwen3_asr_rs_amd/lib/libq3asr_hip_asc.so"
run() { # tag preset batch rounds seconds
  if [ "$1" = new ]; then L=""; else L="$OLD"; fi
  env $L timeout 400 python tools/ab_knobs.py --preset $2 --batch $3 --rounds $4 --seconds ${5:-30} base 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$1 $2 x $3 (${5:-30} s):', j['ms_per_batch'], 'ms  enc', j['encoder_ms'], 'prefill', j['prefill_ms'], 'decode', j['decode_ms'], j['audio_s_per_s'], 'audio-s/s crc', j['ids_crc32'])"
}
for pass in 1 2 3; do
  for tag in old new; do run $tag 0.6b 32 3; done
done 2>&1 | tee gpurun_out/r6_ab_fattn_heavy_first.txt
for tag in old new; do run $tag 1.7b 16 3; done 2>&1 | tee -a gpurun_out/r6_ab_fattn_heavy_first.txt
for tag in old new; do run $tag 0.6b 8 3 120; done 2>&1 | tee -a gpurun_out/r6_ab_fattn_heavy_first.txt
for tag in old new; do run $tag 0.6b 1 5; done 2>&1 | tee -a gpurun_out/r6_ab_fattn_heavy_first.txt
TRACE_ARGS="--preset 0.6b --batch 32 --seconds 30 --new-tokens 4 --steps 2 --warmup 1" bash tools/trace_env.sh gpurun_out/r6_fattn_heavy_traces "$OLD" base > /dev/null 2>&1
for f in gpurun_out/r6_fattn_heavy_traces/*.txt; do echo "== $f"; grep -h "fattn\|^#" $f; done | tee -a gpurun_out/r6_ab_fattn_heavy_first.txt

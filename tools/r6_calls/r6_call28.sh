#!/bin/bash
# round 6, GPU call 28: flash attention -- the query tiles of one (segment, kv head) on ONE XCD (bijective block remap) vs the plain
# blockIdx mapping (library built with -DQ3A_FATTN_XCD_REMAP=0).  One process per library, interleaved, two passes; in-situ kernel traces.
OLD="Q3A_LIB=$PWD/qwen3_asr_rs_amd/lib/libq3asr_hip_noremap.so"
run() { # tag preset batch rounds
  if [ "$1" = new ]; then L=""; else L="$OLD"; fi
  env $L timeout 400 python tools/ab_knobs.py --preset $2 --batch $3 --rounds $4 base 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$1 $2 x $3:', j['ms_per_batch'], 'ms  enc', j['encoder_ms'], 'prefill', j['prefill_ms'], 'decode', j['decode_ms'], j['audio_s_per_s'], 'audio-s/s crc', j['ids_crc32'])"
}
for pass in 1 2; do
  for tag in old new; do run $tag 0.6b 32 3; done
  for tag in old new; do run $tag 1.7b 16 3; done
  for tag in old new; do run $tag 0.6b 1 5; done
done 2>&1 | tee gpurun_out/r6_ab_fattn_xcd_remap.txt
for tag in old new; do run $tag 1.7b 32 3; done 2>&1 | tee -a gpurun_out/r6_ab_fattn_xcd_remap.txt
TRACE_ARGS="--preset 0.6b --batch 32 --seconds 30 --new-tokens 4 --steps 2 --warmup 1" bash tools/trace_env.sh gpurun_out/r6_fattn_remap_traces "$OLD" base > /dev/null 2>&1
grep -h "fattn\|^#" gpurun_out/r6_fattn_remap_traces/*.txt | tee -a gpurun_out/r6_ab_fattn_xcd_remap.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "attention or config2 or stage_parity or config0 or ragged or window" > gpurun_out/r6_gputest_remap.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6_gputest_remap.log; tail -3 gpurun_out/r6_gputest_remap.log

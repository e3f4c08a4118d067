#!/bin/bash
# round 6, GPU call 31: gemm256 epilogue outputs written through (sc1 stores) vs plain stores (-DQ3A_SC1_STORES=0): the kernel's end is an
# agent-scope release that writes the L2's dirty lines back.  One process per library, interleaved, two passes; per-shape probe timing.
OLD="Q3A_LIB=$PWD/qwen3_asr_rs_amd/lib/libq3asr_hip_plainst.so"
run() { # tag preset batch rounds
  if [ "$1" = new ]; then L=""; else L="$OLD"; fi
  env $L timeout 400 python tools/ab_knobs.py --preset $2 --batch $3 --rounds $4 base 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$1 $2 x $3:', j['ms_per_batch'], 'ms  enc', j['encoder_ms'], 'prefill', j['prefill_ms'], 'decode', j['decode_ms'], j['audio_s_per_s'], 'audio-s/s crc', j['ids_crc32'])"
}
for pass in 1 2; do
  for tag in old new; do run $tag 0.6b 32 3; done
  for tag in old new; do run $tag 1.7b 16 3; done
done 2>&1 | tee gpurun_out/r6_ab_sc1_stores.txt
for tag in old new; do run $tag 1.7b 32 3; done 2>&1 | tee -a gpurun_out/r6_ab_sc1_stores.txt
for tag in old new; do
  if [ "$tag" = new ]; then L=""; else L="$OLD"; fi
  echo "== probe timing, $tag library ==" | tee -a gpurun_out/r6_ab_sc1_stores.txt
  env $L timeout 600 python tools/gemm256_probe.py quick 2>&1 | grep "M=" | cut -c1-230 | tee -a gpurun_out/r6_ab_sc1_stores.txt
done

#!/bin/bash
# round 6, GPU call 22: skinny GEMM -- every weight fragment of a pass read from LDS in ONE batch in front of the MFMAs (the ISA had
# ds_read -> lgkmcnt(0) -> mfma per k-step: a chain of LDS round trips) vs the same library built with -DQ3A_SK_WBATCH=0.  One process
# per library, interleaved, two passes; then the skinny / batched-decode GPU tests on the product library.
run() { # lib-tag preset batch rounds
  if [ "$1" = new ]; then L=""; else L="Q3A_LIB=$PWD/qwen3_asr_rs_amd/lib/libq3asr_hip_nowb.so"; fi
  env $L timeout 400 python tools/ab_knobs.py --preset $2 --batch $3 --rounds $4 base 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$1 $2 x $3:', j['decode_us_per_step'], 'us/step', j['ms_per_batch'], 'ms', j['audio_s_per_s'], 'audio-s/s crc', j['ids_crc32'])"
}
for pass in 1 2; do
  for tag in old new; do run $tag 0.6b 32 3; done
  for tag in old new; do run $tag 1.7b 16 3; done
done 2>&1 | tee gpurun_out/r6_ab_skinny_wbatch.txt
for tag in old new; do run $tag 1.7b 32 3; done 2>&1 | tee -a gpurun_out/r6_ab_skinny_wbatch.txt
for tag in old new; do run $tag 0.6b 8 3; done 2>&1 | tee -a gpurun_out/r6_ab_skinny_wbatch.txt
timeout 1200 python -m pytest tests -m gpu -x -q -k "skinny or quarter or batch or config2 or config3 or stage_parity or eos" > gpurun_out/r6_gputest_wbatch.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6_gputest_wbatch.log; tail -4 gpurun_out/r6_gputest_wbatch.log

#!/bin/bash
# Round 5, GPU call 16: trim of the prologue's second tile (needs `pos` before it is requested) against in-loop trim only, same box, interleaved.
O=gpurun_out/r5c16; mkdir -p $O
export PYTHONUNBUFFERED=1
LOOP=$PWD/qwen3_asr_rs_amd/lib/libq3asr_hip_trimloop.so
for v in all loop all loop all loop; do
  if [ $v = loop ]; then export Q3A_LIB=$LOOP; else unset Q3A_LIB; fi
  echo "== 0.6b x 32 x 30s trim=$v" | tee -a $O/ab_dattn_trim3.txt
  timeout 200 python tools/ab_knobs.py --preset 0.6b --batch 32 --rounds 3 base 2>&1 | grep setting | cut -c1-330 | tee -a $O/ab_dattn_trim3.txt
done

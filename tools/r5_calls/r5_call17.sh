#!/bin/bash
# Round 5, GPU call 17: the gated prologue trim: 32 x 30 s (hint off) and 32 x 7 s (hint on), then the whole GPU suite at HEAD.
O=gpurun_out/r5c17; mkdir -p $O
export PYTHONUNBUFFERED=1
for sec in 30 7; do
  echo "== 0.6b x 32 x ${sec}s" | tee -a $O/ab_dattn_trim4.txt
  timeout 200 python tools/ab_knobs.py --preset 0.6b --batch 32 --seconds $sec --rounds 3 base 2>&1 | grep setting | cut -c1-330 | tee -a $O/ab_dattn_trim4.txt
done
timeout 1300 python -m pytest tests -m gpu -q -s 2>&1 | tail -100 > $O/r5_gputest_final.log; tail -3 $O/r5_gputest_final.log | cut -c1-300

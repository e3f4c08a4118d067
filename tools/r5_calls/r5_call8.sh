#!/bin/bash
# Round 5, GPU call 8: batched decode attention on half the CUs (16 sequences): more tiles in flight per wave (existing env knobs), A/B.
O=gpurun_out/r5c8; mkdir -p $O
export PYTHONUNBUFFERED=1
for v in base ring4 tile64 base2; do
  case $v in ring4) E="Q3A_DATTN_RING4=1";; tile64) E="Q3A_DATTN_TILE64=1";; *) E="Q3A_NOP=1";; esac
  echo "== $v" | tee -a $O/ab_dattn_ring_1p7b_b16.txt
  env $E timeout 200 python tools/ab_knobs.py --preset 1.7b --batch 16 --rounds 3 base 2>&1 | grep setting | cut -c1-330 | tee -a $O/ab_dattn_ring_1p7b_b16.txt
done
for v in base ring4; do
  case $v in ring4) E="Q3A_DATTN_RING4=1";; *) E="Q3A_NOP=1";; esac
  echo "== 0.6b x 16 $v" | tee -a $O/ab_dattn_ring_1p7b_b16.txt
  env $E timeout 200 python tools/ab_knobs.py --preset 0.6b --batch 16 --rounds 3 base 2>&1 | grep setting | cut -c1-330 | tee -a $O/ab_dattn_ring_1p7b_b16.txt
done

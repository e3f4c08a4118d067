#!/bin/bash
# Round 5, GPU call 19: batched decode attention with an INNER form for the tiles in front of the one that holds `pos` (no patching, no masks):
# A/B against a -DQ3A_DATTN_EDGE=0 build of the same sources (ids must be equal: the arithmetic is the same), then the whole GPU suite.
O=gpurun_out/r5c19; mkdir -p $O
export PYTHONUNBUFFERED=1
OLD=$PWD/qwen3_asr_rs_amd/lib/libq3asr_hip_noedge.so
for round in 1 2; do
  for v in edge noedge; do
    if [ $v = noedge ]; then export Q3A_LIB=$OLD; else unset Q3A_LIB; fi
    echo "== 0.6b x 32 $v (round $round)" | tee -a $O/ab_dattn_edge.txt
    timeout 120 python tools/ab_knobs.py --preset 0.6b --batch 32 --rounds 3 base 2>&1 | grep setting | cut -c1-420 | tee -a $O/ab_dattn_edge.txt
  done
done
unset Q3A_LIB
timeout 450 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $O/gputest.log; tail -4 $O/gputest.log

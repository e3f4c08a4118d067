#!/bin/bash
# Round 5, GPU call 13: batched decode attention with in-loop tile requests trimmed to the live rows: A/B against a -DQ3A_DATTN_TRIM=0 build, tests.
O=gpurun_out/r5c13; mkdir -p $O
export PYTHONUNBUFFERED=1
NOTRIM=$PWD/qwen3_asr_rs_amd/lib/libq3asr_hip_notrim.so
for round in 1 2; do
  for v in trim notrim; do
    if [ $v = notrim ]; then export Q3A_LIB=$NOTRIM; else unset Q3A_LIB; fi
    echo "== 0.6b x 32 $v (round $round)" | tee -a $O/ab_dattn_trim.txt
    timeout 200 python tools/ab_knobs.py --preset 0.6b --batch 32 --rounds 3 base 2>&1 | grep setting | cut -c1-330 | tee -a $O/ab_dattn_trim.txt
  done
done
for v in trim notrim; do
  if [ $v = notrim ]; then export Q3A_LIB=$NOTRIM; else unset Q3A_LIB; fi
  echo "== 1.7b x 16 $v" | tee -a $O/ab_dattn_trim.txt
  timeout 200 python tools/ab_knobs.py --preset 1.7b --batch 16 --rounds 3 base 2>&1 | grep setting | cut -c1-330 | tee -a $O/ab_dattn_trim.txt
done
unset Q3A_LIB
timeout 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_longform.py tests/test_gpu_eos.py -q -s -m gpu -k "batched_decode or config2 or config3 or graph_replay or batch_above or eos or pair_split" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -5 $O/tests.log | cut -c1-300

#!/bin/bash
# Round 5, GPU call 15: prologue tile 1 trimmed as well: tests + A/B at 32 x 30 s (must stay at the in-loop-trim numbers) and 32 x 7 s clips.
O=gpurun_out/r5c15; mkdir -p $O
export PYTHONUNBUFFERED=1
NOTRIM=$PWD/qwen3_asr_rs_amd/lib/libq3asr_hip_notrim.so
for sec in 30 7; do
  for v in trim notrim trim notrim; do
    if [ $v = notrim ]; then export Q3A_LIB=$NOTRIM; else unset Q3A_LIB; fi
    echo "== 0.6b x 32 x ${sec}s $v" | tee -a $O/ab_dattn_trim2.txt
    timeout 200 python tools/ab_knobs.py --preset 0.6b --batch 32 --seconds $sec --rounds 3 base 2>&1 | grep setting | cut -c1-330 | tee -a $O/ab_dattn_trim2.txt
  done
done
unset Q3A_LIB
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_longform.py tests/test_gpu_eos.py -q -s -m gpu -k "batched_decode or config2 or config3 or graph_replay or batch_above or eos or pair_split or stage_parity_tiny or long_audio" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -4 $O/tests.log | cut -c1-300

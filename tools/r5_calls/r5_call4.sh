#!/bin/bash
# Round 5, GPU call 4: pair-split batched decode attention (16 sequences): tests + A/B at 1.7B x 16 and 0.6B x 16.
O=gpurun_out/r5c4; mkdir -p $O; R=$PWD
export PYTHONUNBUFFERED=1
step() { echo "=== $1 ($(date +%T))" | tee -a $O/steps.log; }
step tests_pair
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_longform.py -q -s -m gpu -k "pair_split or batched_decode_attention" > $O/tests_pair.log 2>&1; echo "rc=$?" >> $O/tests_pair.log; tail -8 $O/tests_pair.log | cut -c1-300
step longform_with_pair
Q3A_DATTN_PAIR_SPLIT=1 timeout 300 python -m pytest tests/test_gpu_longform.py -q -s -m gpu -k "batched_decode_attention_long" > $O/longform_pair.log 2>&1; echo "rc=$?" >> $O/longform_pair.log; tail -6 $O/longform_pair.log | cut -c1-300
step ab_1p7b_b16
timeout 300 python tools/ab_knobs.py --preset 1.7b --batch 16 --rounds 3 base dattn_pair_split=1 > $O/ab_1p7b_b16_pair.txt 2>&1; cat $O/ab_1p7b_b16_pair.txt | cut -c1-420
step ab_0p6b_b16
timeout 200 python tools/ab_knobs.py --preset 0.6b --batch 16 --rounds 3 base dattn_pair_split=1 > $O/ab_0p6b_b16_pair.txt 2>&1; cat $O/ab_0p6b_b16_pair.txt | cut -c1-420
step config3_pair
Q3A_DATTN_PAIR_SPLIT=1 timeout 300 python -m pytest tests/test_gpu_configs.py -q -s -m gpu -k "config3" > $O/config3_pair.log 2>&1; echo "rc=$?" >> $O/config3_pair.log; tail -5 $O/config3_pair.log | cut -c1-300
step done

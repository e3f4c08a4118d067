#!/bin/bash
# Round 5, GPU call 3: half-pair gate/up form (bit-identity + A/B at 1.7B x 16 and x 32), then the whole GPU suite on this library.
O=gpurun_out/r5c3; mkdir -p $O; R=$PWD
export PYTHONUNBUFFERED=1
step() { echo "=== $1 ($(date +%T))" | tee -a $O/steps.log; }
step tests_glu
timeout 500 python -m pytest tests/test_gpu_parity.py -q -s -m gpu -k "gate_up_skinny or quarter_workgroup or batched_decode_attention_and" > $O/tests_glu.log 2>&1; echo "rc=$?" >> $O/tests_glu.log; tail -6 $O/tests_glu.log | cut -c1-300
step ab_1p7b_b16
timeout 300 python tools/ab_knobs.py --preset 1.7b --batch 16 --rounds 3 base skinny_glu_hp3=0 skinny_glu_hp3=0,skinny_glu_2pass=0 > $O/ab_1p7b_b16.txt 2>&1; cat $O/ab_1p7b_b16.txt | cut -c1-420
step ab_1p7b_b32
timeout 300 python tools/ab_knobs.py --preset 1.7b --batch 32 --rounds 2 base skinny_glu_hp3=0 skinny_glu_hp3=0,skinny_glu_2pass=0 > $O/ab_1p7b_b32.txt 2>&1; cat $O/ab_1p7b_b32.txt | cut -c1-420
step full_gpu_suite
timeout 1100 python -m pytest tests -q -m gpu -s > $O/gputest.log 2>&1; echo "rc=$?" >> $O/gputest.log; tail -4 $O/gputest.log | cut -c1-300
step done

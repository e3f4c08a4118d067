#!/bin/bash
# Round 5, GPU call 7: the new long-context test (separate split-merge launch) and the extended soak (load engines bit-compared).
O=gpurun_out/r5c7; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_longform.py -q -s -m gpu -k "two_five_minute" > $O/long2.log 2>&1; echo "rc=$?" >> $O/long2.log; tail -8 $O/long2.log | cut -c1-300
timeout 400 python -m pytest tests/test_gpu_soak.py -q -s -m gpu > $O/soak.log 2>&1; echo "rc=$?" >> $O/soak.log; tail -6 $O/soak.log | cut -c1-400

#!/bin/bash
# round 6, GPU call 16: HEAD after the vectorised split statistics of the o_proj merge: one-clip step time (two passes) + GPU suite
for p in 1 2; do python tools/ab_knobs.py --preset 0.6b --batch 1 --rounds 5 base 2>/dev/null | tail -1 | cut -c1-330; done | tee gpurun_out/r6_head_one_clip.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6_gputest_head.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6_gputest_head.log; tail -3 gpurun_out/r6_gputest_head.log

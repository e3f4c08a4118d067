#!/bin/bash
# Round 5, GPU call 9: the whole GPU suite + smoke at HEAD.
O=gpurun_out/r5c9; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1300 python -m pytest tests -m gpu -q -s 2>&1 | tail -100 > $O/r5_gputest_head.log; tail -4 $O/r5_gputest_head.log | cut -c1-300
timeout 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -3 $O/smoke.log

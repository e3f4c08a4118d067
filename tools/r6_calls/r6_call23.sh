#!/bin/bash
# round 6, GPU call 23: the walk on odd grids (2 / 5 / 13 / 250 workgroups) + the bit-identity test again
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "walk" > gpurun_out/r6_gputest_walk_odd.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6_gputest_walk_odd.log; tail -6 gpurun_out/r6_gputest_walk_odd.log

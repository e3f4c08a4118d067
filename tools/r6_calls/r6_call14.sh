#!/bin/bash
# round 6, GPU call 14: the LDS-shared x form as the default of the norm-fused one-sequence GEMVs at K <= 1024: parity + suite
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6_gputest_xl.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6_gputest_xl.log
tail -4 gpurun_out/r6_gputest_xl.log
python tools/ab_knobs.py --preset 0.6b --batch 1 --rounds 5 base > gpurun_out/r6_xl_default.txt 2>&1; tail -1 gpurun_out/r6_xl_default.txt | cut -c1-300
python tools/ab_knobs.py --preset 0.6b --batch 2 --rounds 3 base >> gpurun_out/r6_xl_default.txt 2>&1; tail -1 gpurun_out/r6_xl_default.txt | cut -c1-300

#!/bin/bash
# round 6, GPU call 2: L2 warm-up duty of the one-sequence decode attention launch (k_dattn.hip l2_warm), A/B on one box, one
# engine, settings interleaved: mask bits 1 = o_proj matrix, 2 = down matrix, 4 = next layer's K / V rows
mkdir -p gpurun_out
python tools/ab_knobs.py --preset 0.6b --batch 1 --rounds 9 Q3A_DATTN_WARM=0 Q3A_DATTN_WARM=7 Q3A_DATTN_WARM=1 Q3A_DATTN_WARM=4 Q3A_DATTN_WARM=5 Q3A_DATTN_WARM=3 > gpurun_out/r6_ab_dattn_warm.txt 2>&1
cat gpurun_out/r6_ab_dattn_warm.txt
python tools/ab_knobs.py --preset 1.7b --batch 1 --rounds 7 Q3A_DATTN_WARM=0 Q3A_DATTN_WARM=7 Q3A_DATTN_WARM=5 Q3A_DATTN_WARM=4 > gpurun_out/r6_ab_dattn_warm_1p7b.txt 2>&1
cat gpurun_out/r6_ab_dattn_warm_1p7b.txt
timeout 300 python -m pytest tests/test_gpu_configs.py -x -q -k "config1 or 1p7b_one_clip" 2>&1 | tail -3

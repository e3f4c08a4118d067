#!/bin/bash
# round 6, GPU call 3: where the L2 warm-up duty helps / hurts, per kernel (in-situ traces under four settings)
bash tools/trace_env.sh gpurun_out/r6_warm_traces Q3A_DATTN_WARM=0 Q3A_DATTN_WARM=7 Q3A_DATTN_WARM=1 Q3A_DATTN_WARM=4 Q3A_DATTN_WARM=2

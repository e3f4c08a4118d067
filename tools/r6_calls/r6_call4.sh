#!/bin/bash
# round 6, GPU call 4: the warm-up duty's contention with the attention's own requests: delay / fewer loading waves / fewer warmers
S="Q3A_DATTN_WARM=0 Q3A_DATTN_WARM=7 Q3A_DATTN_WARM=7,Q3A_DATTN_WARM_DELAY=1 Q3A_DATTN_WARM=7,Q3A_DATTN_WARM_DELAY=2 Q3A_DATTN_WARM=7,Q3A_DATTN_WARM_WAVES=2 Q3A_DATTN_WARM=7,Q3A_DATTN_WARM_WAVES=4 Q3A_DATTN_WARM=7,Q3A_DATTN_WARM_DUP=1 Q3A_DATTN_WARM=7,Q3A_DATTN_WARM_LAYERS=14 Q3A_DATTN_WARM=1,Q3A_DATTN_WARM_DELAY=2 Q3A_DATTN_WARM=1,Q3A_DATTN_WARM_DUP=1 Q3A_DATTN_WARM=1,Q3A_DATTN_WARM_LAYERS=8"
python tools/ab_knobs.py --preset 0.6b --batch 1 --rounds 7 $S > gpurun_out/r6_ab_dattn_warm_2.txt 2>&1
cut -c1-130 gpurun_out/r6_ab_dattn_warm_2.txt | sed 's/"all_ms.*audio/ audio/'
python - <<'PY'
import json
for l in open("gpurun_out/r6_ab_dattn_warm_2.txt"):
    if l.startswith("{"):
        j = json.loads(l); print(f'{j["setting"]:60s} {j["decode_us_per_step"]:8.2f} us/step  ids equal {j["ids_equal_to_first_setting"]}')
PY

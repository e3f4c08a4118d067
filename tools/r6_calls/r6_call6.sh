#!/bin/bash
# round 6, GPU call 6: HIP runtime switches that touch the kernel boundary (fence scope, kernarg placement, graph batching), one
# process per setting (read at HIP start-up), two interleaved passes; decode us per step of the one-clip workload
mkdir -p gpurun_out/r6_rt_env
for pass in 1 2; do
 for s in base HIP_FORCE_DEV_KERNARG=1 HIP_FORCE_DEV_KERNARG=0 AMD_OPT_FLUSH=0 ROC_USE_FGS_KERNARG=0 DEBUG_HIP_GRAPH_BATCH_SIZE=1 DEBUG_HIP_GRAPH_BATCH_SIZE=256 DEBUG_HIP_KERNARG_COPY_OPT=0 ROC_SYSTEM_SCOPE_SIGNAL=0 GPU_MAX_HW_QUEUES=1; do
  if [ "$s" = base ]; then e=""; else e="$s"; fi
  env $e timeout 300 python tools/ab_knobs.py --preset 0.6b --batch 1 --rounds 5 base > gpurun_out/r6_rt_env/$s.$pass.txt 2>&1
  python - "$s" "$pass" gpurun_out/r6_rt_env/$s.$pass.txt <<'PY'
import json, sys
try:
    j = [json.loads(l) for l in open(sys.argv[3]) if l.startswith("{")][-1]
    print(f'{sys.argv[1]:36s} pass {sys.argv[2]}  {j["decode_us_per_step"]:8.2f} us/step  {j["ms_per_batch"]:8.3f} ms  crc {j["ids_crc32"]}', flush=True)
except Exception as ex:
    print(sys.argv[1], "pass", sys.argv[2], "FAILED", ex, flush=True)
PY
 done
done 2>&1 | tee gpurun_out/r6_rt_env_summary.txt

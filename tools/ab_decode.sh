# A/B runs of bench.py (batch 1) across library variants / env settings: name then env assignments
mkdir -p gpurun_out/ab
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/ab/$name.log 2>&1; python - <<PY
import json
l=[x for x in open("gpurun_out/ab/$name.log") if x.startswith("{")]
if l:
    j=json.loads(l[-1]); print("$name", j["value"], j["stage_ms"], {k:v["us_per_launch_evt"] for k,v in j["decode_step_profile"].items()})
else: print("$name FAILED")
PY
}
L=$PWD/qwen3_asr_rs_amd/lib
for v in "$@"; do
  if [ "$v" = base ]; then run base A=1; else run $v Q3A_LIB=$L/libq3asr_hip_$v.so; fi
done

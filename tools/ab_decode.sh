mkdir -p gpurun_out/ab
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/ab/$name.log 2>&1; python - <<PY
import json
l=[x for x in open("gpurun_out/ab/$name.log") if x.startswith("{")]
if l:
    j=json.loads(l[-1]); print("$name", j["value"], j["stage_ms"], {k:v["us_per_launch_evt"] for k,v in j["decode_step_profile"].items()})
else: print("$name FAILED")
PY
}
run base A=1
run kernarg1 HIP_FORCE_DEV_KERNARG=1
run kernarg0 HIP_FORCE_DEV_KERNARG=0
run nt Q3A_LIB=$PWD/qwen3_asr_rs_amd/lib/libq3asr_hip_nt.so
run nt_kernarg1 Q3A_LIB=$PWD/qwen3_asr_rs_amd/lib/libq3asr_hip_nt.so HIP_FORCE_DEV_KERNARG=1
run base2 A=1

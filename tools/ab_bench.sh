#!/bin/bash
# A/B of bench.py (default workload, no child passes) across library variants on ONE box, interleaved:
#   bash tools/ab_bench.sh OUTDIR ROUNDS base <variant> [<variant> ...]
# variants are built with Q3A_BUILD_VARIANT=<variant> Q3A_BUILD_DEFINES="-D..." python -m qwen3_asr_rs_amd.build
out=$1; rounds=$2; shift 2
mkdir -p "$out"
L=$PWD/qwen3_asr_rs_amd/lib
for r in $(seq 1 "$rounds"); do
  for v in "$@"; do
    if [ "$v" = base ]; then lib=$L/libq3asr_hip.so; else lib=$L/libq3asr_hip_$v.so; fi
    Q3A_LIB=$lib python bench.py --no-cpu-baseline --no-rocprof --no-extra --steps 5 --warmup 2 $AB_ARGS > "$out/$v.$r.json" 2> "$out/$v.$r.err"
    python - "$out/$v.$r.json" "$v" "$r" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1]))
    print(sys.argv[2], "round", sys.argv[3], "value", j["value"], "ms", j["ms_per_step"], "decode_ms", j["stage_ms"]["decode_ms"], "us/step", j["roofline"]["decode_stage"]["us_per_step"], flush=True)
except Exception as e:
    print(sys.argv[2], "round", sys.argv[3], "FAILED", e, flush=True)
PY
  done
done

#!/bin/bash
# round 6, GPU call 1: pruned library -- GPU suite, measured peaks, default bench line + in-situ trace (start-of-round record)
mkdir -p gpurun_out
python -c "
from qwen3_asr_rs_amd.engine import measure_peaks
import json; print(json.dumps(measure_peaks(0, 5)))" > gpurun_out/r6_peaks_start.json 2> gpurun_out/r6_peaks_start.err
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6_gputest_start.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r6_gputest_start.log
timeout 600 python bench.py --no-extra --trace-out gpurun_out/r6_kernel_trace_b1_start.txt > gpurun_out/r6_bench_start.json 2> gpurun_out/r6_bench_start.err
tail -3 gpurun_out/r6_gputest_start.log; cat gpurun_out/r6_peaks_start.json

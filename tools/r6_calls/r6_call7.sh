#!/bin/bash
# round 6, GPU call 7: the full default bench line (every leg, per-leg roofline, measured peaks, precise mode) + GPU suite at HEAD
mkdir -p gpurun_out
( time timeout 1200 python bench.py --trace-out gpurun_out/r6_kernel_trace_b1.txt ) > gpurun_out/r6_bench_mid.json 2> gpurun_out/r6_bench_mid.err
tail -4 gpurun_out/r6_bench_mid.err
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6_gputest_mid.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6_gputest_mid.log
tail -4 gpurun_out/r6_gputest_mid.log

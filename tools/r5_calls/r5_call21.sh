#!/bin/bash
# Round 5, GPU call 21: the default bench line (one clip; extra legs at 32 clips and 1.7B x 16) + its in-situ trace at HEAD
O=gpurun_out/r5c21; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 200 python bench.py --steps 20 --warmup 5 --trace-out $O/r5_kernel_trace_b1.txt > $O/r5_bench_final.json 2> $O/bench.err
cut -c1-600 $O/r5_bench_final.json

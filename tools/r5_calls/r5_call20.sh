#!/bin/bash
# Round 5, GPU call 20: the 32-clip and 1.7B x 16 bench lines + in-situ traces again at HEAD (inner-tile form of the batched decode attention)
O=gpurun_out/r5c20; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 150 python bench.py --batch 32 --no-cpu-baseline --no-extra --trace-out $O/r5_kernel_trace_b32.txt > $O/r5_bench_b32_final.json 2> $O/err32.log
timeout 150 python bench.py --preset 1.7b --batch 16 --steps 5 --warmup 2 --no-cpu-baseline --no-extra --trace-out $O/r5_kernel_trace_1p7b_b16.txt > $O/r5_bench_1p7b_b16_final.json 2> $O/err16.log
cut -c1-400 $O/r5_bench_b32_final.json; echo; cut -c1-400 $O/r5_bench_1p7b_b16_final.json; echo; head -8 $O/r5_kernel_trace_b32.txt | cut -c1-150

#!/bin/bash
# call 18: HBM traffic (PMC) of the dominant kernel at 1.7B x 16 and 1.7B x 32 -- the final 1.7B lines were taken with --no-pmc
out=gpurun_out/r5c18; mkdir -p $out
export PYTHONUNBUFFERED=1
timeout 300 python bench.py --preset 1.7b --batch 16 --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $out/bench_1p7b_b16_pmc.json 2> $out/err16.log
timeout 300 python bench.py --preset 1.7b --batch 32 --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $out/bench_1p7b_b32_pmc.json 2> $out/err32.log
cut -c1-1500 $out/bench_1p7b_b16_pmc.json; echo; cut -c1-1500 $out/bench_1p7b_b32_pmc.json

#!/bin/bash
# Experiment matrix for the concurrent-engine difference; run on the GPU box: bash tools/soak_matrix.sh OUTDIR
out=${1:-gpurun_out/soak}; mkdir -p $out
run() { tag=$1; shift; echo "=== $tag: $*" | tee -a $out/summary.txt; env "$@" timeout 900 python tools/soak_engines.py --tag $tag $SOAK_ARGS > $out/$tag.log 2>&1; grep -E "SOAK|mismatch|run " $out/$tag.log | cut -c1-300 | tail -${TAILN:-12} | tee -a $out/summary.txt; }
SOAK_ARGS="--runs 600 --load 1" run product_ring1_twice Q3A_GEMM16_RING=1 Q3A_DEBUG_ROPE_TWICE=1
SOAK_ARGS="--runs 400 --load 2" run product_ring1_load2 Q3A_GEMM16_RING=1
SOAK_ARGS="--runs 400 --load 1" run product_ring0 Q3A_GEMM16_RING=0
echo "=== bisect (ring 1)" | tee -a $out/summary.txt
Q3A_GEMM16_RING=1 Q3A_DEBUG_LAYER_TAPS=1 timeout 900 python tools/bisect_layers.py 400 > $out/bisect_ring1.log 2>&1; tail -30 $out/bisect_ring1.log | cut -c1-400 | tee -a $out/summary.txt
echo "=== A/B slp" | tee -a $out/summary.txt
bash tools/ab_bench.sh $out/ab 2 base slp 2>&1 | tee -a $out/summary.txt
AB_ARGS="--batch 32" bash tools/ab_bench.sh $out/ab32 2 base slp 2>&1 | tee -a $out/summary.txt

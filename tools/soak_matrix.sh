#!/bin/bash
# The soak matrix of profiles/r4_pk_op_sel_hazard.txt in one GPU call: bash tools/soak_matrix.sh OUTDIR
#   * the product library: 600 prefills next to one busy engine with every rope vector computed twice, 400 next to two, 400 with
#     two LDS stages (must all report 0 events / 0 mismatching vectors);
#   * if lib/libq3asr_hip_ropeexp.so exists (Q3A_ALLOW_ISA_HAZARD=1 Q3A_BUILD_VARIANT=ropeexp Q3A_BUILD_DEFINES=-DQ3A_ROPE_EXPERIMENT
#     python -m qwen3_asr_rs_amd.build): the hazard-isolation variants 0 (hipcc's SLP form needs -fslp-vectorize too), 3 (packed,
#     operands swapped in registers), 4 / 5 (packed, swap by op_sel) -- 3 must stay at 0, 4 and 5 reproduce the hazard.
out=${1:-gpurun_out/soak}; mkdir -p $out
run() { tag=$1; shift; echo "=== $tag: $*" | tee -a $out/summary.txt; env "$@" timeout 900 python tools/soak_engines.py --tag $tag $SOAK_ARGS > $out/$tag.log 2>&1; grep -E "SOAK|mismatch|run " $out/$tag.log | cut -c1-300 | tail -${TAILN:-12} | tee -a $out/summary.txt; }
SOAK_ARGS="--runs 600 --load 1" run product_ring1_twice Q3A_GEMM16_RING=1 Q3A_DEBUG_ROPE_TWICE=1
SOAK_ARGS="--runs 400 --load 2" run product_ring1_load2 Q3A_GEMM16_RING=1
SOAK_ARGS="--runs 400 --load 1" run product_ring0 Q3A_GEMM16_RING=0
X=$PWD/qwen3_asr_rs_amd/lib/libq3asr_hip_ropeexp.so
if [ -f $X ]; then
  for v in 3 4 5; do SOAK_ARGS="--runs 250 --load 1" run ropeexp_var$v Q3A_LIB=$X Q3A_GEMM16_RING=1 Q3A_ROPE_VARIANT=$v Q3A_DEBUG_ROPE_TWICE=1; done
fi

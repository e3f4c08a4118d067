#!/bin/bash
# round 6, GPU call 24: HBM-side traffic of the encoder + prefill GEMMs at 32 clips, per shape, with the tile walk on and off
# (separate --pmc passes for FETCH_SIZE and WRITE_SIZE, kernel-trace only)
R=$PWD; out=gpurun_out/r6_gemm_traffic; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for p in 1 0; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace -d $R/$out/${c}_$p -o t -- env PMC_BATCH=32 Q3A_GEMM256_PERSIST=$p python $R/tools/pmc_target_enc.py > $R/$out/${c}_$p.log 2>&1
  done
done
cd $R
for p in 1 0; do
  Q3A_GEMM256_PERSIST=$p python tools/gemm_traffic.py --fetch $(find $out/FETCH_SIZE_$p -name "*_results.db" | head -1) --write $(find $out/WRITE_SIZE_$p -name "*_results.db" | head -1) --batch 32
done > gpurun_out/r6_gemm256_hbm_traffic_by_shape.txt 2>&1
rm -rf $out/FETCH_SIZE_* $out/WRITE_SIZE_*
cat gpurun_out/r6_gemm256_hbm_traffic_by_shape.txt | cut -c1-200

#!/bin/bash
# The measurement set committed under profiles/ at the end of a round (run on the GPU box): bash tools/final_measure.sh rN
# Round 6: the default bench line now carries every leg (configs[1..3] with their own in-situ roofline + PMC traffic, measured
# peaks, precise mode, natural EOS, two engines), so the separate batch-32 / 1.7B bench runs of round 5 are gone; their traces
# come out of the same run (--trace-out FILE -> FILE, FILE.0p6b_b32, FILE.1p7b_b16).
r=${1:-r6}; out=gpurun_out/final_$r; mkdir -p $out; R=$PWD
export PYTHONUNBUFFERED=1
step() { echo "=== $1 ($(date +%T))" | tee -a $out/steps.log; }
step gputest
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | tail -120 > $out/${r}_gputest_final.log; tail -3 $out/${r}_gputest_final.log
step bench
timeout 900 python bench.py --steps 20 --warmup 5 --trace-out $out/${r}_kernel_trace_b1.txt > $out/${r}_bench_final.json 2> $out/bench.err
mv $out/${r}_kernel_trace_b1.txt.0p6b_b32 $out/${r}_kernel_trace_b32.txt 2>/dev/null
mv $out/${r}_kernel_trace_b1.txt.1p7b_b16 $out/${r}_kernel_trace_1p7b_b16.txt 2>/dev/null
step bench_1p7b_b32
timeout 400 python tools/ab_knobs.py --preset 1.7b --batch 32 --rounds 3 base > $out/${r}_1p7b_b32.txt 2>&1
step mfma_table
( cd /tmp && export TMPDIR=/tmp
  timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --kernel-trace -d $R/$out/pmcb -o b -- env PMC_BATCH=32 python $R/tools/pmc_target_enc.py > $R/$out/pmcb.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --stats -d $R/$out/trc -o t -- env PMC_BATCH=32 python $R/tools/pmc_target_enc.py > $R/$out/trc.log 2>&1 )
python tools/mfma_table.py --pmc $(find $out/pmcb -name "*_results.db" | head -1) --trace $(find $out/trc -name "*_results.db" | head -1) --batch 32 > $out/${r}_mfma_table_b32.txt 2>&1
python tools/pmc_kernels.py $(find $out/pmcb -name "*_results.db" | head -1) --by-grid > $out/${r}_pmc_by_shape.txt 2>&1
# memory-side traffic of the same launches per shape (separate FETCH_SIZE / WRITE_SIZE passes)
( cd /tmp && export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do timeout 300 rocprofv3 --pmc $c --kernel-trace -d $R/$out/tr_$c -o t -- env PMC_BATCH=32 python $R/tools/pmc_target_enc.py > $R/$out/tr_$c.log 2>&1; done )
python tools/gemm_traffic.py --fetch $(find $out/tr_FETCH_SIZE -name "*_results.db" | head -1) --write $(find $out/tr_WRITE_SIZE -name "*_results.db" | head -1) --batch 32 > $out/${r}_gemm256_hbm_traffic_by_shape.txt 2>&1
rm -rf $out/pmcb $out/trc $out/tr_FETCH_SIZE $out/tr_WRITE_SIZE
step phase_probe
timeout 120 tools/bin/phase_probe layer,one > $out/${r}_phase_probe_decode_layers.txt 2>&1
step streams
timeout 300 python tools/streams_probe.py 32 > $out/${r}_streams_b32.txt 2>&1
step done
tail -3 $out/${r}_gputest_final.log; cut -c1-700 $out/${r}_bench_final.json

#!/bin/bash
# Round 5, GPU call 1: new tests first, then the measurements that need no kernel change.  Every step logs under gpurun_out/r5c1/.
O=gpurun_out/r5c1; mkdir -p $O; R=$PWD
export PYTHONUNBUFFERED=1
step() { echo "=== $1 ($(date +%T))" | tee -a $O/steps.log; }
step longform_tests
timeout 600 python -m pytest tests/test_gpu_longform.py -q -s -m gpu > $O/longform.log 2>&1; echo "rc=$?" >> $O/longform.log; tail -3 $O/longform.log
step upload_ab
timeout 300 python tools/upload_ab.py --batch 32 --steps 4 > $O/upload_ab_b32.txt 2>&1
timeout 120 python tools/upload_ab.py --batch 1 --steps 6 --variants 0:-1:8,1:-1:8 > $O/upload_ab_b1.txt 2>&1
cat $O/upload_ab_b32.txt $O/upload_ab_b1.txt
step pk_hazard
timeout 240 tools/bin/pk_hazard --seconds 0.2 > $O/pk_hazard.txt 2>&1; echo "rc=$?" >> $O/pk_hazard.txt; tail -60 $O/pk_hazard.txt
step full_gpu_suite
timeout 900 python -m pytest tests -q -m gpu --deselect tests/test_gpu_longform.py -s > $O/gputest.log 2>&1; echo "rc=$?" >> $O/gputest.log; tail -4 $O/gputest.log
step bench_default
timeout 420 python bench.py --trace-out $O/kernel_trace_b1.txt > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json
step trace_1p7b_b16
timeout 420 python bench.py --preset 1.7b --batch 16 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-extra --trace-out $O/kernel_trace_1p7b_b16.txt > $O/bench_1p7b_b16.json 2> $O/bench_1p7b_b16.err; cat $O/kernel_trace_1p7b_b16.txt | head -30
step pmc_mfma
( cd /tmp && export TMPDIR=/tmp
  timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --kernel-trace -d $R/$O/pmcb -o b -- env PMC_BATCH=32 python $R/tools/pmc_target_enc.py > $R/$O/pmcb.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --stats -d $R/$O/trc -o t -- env PMC_BATCH=32 python $R/tools/pmc_target_enc.py > $R/$O/trc.log 2>&1 )
python tools/mfma_table.py --pmc $(find $O/pmcb -name "*_results.db" | head -1) --trace $(find $O/trc -name "*_results.db" | head -1) --batch 32 > $O/mfma_table_b32.txt 2>&1
python tools/pmc_kernels.py $(find $O/pmcb -name "*_results.db" | head -1) --by-grid > $O/pmc_by_grid.txt 2>&1
rm -rf $O/pmcb $O/trc
cat $O/mfma_table_b32.txt
step done

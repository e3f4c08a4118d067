#!/bin/bash
# Two rocprofv3 --pmc passes (never combined with other trace domains) over mel + encoder + prefill of PMC_BATCH clips, then the
# per-kernel table.  bash tools/pmc_passes.sh OUTDIR [match]
out=${1:-gpurun_out/pmc}; mkdir -p $out; R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU --kernel-trace -d $R/$out/a -o a -- python $R/tools/pmc_target_enc.py > $R/$out/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --kernel-trace -d $R/$out/b -o b -- python $R/tools/pmc_target_enc.py > $R/$out/b.log 2>&1
cd $R
python tools/pmc_kernels.py $(find $out/a $out/b -name "*_results.db") --by-grid > $out/by_grid.txt 2>&1
rm -rf $out/a $out/b

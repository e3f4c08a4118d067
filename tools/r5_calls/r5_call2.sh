#!/bin/bash
# Round 5, GPU call 2: pipelined flash attention (bit-identity + A/B), two-pass gate/up skinny GEMM and split attention at 1.7B x 16,
# second hazard sweep (which neighbour, which operand), traces.
O=gpurun_out/r5c2; mkdir -p $O; R=$PWD
export PYTHONUNBUFFERED=1
step() { echo "=== $1 ($(date +%T))" | tee -a $O/steps.log; }
step tests
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_longform.py -q -s -m gpu -k "pipelined_flash or mfma_attention or stage_parity_tiny or transcribe_ptrs or quarter_workgroup or batch_above" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -5 $O/tests.log
step ab_fattn_pipe_b32
timeout 200 python tools/ab_knobs.py --preset 0.6b --batch 32 --rounds 3 base fattn_pipe=1 > $O/ab_fattn_pipe_b32.txt 2>&1; cat $O/ab_fattn_pipe_b32.txt
step ab_1p7b_b16
timeout 300 python tools/ab_knobs.py --preset 1.7b --batch 16 --rounds 3 base skinny_glu_2pass=0 dattn_batched_min_wgs=256 skinny_glu_2pass=0,dattn_batched_min_wgs=256 > $O/ab_1p7b_b16.txt 2>&1; cat $O/ab_1p7b_b16.txt
step hazard2
PKF=v_mul_plain,v_mul_sel01_hi00,v_mul_sel01,v_mul_sel10,v_mul_sel11,v_mul_sel01_hi01,v_mul_sel01_hi10,v_mul_sel10_hi01,v_mul_hi01,v_mul_hi10,v_add_sel01,v_add_sel10,v_add_sel11,v_add_hi01,v_mov_sel10,v_mov_sel01,v_fma_plain,v_fma_sel010,v_fma_sel001,v_fma_sel100,v_fma_sel011,v_fma_sel110,v_fma_sel111,v_fma_sel010_hi000,v_fma_hi011,v_fma_hi110,v_mul_sel01_mfma_waves_same_kernel
timeout 200 tools/bin/pk_hazard --seconds 0.12 --forms $PKF --aggr none,mfma_bf16,mfma_bf16_nolds,mfma_bf16_32x32,mfma_f16,lds_barrier,mfma_f32_lds,mfma_f32 > $O/pk_hazard_2.txt 2>&1; echo "rc=$?" >> $O/pk_hazard_2.txt
timeout 60 tools/bin/pk_hazard --seconds 0.12 --forms v_fma_mixlo_sel,v_fma_f16_vop3sel,v_fma_mix_sel,v_fma_f16_sel --aggr none,mfma_bf16,mfma_bf16_nolds >> $O/pk_hazard_2.txt 2>&1
grep -v "sample:" $O/pk_hazard_2.txt | cut -c1-260 | head -80
step trace_b32_pipe
( cd /tmp && export TMPDIR=/tmp
  timeout 200 rocprofv3 --kernel-trace --stats -d $R/$O/tr_base -o t -- python $R/bench.py --inner --preset 0.6b --batch 32 --new-tokens 8 --steps 3 --warmup 1 > $R/$O/tr_base.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --stats -d $R/$O/tr_pipe -o t -- env Q3A_FATTN_PIPE=1 python $R/bench.py --inner --preset 0.6b --batch 32 --new-tokens 8 --steps 3 --warmup 1 > $R/$O/tr_pipe.log 2>&1 )
for v in base pipe; do python tools/rocpd_stats.py $(find $O/tr_$v -name "*_results.db" | head -1) 2>/dev/null | grep -E "fattn|gemm256|conv1|Name|name" | head -14 > $O/trace_b32_$v.txt; rm -rf $O/tr_$v; done
cat $O/trace_b32_base.txt $O/trace_b32_pipe.txt | cut -c1-200
step config3_test
timeout 400 python -m pytest tests/test_gpu_configs.py -q -s -m gpu -k "config3 or 1p7b" > $O/config3.log 2>&1; echo "rc=$?" >> $O/config3.log; tail -5 $O/config3.log | cut -c1-250
step done

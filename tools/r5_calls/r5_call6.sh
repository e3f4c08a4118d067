#!/bin/bash
# Round 5, GPU call 6: natural-EOS run-ahead 1 vs 2 paired with fixed-N (twice), long hazard cells for the forms the product contains.
O=gpurun_out/r5c6; mkdir -p $O
export PYTHONUNBUFFERED=1
for i in 1 2; do timeout 300 python tools/eos_probe.py > $O/eos_probe_$i.txt 2>&1; grep -E "fixed-N|ahead" $O/eos_probe_$i.txt | cut -c1-200; done
PROD=v_mul_plain,v_mul_hi01,v_mul_hi10,v_add_hi01,v_add_hi10,v_fma_plain,v_fma_hi011,v_fma_hi110,v_fma_hi100,v_fma_hi101,v_fma_hi010,v_cvt_pk_bf16,v_dot2c_bf16,v_dpp_quad1032,v_dpp_quad2301,v_dpp_row_mirror,v_dpp_row_half_mirror,v_permlane32_swap,v_permlane16_swap,v_mul_sel01
timeout 300 tools/bin/pk_hazard --seconds 1.5 --forms $PROD --aggr mfma_bf16_nolds,mfma_f16 > $O/pk_hazard_product_forms_long.txt 2>&1; echo "rc=$?" >> $O/pk_hazard_product_forms_long.txt
grep -v "sample:" $O/pk_hazard_product_forms_long.txt | cut -c1-200 | head -40

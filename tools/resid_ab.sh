#!/bin/bash
# A/B of gemm256's fp32-residual epilogue (Q3A_GEMM256_RESID_PREFETCH=0|1) on ONE box: stamped per-tile timeline of the residual
# shapes (tools/bin/phase_probe gemm) and the 32-clip bench (encoder / prefill ms), interleaved.   bash tools/resid_ab.sh OUT
out=${1:-gpurun_out/resid_ab}; mkdir -p $out
for v in 0 1; do Q3A_GEMM256_RESID_PREFETCH=$v tools/bin/phase_probe gemm > $out/phase_probe_gemm_prefetch$v.txt 2>&1; done
for r in 1 2; do for v in 0 1; do
  Q3A_GEMM256_RESID_PREFETCH=$v python bench.py --batch 32 --no-cpu-baseline --no-extra --no-rocprof --no-pmc 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=j['stage_ms']
print('prefetch $v round $r value', j['value'], 'encoder_ms', s['encoder_ms'], 'prefill_ms', s['prefill_ms'], 'decode_ms', s['decode_ms'])" | tee -a $out/bench_ab.txt
done; done
grep -E "fp32 residual|mean per tile \(all\)" $out/phase_probe_gemm_prefetch0.txt | paste - - | cut -c1-250
grep -E "fp32 residual|mean per tile \(all\)" $out/phase_probe_gemm_prefetch1.txt | paste - - | cut -c1-250

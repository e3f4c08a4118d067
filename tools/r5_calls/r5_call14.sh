#!/bin/bash
# Round 5, GPU call 14: final numbers again after the attention trim: GPU suite at HEAD, bench default (+ extras), the 32-clip line with its trace and PMC traffic.
O=gpurun_out/r5c14; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1300 python -m pytest tests -m gpu -q -s 2>&1 | tail -100 > $O/r5_gputest_final.log; tail -3 $O/r5_gputest_final.log | cut -c1-300
timeout 500 python bench.py --steps 20 --warmup 5 --trace-out $O/r5_kernel_trace_b1.txt > $O/r5_bench_final.json 2> $O/bench.err
timeout 400 python bench.py --batch 32 --no-cpu-baseline --no-extra --trace-out $O/r5_kernel_trace_b32.txt > $O/r5_bench_b32_final.json 2>> $O/bench.err
python -c "
import json
d=json.loads(open('$O/r5_bench_final.json').read().strip().splitlines()[-1]); print(d['value'], d['host_to_host']['value'], d['stage_ms'], d['natural_eos']['vs_fixed_n_ms'])
for e in d['extra']: print(e['value'], e['host_to_host']['value'], e['decode_stage'])
d=json.loads(open('$O/r5_bench_b32_final.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['host_to_host']['value'], r['kernel'], r['avg_launch_us'], r['bytes_per_launch'], r['frac'], r['traffic'], r['decode_stage'])
"

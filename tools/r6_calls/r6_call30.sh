#!/bin/bash
# round 6, GPU call 30: full GPU suite at HEAD (flash-attention XCD remap, random-operand peak) + smoke + a default bench line
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r6_gputest_head_final.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6_gputest_head_final.log; tail -4 gpurun_out/r6_gputest_head_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r6_bench_head_final.json 2> gpurun_out/r6_bench_head_final.err; cut -c1-300 gpurun_out/r6_bench_head_final.json

#!/bin/bash
# The measurement set committed under profiles/ at the end of a round (run on the GPU box): bash tools/final_measure.sh rN
r=${1:-r4}; out=gpurun_out/final_$r; mkdir -p $out
python -m pytest tests -m gpu -q -s 2>&1 | tail -70 > $out/${r}_gputest_final.log
python bench.py --trace-out $out/${r}_kernel_trace_b1.txt > $out/${r}_bench_final.json 2> $out/bench.err
python bench.py --batch 32 --no-cpu-baseline --no-extra --trace-out $out/${r}_kernel_trace_b32.txt > $out/${r}_bench_b32_final.json 2>> $out/bench.err
PMC_BATCH=32 bash tools/pmc_passes.sh $out/pmc > /dev/null 2>&1; cp $out/pmc/by_grid.txt $out/${r}_pmc_by_shape.txt
python tools/streams_probe.py 1 > $out/${r}_streams_b1.txt 2>&1
python tools/streams_probe.py 32 > $out/${r}_streams_b32.txt 2>&1
tail -3 $out/${r}_gputest_final.log; cut -c1-600 $out/${r}_bench_final.json

#!/bin/bash
# round 6, GPU call 18: (a) stamped gemm256 timelines with the persistent walk, (b) gaps between consecutive kernels of the one-clip
# decode loop: is the seam between two per-step hipGraph launches wider than an in-graph kernel boundary?
timeout 200 tools/bin/phase_probe gemm > gpurun_out/r6_phase_probe_gemm256_persistent.txt 2>&1; tail -40 gpurun_out/r6_phase_probe_gemm256_persistent.txt | cut -c1-220
export TMPDIR=/tmp; R=$PWD; d=$(mktemp -d /tmp/q3a_gap_XXXX)
(cd /tmp && timeout 400 rocprofv3 --kernel-trace -d "$d" -o t -- python $R/bench.py --inner --preset 0.6b --batch 1 --seconds 30 --new-tokens 100 --steps 3 --warmup 1 > /dev/null 2>&1)
python tools/gap_stats.py "$d" > gpurun_out/r6_gap_stats_b1.txt 2>&1; cat gpurun_out/r6_gap_stats_b1.txt | cut -c1-200

#!/bin/bash
# round 6, GPU call 29: measured peaks with the random-operand GEMM; its GPU test
python -c "
from qwen3_asr_rs_amd.engine import measure_peaks
import json; print(json.dumps(measure_peaks(0, 5)))" | tee gpurun_out/r6_peaks_random_operands.json
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "peaks" 2>&1 | tail -2

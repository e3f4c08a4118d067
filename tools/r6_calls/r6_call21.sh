#!/bin/bash
# round 6, GPU call 21: the end-of-round measurement set at HEAD (persistent gemm256 walk in)
bash tools/final_measure.sh r6b

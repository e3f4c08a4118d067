#!/bin/bash
# round 6, GPU call 27: the end-of-round measurement set at HEAD (tile walk + tile-order rule), with the per-shape memory-side traffic
bash tools/final_measure.sh r6c

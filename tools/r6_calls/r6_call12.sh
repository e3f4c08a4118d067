#!/bin/bash
# round 6, GPU call 12: the end-of-round measurement set
bash tools/final_measure.sh r6

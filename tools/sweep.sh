#!/bin/bash
for e in 4 8 16 32; do
RLVI_ESTEP_BLOCK=1024 RLVI_ESTEP_E=$e python tools/time_parts.py --what estep --n 524288 --steps 50 --tag "iterative N=524288 E=$e" 2>/dev/null | tail -1
done

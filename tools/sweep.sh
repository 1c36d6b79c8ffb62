#!/bin/bash
for n in 131072 262144 524288; do
python tools/time_parts.py --what estep --n $n --steps 50 --tag "traj N=$n" 2>/dev/null | tail -1
RLVI_ESTEP_TRAJ=0 python tools/time_parts.py --what estep --n $n --steps 50 --tag "iterative N=$n" 2>/dev/null | tail -1
done

#!/bin/bash
python tools/time_parts.py --what estep --n 2048 --steps 50 --tag "iterative N=2048 (1 WG)" 2>/dev/null | tail -1
RLVI_ESTEP_TRAJ=0 python tools/time_parts.py --what estep --n 65536 --steps 50 --tag "iterative N=65536" 2>/dev/null | tail -1
python tools/time_parts.py --what estep --n 524288 --steps 50 --tag "iterative N=524288" 2>/dev/null | tail -1

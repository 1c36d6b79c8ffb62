#!/bin/bash
for g in 4 8; do
  RLVI_MSTEP_G=$g python tools/time_parts.py --what mstep --tag "tile G=$g" 2>/dev/null | tail -1
done
RLVI_MSTEP_TILE=0 python tools/time_parts.py --what mstep --tag "regs G=4" 2>/dev/null | tail -1
python tools/time_parts.py --what mstep_fwd --tag "tile fwd only" 2>/dev/null | tail -1
python tools/time_parts.py --what mstep --classes 10 --tag "C=10" 2>/dev/null | tail -1
python tools/time_parts.py --what mstep --classes 101 --tag "C=101" 2>/dev/null | tail -1
python tools/time_parts.py --what mstep --classes 1000 --rows 16384 --tag "C=1000" 2>/dev/null | tail -1
python tools/time_parts.py --what mstep --classes 104 --dtype bf16 --tag "bf16 C=104" 2>/dev/null | tail -1

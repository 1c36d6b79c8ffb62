#!/bin/bash
python tools/time_parts.py --what estep --tag "estep default" 2>/dev/null | tail -1
RLVI_ESTEP_E=8 python tools/time_parts.py --what estep --tag "estep 256thr E=8 (32 WG)" 2>/dev/null | tail -1
RLVI_ESTEP_BLOCK=1024 python tools/time_parts.py --what estep --tag "estep 1024thr E=8 (8 WG)" 2>/dev/null | tail -1
RLVI_ESTEP_BLOCK=1024 RLVI_ESTEP_E=4 python tools/time_parts.py --what estep --tag "estep 1024thr E=4 (16 WG)" 2>/dev/null | tail -1
python tools/time_parts.py --what estep --n 4096 --tag "estep N=4096 (1 WG)" 2>/dev/null | tail -1
python tools/time_parts.py --what estep --n 524288 --tag "estep N=524288" 2>/dev/null | tail -1

#!/bin/bash
for s in 6 3 2 1; do
RLVI_TJ_S=$s python tools/time_parts.py --what step --tag "step S=$s" 2>/dev/null | tail -1
RLVI_TJ_S=$s python tools/time_parts.py --what estep --tag "estep(1 round) S=$s" 2>/dev/null | tail -1
done

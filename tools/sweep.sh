#!/bin/bash
for nb in 256 342 384 512 600 683 768 820; do
RLVI_MSTEP_BLOCKS=$nb python tools/time_parts.py --what mstep --tag "tile prefetch blocks=$nb" 2>/dev/null | tail -1
done

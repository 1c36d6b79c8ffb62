#!/bin/bash
for g in 4 8; do for nb in 768 1024 1536 2048; do
RLVI_MSTEP_G=$g RLVI_MSTEP_BLOCKS=$nb python tools/time_parts.py --what mstep --tag "f32 C=100 G=$g blocks=$nb" 2>/dev/null | tail -1
done; done
for g in 2 4 8; do for nb in 512 768 1024; do
RLVI_MSTEP_G=$g RLVI_MSTEP_BLOCKS=$nb python tools/time_parts.py --what mstep --classes 104 --dtype bf16 --tag "bf16 C=104 G=$g blocks=$nb" 2>/dev/null | tail -1
done; done

#!/bin/bash
# per-generation hold of the M-step (RLVI_MSTEP_GEN, ticks of 10 ns) for launches whose waves take several tiles
run() { python tools/time_parts.py --what mstep --rows $1 --classes $2 --dtype ${4:-f32} --steps 60 --sweep RLVI_MSTEP_GEN=$3 2>&1 | grep "us/launch" | cut -c1-150; }
run 131072 64 0,460,520,580,640,700
run 131072 128 0,1000,1080,1160,1240,1320
run 262144 128 0,1080,1160,1240
run 131072 104 0,380,430,480,530,580 bf16
run 98304 100 0,850,900,950
run 524288 100 0,880,900,920

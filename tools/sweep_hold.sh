#!/bin/bash
# M-step reads-then-writes hold (RLVI_MSTEP_HOLD, ticks of 10 ns; 0 = off, -1 = from the bytes read): sweep per shape
run() { python tools/time_parts.py --what mstep --rows $1 --classes $2 --dtype ${4:-f32} --sweep RLVI_MSTEP_HOLD=$3 2>&1 | grep "us/launch"; }
run 65536 100 0,-1,340,370,400,430,460
run 49152 100 0,-1,250,280,310,340,370
run 32768 100 0,-1,150,180,210,240,270
run 16384 100 0,-1,70,100,130,160
run 65536 64 0,-1,200,230,260,290,320
run 65536 128 0,-1,440,480,520,560,600
run 65536 104 0,-1,180,210,240,270 bf16
run 65536 10 0,-1,30,50,70
run 4096 10 0,-1

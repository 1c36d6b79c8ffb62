#!/bin/bash
# M-step: 16-wave workgroups with a barrier behind the issue of the tile loads (RLVI_MSTEP_CUWIDE) against the
# four-wave workgroups, per shape; HBM-cold rotation
run() { python tools/time_parts.py --what ${5:-mstep} --rows $1 --classes $2 --dtype ${3:-f32} --tune RLVI_MSTEP_HOLD=0 --sweep RLVI_MSTEP_CUWIDE=$4 2>&1 | grep "us/launch" | cut -c1-150; }
run 65536 100 f32 0,1
run 61440 100 f32 0,1
run 57344 100 f32 0,1
run 49152 100 f32 0,1
run 32768 100 f32 0,1
run 65536 64 f32 0,1
run 65536 128 f32 0,1
run 65536 32 f32 0,1
run 65536 104 bf16 0,1
run 65536 100 f32 0,1 step

#!/bin/bash
mkdir -p gpurun_out/lab3
: > gpurun_out/lab3/variants.log
for pad in 0 12000 24000; do
  timeout -k 10 200 python tools/time_parts.py --what mstep --tune RLVI_MSTEP_WPC=64 --tune RLVI_MSTEP_LDS_PAD=$pad --tag pad$pad 2>&1 | grep -v "amdgpu.ids\|reps" >> gpurun_out/lab3/variants.log || exit 1
done
cat gpurun_out/lab3/variants.log

#!/bin/bash
mkdir -p gpurun_out/lab3
: > gpurun_out/lab3/variants.log
for round in 1 2; do
for v in gfx950 latepi dma; do
  RLVI_LIB_PATH=$PWD/rlvi_amd/librlvi_$v.so timeout -k 10 200 python tools/time_parts.py --what mstep --tag $v --sweep RLVI_MSTEP_WPC=16,12 2>&1 | grep -v amdgpu.ids >> gpurun_out/lab3/variants.log || exit 1
done
done
cat gpurun_out/lab3/variants.log

#!/bin/bash
mkdir -p gpurun_out/lab3
: > gpurun_out/lab3/variants.log
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mstep or evaluate or fused or small_loss or top1 or out_of_range" 2>&1 | tail -3
for round in 1 2; do
for v in base gfx950; do
  RLVI_LIB_PATH=$PWD/rlvi_amd/librlvi_$v.so timeout -k 10 200 python tools/time_parts.py --what mstep --tag $v 2>&1 | grep -v "amdgpu.ids\|reps" >> gpurun_out/lab3/variants.log || exit 1
done
done
cat gpurun_out/lab3/variants.log

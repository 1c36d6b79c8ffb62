#!/bin/bash
mkdir -p gpurun_out/lab1
timeout -k 10 600 python -m pytest tests -q -m gpu -x -k "thr or trunc or mask or criterion or cooperating or epoch" > gpurun_out/lab1/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/lab1/tests.log
tail -4 gpurun_out/lab1/tests.log
for n in 4096 16384 54000 65536 75750 300000 1000000; do
timeout -k 10 120 python tools/time_parts.py --what thr --n $n --tag thr$n 2>&1 | grep -v "amdgpu.ids\|reps" 
done | tee gpurun_out/lab1/thr.log
timeout -k 10 120 python tools/time_parts.py --what thr --n 65536 --tune RLVI_THR_DEBUG=1 --steps 20 2>&1 | grep -v "amdgpu.ids\|reps" | tee -a gpurun_out/lab1/thr.log

#!/bin/bash
mkdir -p gpurun_out/lab1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fused_em" > gpurun_out/lab1/tests_fused.log 2>&1
tail -40 gpurun_out/lab1/tests_fused.log
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/lab1/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/lab1/tests.log
tail -15 gpurun_out/lab1/tests.log
timeout -k 10 300 python tools/time_parts.py --what mstep --sweep RLVI_MSTEP_WPC=16,12 > gpurun_out/lab1/sweep.log 2>&1
timeout -k 10 300 python tools/time_parts.py --what mstep --tune RLVI_MSTEP_FORM=1 --tag form1 >> gpurun_out/lab1/sweep.log 2>&1
grep -v amdgpu.ids gpurun_out/lab1/sweep.log

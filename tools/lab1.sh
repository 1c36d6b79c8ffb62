#!/bin/bash
mkdir -p gpurun_out/lab1
timeout -k 10 900 python -m pytest tests -q -m gpu -k "estep or epoch or fused or cooperating or bench or train_rlvi or warm or random" > gpurun_out/lab1/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/lab1/tests.log
tail -4 gpurun_out/lab1/tests.log
RLVI_TJ_DEBUG=1 python tools/time_parts.py --what step --steps 12 2>&1 | grep -v amdgpu | grep -E "stamps|accept|band|E_k" -A3 | head -30
timeout -k 10 200 python tools/time_parts.py --what step 2>&1 | grep -v "amdgpu.ids\|reps"

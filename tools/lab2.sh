#!/bin/bash
mkdir -p gpurun_out/lab2
export RLVI_LIB_PATH=$PWD/rlvi_amd/librlvi_stamps.so
timeout -k 10 120 python tools/mstep_stamps.py --tune RLVI_MSTEP_WPC=16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/lab2/stamps.log

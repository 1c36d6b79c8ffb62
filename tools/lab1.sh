#!/bin/bash
mkdir -p gpurun_out/lab1
timeout -k 10 600 python -m pytest tests -q -m gpu -x -k "cooperating or thr or estep or epoch" > gpurun_out/lab1/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/lab1/tests.log
tail -25 gpurun_out/lab1/tests.log

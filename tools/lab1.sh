#!/bin/bash
mkdir -p gpurun_out/lab1
timeout -k 10 1100 python -m pytest tests -q -m gpu > gpurun_out/lab1/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/lab1/tests.log
tail -12 gpurun_out/lab1/tests.log
timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | tail -2

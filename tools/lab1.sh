#!/bin/bash
mkdir -p gpurun_out/lab1
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/lab1/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/lab1/tests.log
tail -30 gpurun_out/lab1/tests.log
timeout -k 10 600 python bench.py > gpurun_out/lab1/bench.json 2> gpurun_out/lab1/bench.err
echo "bench rc=$?"
tail -3 gpurun_out/lab1/bench.err
cat gpurun_out/lab1/bench.json

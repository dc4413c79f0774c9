#!/bin/bash
# GPU box: grouped weight-gradient parity test + micro-benchmark of the kernel variants
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "grouped_weight" > gpurun_out/wgrad_test.txt 2>&1 || { tail -30 gpurun_out/wgrad_test.txt; exit 1; }
tail -3 gpurun_out/wgrad_test.txt
timeout -k 10 300 python tools/wgrad_bench.py > gpurun_out/wgrad_bench.txt 2>&1 || { tail -30 gpurun_out/wgrad_bench.txt; exit 1; }
cat gpurun_out/wgrad_bench.txt

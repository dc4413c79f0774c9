#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "pipelined" > gpurun_out/pipek_test.txt 2>&1 || { tail -30 gpurun_out/pipek_test.txt; exit 1; }
tail -2 gpurun_out/pipek_test.txt
AB_VAR=LIDK_GEMM_PIPEK AB_VALUES="0 1 2" bash tools/gpu_ab.sh

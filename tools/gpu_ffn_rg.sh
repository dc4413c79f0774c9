#!/bin/bash
# 48-row vs 64-row workgroups of csrc/ffn.hip: parity tests at both heights, then the graph-timed micro-benchmark.
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_ffn.py -q -x 2>&1 | tail -4 > gpurun_out/ffn_rg_tests.txt || { cat gpurun_out/ffn_rg_tests.txt; exit 1; }
cat gpurun_out/ffn_rg_tests.txt
for rg in 4 3; do LIDK_FFN_RG=$rg timeout -k 10 200 python tools/ffn_bench.py 2>&1 | grep -v Warn; done | tee gpurun_out/ffn_rg_bench.txt

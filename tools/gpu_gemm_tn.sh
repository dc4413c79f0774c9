#!/bin/bash
# TN (weight-gradient) GEMM: parity tests + micro-benchmark per tile size (LIDK_TN_TILE: 0 auto, 64, 128)
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" > gpurun_out/t_gemm.log 2>&1; echo "t_gemm rc=$? $(tail -1 gpurun_out/t_gemm.log)"
for tile in ${TILES:-64 128}; do
  echo "== LIDK_TN_TILE=$tile"
  NT_ONLY= LIDK_TN_TILE=$tile timeout -k 10 200 python tools/gemm_bench.py 2>&1 | grep "^TN" | tee gpurun_out/gemm_tn$tile.log | cut -c1-150
done

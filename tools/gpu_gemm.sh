#!/bin/bash
# GEMM parity tests + micro-benchmark of the NT kernel variants (LIDK_GEMM_DIRECT: 0 LDS-staged epilogue, 1 direct 64x64, 128 direct 128x128)
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" > gpurun_out/t_gemm.log 2>&1; echo "t_gemm rc=$? $(tail -1 gpurun_out/t_gemm.log)"
for d in ${CONFIGS:-0 1 128}; do
  echo "== LIDK_GEMM_DIRECT=$d"
  NT_ONLY=${NT_ONLY-1} LIDK_GEMM_DIRECT=$d timeout -k 10 200 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gemm_d$d.log | cut -c1-150
done

#!/bin/bash
# tile-pipelined NT GEMM (LIDK_GEMM_PIPE: 0 off, 1 auto tiles-per-workgroup, n>1 fixed n tiles per workgroup)
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" > gpurun_out/t_gemm.log 2>&1; echo "t_gemm rc=$? $(tail -1 gpurun_out/t_gemm.log)"
for p in ${PIPES:-0 1 2 3 4}; do
  echo "== LIDK_GEMM_PIPE=$p"
  NT_ONLY=1 LIDK_GEMM_PIPE=$p timeout -k 10 200 python tools/gemm_bench.py 2>&1 | grep -E "ff up|pw1|qkv|dgrad da" | cut -c1-120
done

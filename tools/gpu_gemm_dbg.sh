#!/bin/bash
# ablation of the NT GEMM kernel phases (LIDK_GEMM_DBG bits: 1 no stores, 2 no reloads in the K loop, 4 no LDS/MFMA)
mkdir -p gpurun_out
for d in 0 1 2 3 4 5 6 7; do
  echo "== LIDK_GEMM_DBG=$d"
  NT_ONLY=1 LIDK_GEMM_DBG=$d timeout -k 10 200 python tools/gemm_bench.py 2>&1 | grep -E "ff up|ff down|qkv|dgrad \(N" | cut -c1-110
done

#!/bin/bash
# A/B of the LDS-DMA 128x128 NT GEMM (LIDK_GEMM_DMA = minimum K, 0 = off) on the transformer-backbone workloads
for v in ${VALUES:-512 0 512 0}; do
  line=$(LIDK_GEMM_DMA=$v timeout -k 10 300 python3 bench.py --model w2v2 --ragged --steps 30 --warmup 6 --no-cpu-baseline --cavg-steps 0 --fit-epochs 0 2>&1 | grep '^{')
  echo "LIDK_GEMM_DMA=$v w2v2 ragged $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
done

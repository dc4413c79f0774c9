#!/bin/bash
# A/B of the persistent 256x256 LDS-DMA NT GEMM (LIDK_GEMM_DMA256 = tile-count floor, 0 = off) on the backbone workloads, same box
C="--no-cpu-baseline --cavg-steps 0 --fit-epochs 0"
run() {  # tag, args
  for v in ${VALUES:-0 150 0 150}; do
    line=$(LIDK_GEMM_DMA256=$v timeout -k 10 400 python3 bench.py $2 $C 2>&1 | grep '^{')
    echo "LIDK_GEMM_DMA256=$v $1 $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
  done
}
run "xlsr ragged finetune" "--model xlsr --ragged --wavlm-regime finetune --steps 30 --warmup 50"
run "xlsr ragged frozen" "--model xlsr --ragged --wavlm-regime frozen --steps 30 --warmup 50"
run "wavlm finetune" "--model wavlm --wavlm-regime finetune --steps 40 --warmup 10"
run "w2v2 ragged" "--model w2v2 --ragged --steps 40 --warmup 50"

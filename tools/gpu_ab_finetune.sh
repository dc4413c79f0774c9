#!/bin/bash
# A/B of one environment switch on the WavLM fine-tune regime, 3 alternating rounds of 60 steps: AB_VAR=NAME AB_VALUES="1 0" bash tools/gpu_ab_finetune.sh
for r in 1 2 3; do for v in $AB_VALUES; do
  line=$(env $AB_VAR=$v timeout -k 10 300 python3 bench.py --model wavlm --wavlm-regime ${REGIME:-finetune} --steps 60 --warmup 8 --no-cpu-baseline --cavg-steps 0 --fit-epochs 0 2>&1 | grep '^{')
  echo "$AB_VAR=$v $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d.get("chunks_ms_per_step"))')"
done; done

#!/bin/bash
# A/B of one environment switch on the transformer-backbone workloads: AB_VAR=NAME AB_VALUES="1 0" bash tools/gpu_ab_regimes.sh
for v in $AB_VALUES; do
  for cfg in "--model wavlm --wavlm-regime finetune" "--model wavlm --wavlm-regime frozen" "--model w2v2 --ragged"; do
    line=$(env $AB_VAR=$v timeout -k 10 300 python3 bench.py $cfg --steps 30 --warmup 6 --no-cpu-baseline --cavg-steps 0 --fit-epochs 0 2>&1 | grep '^{')
    echo "$AB_VAR=$v $cfg: $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step", d["value"], d["unit"])')"
  done
done

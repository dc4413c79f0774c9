#!/bin/bash
# A/B of one environment switch on the same box: AB_VAR=NAME AB_VALUES="1 0" [BENCH_ARGS=...] bash tools/gpu_ab.sh   (two rounds each)
for r in 1 2; do for v in $AB_VALUES; do
  line=$(env $AB_VAR=$v timeout -k 10 300 python3 bench.py --no-cpu-baseline --cavg-steps 0 --fit-epochs 0 $BENCH_ARGS 2>&1 | grep '^{')
  echo "$AB_VAR=$v $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["phases_ms"])')"
done; done

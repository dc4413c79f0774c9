#!/bin/bash
# rocprofv3 kernel stats of one backbone regime: REGIME_ARGS="--model wavlm --wavlm-regime finetune" bash tools/gpu_prof_regime.sh <tag>
mkdir -p gpurun_out && rm -rf gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py $REGIME_ARGS --steps 12 --warmup 4 --no-cpu-baseline --cavg-steps 0 --fit-epochs 0 > gpurun_out/prof_$1.log 2>&1; echo "prof rc=$?"
grep '^{' gpurun_out/prof_$1.log | cut -c1-220
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${1}_kernel_stats.csv; head -25 gpurun_out/${1}_kernel_stats.csv | cut -c1-150
rm -rf gpurun_out/prof

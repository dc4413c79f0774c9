#!/bin/bash
# Kernel timeline of a short bench run under the current environment: gpurun_out/step_timeline.txt + timeline_busy.txt + kernel_stats.csv
mkdir -p gpurun_out && rm -rf gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --cavg-steps 0 --fit-epochs 0 $BENCH_ARGS > gpurun_out/prof.log 2>&1; echo "prof rc=$?"
grep '^{' gpurun_out/prof.log | cut -c1-200
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/kernel_stats.csv
t=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1)
python3 tools/timeline_busy.py "$t" gpurun_out/step_timeline.txt > gpurun_out/timeline_busy.txt 2>&1; tail -4 gpurun_out/timeline_busy.txt
rm -rf gpurun_out/prof

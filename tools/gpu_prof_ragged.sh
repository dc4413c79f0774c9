#!/bin/bash
# rocprofv3 kernel stats + queue-busy summary of the ragged (1-10 s, bucketed) Conformer workload -> gpurun_out/conformer_ragged_*
mkdir -p gpurun_out; rm -rf gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
C="--no-cpu-baseline --cavg-steps 0 --fit-epochs 0"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --ragged --steps 20 --warmup 60 $C > gpurun_out/prof_ragged.log 2>&1; echo "prof rc=$?"
grep '^{' gpurun_out/prof_ragged.log | cut -c1-200
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/conformer_ragged_kernel_stats.csv
t=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1)
TIMELINE_DELIM=wav_stats_kernel python3 tools/timeline_busy.py "$t" gpurun_out/ragged_step_timeline.txt > gpurun_out/conformer_ragged_timeline_busy.txt 2>&1; tail -3 gpurun_out/conformer_ragged_timeline_busy.txt
rm -rf gpurun_out/prof gpurun_out/ragged_step_timeline.txt

#!/bin/bash
# rocprofv3 kernel stats + queue-busy summary of one backbone workload: TAG=name ARGS="bench.py args" DELIM=kernel-name
mkdir -p gpurun_out; rm -rf gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
C="--no-cpu-baseline --cavg-steps 0 --fit-epochs 0"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py $ARGS $C > gpurun_out/prof_$TAG.log 2>&1; echo "prof rc=$?"
grep '^{' gpurun_out/prof_$TAG.log | cut -c1-200
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${TAG}_kernel_stats.csv
t=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1)
TIMELINE_DELIM=$DELIM python3 tools/timeline_busy.py "$t" gpurun_out/${TAG}_step_timeline.txt > gpurun_out/${TAG}_timeline_busy.txt 2>&1; tail -3 gpurun_out/${TAG}_timeline_busy.txt
rm -rf gpurun_out/prof gpurun_out/${TAG}_step_timeline.txt

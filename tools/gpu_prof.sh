#!/bin/bash
# rocprofv3 kernel trace of the bench (summary CSV lands in gpurun_out/prof/)
mkdir -p gpurun_out && rm -rf gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/prof.log 2>&1
echo "rc=$?"; grep '^{' gpurun_out/prof.log | cut -c1-300
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/kernel_stats.csv; head -30 gpurun_out/kernel_stats.csv | cut -c1-160

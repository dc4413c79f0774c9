#!/bin/bash
# Round-end evidence: the full bench line and the rocprofv3 kernel stats of the same command.  Outputs in gpurun_out/
# (copy bench_full.json / kernel_stats.csv into profiles/rNN/).  HBM traffic counters: tools/gpu_pmc_bench.sh.
mkdir -p gpurun_out && rm -rf gpurun_out/prof
timeout -k 10 500 python3 bench.py > gpurun_out/bench_full.log 2>&1; echo "bench rc=$?"; grep '^{' gpurun_out/bench_full.log > gpurun_out/bench_full.json; cut -c1-400 gpurun_out/bench_full.json
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/prof.log 2>&1; echo "prof rc=$?"
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/kernel_stats.csv; head -8 gpurun_out/kernel_stats.csv | cut -c1-160
